from .engine import SupervisedEngine, CGANEngine
from .supervised import SupervisedTrainer
from .cgan import CGANTrainer
