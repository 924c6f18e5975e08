from .engine import SupervisedEngine
