from .engine import SupervisedEngine, CGANEngine
