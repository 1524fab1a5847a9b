from .optimizer import Optimizer
from .adam import Adam
from .sgd import SGD
from .scheduler import LRScheduler, StepLR, CosineAnnealingLR, WarmupCosineLR
