from .module import Module
from .container import Sequential, ModuleList, ModuleDict
from .conv import Conv2d
from .batchnorm import BatchNorm2d
from .pool import MaxPool2d, AvgPool2d
from .linear import Linear
from .loss import NLLLoss, MSELoss, CrossEntropyLoss
from .activation import ReLU, Tanh, LeakyReLU, Softmax, LogSoftmax
from .dropout import Dropout
