from .modules import *  # noqa: F401,F403
from .parameter import Parameter
from . import init
from . import functional
