"""deepflows_amd — an MI355X-native `hip` device for the DeepFlows tensor/autograd hot path,
plus the host-side mirror of the reference's front-end (BackendTensor / Tensor / nn / optim)
that drives it.  See DESIGN.md and INTEGRATION.md at the repository root."""
from .backend_selection import backend_api, BackendTensor, Device, all_devices, cuda, hip, default_device
from .tensor import *  # noqa: F401,F403
from .tensor import Tensor, Graph
from .autograd import enable_grad, no_grad
from . import nn, optim

__version__ = "0.1.0"
