"""Device registry + strided array (mirror of DeepFlows/backend/__init__.py)."""
from .backend_tensor import *  # noqa: F401,F403
from .backend_tensor import (BackendDevice, BackendTensor, Device, all_devices, cpu, cuda,
                             default_device, hip, register_device, set_default_device,
                             unregister_device)
