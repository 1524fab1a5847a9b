"""Re-exports with the reference's names (DeepFlows/backend_selection.py:6-18)."""
from . import backend as backend_api
from .backend import (BackendDevice as Device, all_devices, cpu, cuda, default_device, hip)

BackendTensor = backend_api.BackendTensor
