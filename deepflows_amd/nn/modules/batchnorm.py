"""BatchNorm2d — mirror of DeepFlows/nn/modules/batchnorm.py:8-55.  weight/bias and the
running statistics are (1,C,1,1); the running statistics are plain attributes (not
registered buffers, so they are absent from state_dict, as in the reference); biased
variance is used both for normalisation and for running_var; running = (1-m)*running + m*batch.
"""
from ... import backend_api, tensor
from ...autograd import is_grad_enable
from ...tensor import Tensor
from .. import functional as F
from .. import lazy
from ..parameter import Parameter
from .module import Module


class BatchNorm2d(Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 device='hip', dtype=None) -> None:
        super().__init__()
        kw = {"device": backend_api.Device(device), "dtype": dtype}
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        shape = (1, num_features, 1, 1)
        self.weight = Parameter(tensor.ones(shape, **kw)) if affine else None
        self.bias = Parameter(tensor.zeros(shape, **kw)) if affine else None
        self.running_mean = tensor.zeros(shape, **kw) if track_running_stats else None
        self.running_var = tensor.ones(shape, **kw) if track_running_stats else None

    def forward(self, x: Tensor, residual: Tensor = None) -> Tensor:
        """`residual` (extension): returns bn(x) + residual; on a fused device in training mode the
        add happens inside the normalisation kernel."""
        if residual is None:
            deferred = lazy.batch_norm2d(self, x)      # (behind a recorded convolution the call is recorded too: nn/lazy.py)
            if deferred is not None:
                return deferred
        x, residual = lazy.realize(x), lazy.realize(residual)
        if residual is not None:
            if self.training and bool(getattr(x.device.mod, "fused", False)):
                return F.batch_norm2d_train(x, self.weight, self.bias, self.running_mean,
                                            self.running_var, self.eps, self.momentum, residual)
            return self.forward(x) + residual
        fused = bool(getattr(x.device.mod, "fused", False))
        if self.training:
            if fused:
                return F.batch_norm2d_train(x, self.weight, self.bias, self.running_mean,
                                            self.running_var, self.eps, self.momentum)
            n, c, h, w = x.shape
            m = n * h * w
            mean = tensor.sum(x, axes=0, keepdims=True)
            mean = tensor.sum(mean, axes=2, keepdims=True)
            mean = tensor.sum(mean, axes=3, keepdims=True) / m
            diff = x - mean
            var = tensor.sum(diff * diff, axes=0, keepdims=True)
            var = tensor.sum(var, axes=2, keepdims=True)
            var = tensor.sum(var, axes=3, keepdims=True) / m
            if self.track_running_stats:
                mo = self.momentum
                self.running_mean.data = self.running_mean.data * (1 - mo) + mean.data * mo
                self.running_var.data = self.running_var.data * (1 - mo) + var.data * mo
            x_hat = (x - mean) / (var + self.eps) ** 0.5
        elif self.track_running_stats:
            if fused and not is_grad_enable():
                return F.batch_norm2d_eval(x, self.weight, self.bias, self.running_mean,
                                           self.running_var, self.eps)
            x_hat = (x - self.running_mean) / (self.running_var + self.eps) ** 0.5
        else:
            x_hat = x
        return x_hat * self.weight + self.bias if self.affine else x_hat

    def extra_repr(self):
        return (f"num_features={self.num_features}, eps={self.eps}, momentum={self.momentum}, "
                f"affine={self.affine}, track_running_stats={self.track_running_stats}")
