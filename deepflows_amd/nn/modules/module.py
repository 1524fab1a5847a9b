"""nn.Module — lean mirror of DeepFlows/nn/modules/module.py: attribute-assignment
registration of Parameters and sub-Modules (:373-444), `parameters()` / `named_parameters()`
(:544-604), `train()` / `eval()` which also flip the GLOBAL autograd flag (:754-775),
`state_dict` / `load_state_dict` (:471-542), `zero_grad`, `to` / `move`.

As in the reference, a plain Python list of modules is NOT registered (that is how the
reference's ResNet scripts end up optimising 5 of their 62 tensors); use
`nn.ModuleList` / `nn.Sequential` to register them.
"""
from collections import OrderedDict
from copy import deepcopy

import numpy as np

from ... import backend_api
from ...autograd import set_grad_enabled
from ...tensor import Tensor
from ..parameter import Parameter


class Module:
    def __init__(self) -> None:
        object.__setattr__(self, 'training', True)
        object.__setattr__(self, '_parameters', OrderedDict())
        object.__setattr__(self, '_buffers', OrderedDict())
        object.__setattr__(self, '_modules', OrderedDict())
        object.__setattr__(self, 'device', None)

    def __call__(self, *inputs, **kwargs):
        return self.forward(*inputs, **kwargs)

    def forward(self, *inputs):
        raise NotImplementedError(f"{type(self).__name__} has no forward()")

    # ---- registration ----------------------------------------------------------------------
    def register_parameter(self, name, param):
        if '_parameters' not in self.__dict__:
            raise AttributeError("cannot assign parameter before Module.__init__() call")
        if param is not None and not isinstance(param, Parameter):
            raise TypeError(f"cannot assign '{type(param).__name__}' object to parameter '{name}'")
        self._parameters[name] = param

    def register_buffer(self, name, tensor, persistent=True):
        if tensor is not None and not isinstance(tensor, Tensor):
            raise TypeError(f"cannot assign '{type(tensor).__name__}' object to buffer '{name}'")
        self._buffers[name] = tensor

    def add_module(self, name, module):
        if module is not None and not isinstance(module, Module):
            raise TypeError(f"{type(module).__name__} is not a Module subclass")
        self._modules[name] = module

    def __setattr__(self, name, value):
        params = self.__dict__.get('_parameters')
        modules = self.__dict__.get('_modules')
        buffers = self.__dict__.get('_buffers')
        if isinstance(value, Parameter):
            if params is None:
                raise AttributeError("cannot assign parameters before Module.__init__() call")
            for d in (self.__dict__, buffers, modules):
                d.pop(name, None)
            self.register_parameter(name, value)
        elif params is not None and name in params:
            if value is not None:
                raise TypeError(f"cannot assign this value as parameter '{name}' (Parameter or None expected)")
            self.register_parameter(name, value)
        elif isinstance(value, Module):
            if modules is None:
                raise AttributeError("cannot assign module before Module.__init__() call")
            for d in (self.__dict__, params, buffers):
                d.pop(name, None)
            modules[name] = value
        elif modules is not None and name in modules:
            if value is not None:
                raise TypeError(f"cannot assign this as child module '{name}' (nn.Module or None expected)")
            modules[name] = value
        elif buffers is not None and name in buffers:
            if value is not None and not isinstance(value, Tensor):
                raise TypeError(f"cannot assign this as buffer '{name}' (Tensor or None expected)")
            buffers[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        for store in ('_parameters', '_buffers', '_modules'):
            d = self.__dict__.get(store)
            if d is not None and name in d:
                return d[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __delattr__(self, name):
        for store in (self._parameters, self._buffers, self._modules):
            if name in store:
                del store[name]
                return
        object.__delattr__(self, name)

    # ---- traversal ---------------------------------------------------------------------------
    def children(self):
        for _, m in self.named_children():
            yield m

    def named_children(self):
        seen = set()
        for name, m in self._modules.items():
            if m is not None and id(m) not in seen:
                seen.add(id(m))
                yield name, m

    def modules(self):
        for _, m in self.named_modules():
            yield m

    def named_modules(self, memo=None, prefix='', remove_duplicate=True):
        memo = set() if memo is None else memo
        if id(self) in memo and remove_duplicate:
            return
        memo.add(id(self))
        yield prefix, self
        for name, m in self._modules.items():
            if m is None:
                continue
            yield from m.named_modules(memo, prefix + ('.' if prefix else '') + name, remove_duplicate)

    def _named_members(self, getter, prefix='', recurse=True):
        seen = set()
        mods = self.named_modules(prefix=prefix) if recurse else [(prefix, self)]
        for mprefix, m in mods:
            for k, v in getter(m):
                if v is None or id(v) in seen:
                    continue
                seen.add(id(v))
                yield mprefix + ('.' if mprefix else '') + k, v

    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        yield from self._named_members(lambda m: m._parameters.items(), prefix, recurse)

    def parameters(self, recurse=True):
        for _, p in self.named_parameters(recurse=recurse):
            yield p

    def named_buffers(self, prefix='', recurse=True):
        yield from self._named_members(lambda m: m._buffers.items(), prefix, recurse)

    def buffers(self, recurse=True):
        for _, b in self.named_buffers(recurse=recurse):
            yield b

    # ---- modes ---------------------------------------------------------------------------------
    def train(self, mode: bool = True):
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        self.training = mode
        set_grad_enabled(self.training)      # the reference ties autograd to the mode (:764)
        for m in self.children():
            m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def requires_grad_(self, requires_grad=True):
        for p in self.parameters():
            p.requires_grad = requires_grad
        return self

    def zero_grad(self, set_to_none=True):
        for p in self.parameters():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad = backend_api.zeros_like(p.data)

    # ---- state ---------------------------------------------------------------------------------
    def state_dict(self, prefix=''):
        out = OrderedDict()
        for name, p in self.named_parameters(prefix=prefix):
            out[name] = p
        for name, b in self.named_buffers(prefix=prefix):
            out[name] = b
        return out

    def load_state_dict(self, state_dict, strict=True):
        own = self.state_dict()
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {missing}, unexpected keys {unexpected}")
        for k, dst in own.items():
            if k not in state_dict:
                continue
            src = state_dict[k]
            arr = src.numpy() if isinstance(src, Tensor) else np.asarray(src, dtype=np.float32)
            if tuple(arr.shape) != tuple(dst.shape):
                raise RuntimeError(f"size mismatch for {k}: {arr.shape} vs {tuple(dst.shape)}")
            dst.data.device.from_numpy(np.ascontiguousarray(arr), dst.data._handle)
        return missing, unexpected

    def load_weights(self, weights):
        self.load_state_dict(weights, strict=False)

    def to(self, device):
        dev = backend_api.Device(device) if isinstance(device, str) else device
        m = deepcopy(self)
        m.move(dev)
        return m

    def move(self, device):
        self.device = device
        for name, p in list(self._parameters.items()):
            if p is not None:
                self._parameters[name] = p.to(device)
        for name, b in list(self._buffers.items()):
            if b is not None:
                self._buffers[name] = b.to(device)
        for m in self._modules.values():
            if m is not None:
                m.move(device)

    def extra_repr(self):
        return ''

    def __repr__(self):
        lines = [f"  ({k}): " + repr(m).replace('\n', '\n  ') for k, m in self._modules.items()]
        head = f"{type(self).__name__}({self.extra_repr()}"
        return head + ")" if not lines else head + "\n" + "\n".join(lines) + "\n)"
