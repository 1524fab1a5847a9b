"""Containers: Sequential (mirror of DeepFlows/nn/modules/container.py:10-103) and working
ModuleList / ModuleDict (the reference's are empty stubs, container.py:106-111)."""
from collections import OrderedDict

from .module import Module


class Sequential(Module):
    def __init__(self, *args):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return Sequential(OrderedDict(list(self._modules.items())[idx]))
        return list(self._modules.values())[idx]

    def append(self, module):
        self.add_module(str(len(self)), module)
        return self

    def forward(self, x):
        for module in self._modules.values():
            x = module(x)
        return x


class ModuleList(Module):
    def __init__(self, modules=None):
        super().__init__()
        for m in (modules or []):
            self.append(m)

    def append(self, module):
        self.add_module(str(len(self._modules)), module)
        return self

    def extend(self, modules):
        for m in modules:
            self.append(m)
        return self

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]


class ModuleDict(Module):
    def __init__(self, modules=None):
        super().__init__()
        for k, m in (modules or {}).items():
            self.add_module(k, m)

    def __getitem__(self, key):
        return self._modules[key]

    def __setitem__(self, key, module):
        self.add_module(key, module)

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules)

    def keys(self):
        return self._modules.keys()

    def values(self):
        return self._modules.values()

    def items(self):
        return self._modules.items()
