"""Global autograd switch and the no_grad / enable_grad context-decorators
(mirror of DeepFlows/autograd.py:3-74)."""
import functools

grad_enable = True


def is_grad_enable():
    return grad_enable


def set_grad_enabled(mode: bool):
    global grad_enable
    grad_enable = bool(mode)


class _GradMode:
    _mode = True

    def __enter__(self):
        self.prev = is_grad_enable()
        set_grad_enabled(self._mode)

    def __exit__(self, exc_type, exc_value, tb):
        set_grad_enabled(self.prev)

    def __call__(self, func):
        @functools.wraps(func)
        def wrapped(*args, **kwargs):
            with self.__class__():
                return func(*args, **kwargs)
        return wrapped


class no_grad(_GradMode):
    _mode = False


class enable_grad(_GradMode):
    _mode = True
