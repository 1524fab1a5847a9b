"""TEST INFRASTRUCTURE (oracle): the mathematically correct convolution input gradient on the primitive path.

The reference's `__im2col2d.grad_fn` (nn/functional.py:285-294) ASSIGNS the k*k taps one after the other, so
overlapping windows overwrite; the hip device's `conv_dgrad_mode = "accumulate"` computes the true
gradient instead -- the version of the same loop the reference's own 1-D code uses (`+=`, nn/functional.py:145).
This context manager switches the front end's primitive `_im2col2d.grad_fn` to that `+=` form for the duration of an
oracle run, so that the accumulate mode has a whole-step oracle too (its single-layer values are pinned by the torch
goldens `*_dx_accumulate` of tests/golden/g2_conv2d.npz).  Only tests use it; the product never imports oracle/.
"""
import contextlib


@contextlib.contextmanager
def accumulate_im2col_gradient():
    from deepflows_amd import backend_api
    from deepflows_amd.nn import functional as F
    node = F._im2col2d
    saved = node.grad_fn

    def grad_fn(self, x, grad):
        gx = backend_api.zeros((self.N, self.C, self.H, self.W), device=x.device)
        for i in range(self.k):
            i_hi = i + self.oh * self.s
            for j in range(self.k):
                j_hi = j + self.ow * self.s
                tap = grad[:, :, i, j, :, :].compact().reshape((self.N, self.C, self.oh, self.ow))
                gx[:, :, i:i_hi:self.s, j:j_hi:self.s] = gx[:, :, i:i_hi:self.s, j:j_hi:self.s] + tap
        return gx

    node.grad_fn = grad_fn
    try:
        yield
    finally:
        node.grad_fn = saved
