#!/usr/bin/env python3
"""Timing of the classifier-head kernels (pool_linear forward / backward with parts switched off, CE forward+grad)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepflows_amd import backend_api
dev = backend_api.Device("hip"); mod = dev.mod
rng = np.random.default_rng(0)
N, C, HW, P = 256, 256, 4, 10
def arr(shape):
    h = mod.Array(int(np.prod(shape))); mod.from_numpy(rng.standard_normal(shape).astype(np.float32), h); return h
x, w, b, dy = arr((N, C, HW)), arr((C, P)), arr((P,)), arr((N, P))
pooled, y, dx, dw, db = mod.Array(N * C), mod.Array(N * P), mod.Array(N * C * HW), mod.Array(C * P), mod.Array(P)
def timeit(name, fn, iters=200):
    for _ in range(5): fn()
    e0, e1 = mod.event(), mod.event()
    mod.event_record(e0)
    for _ in range(iters): fn()
    mod.event_record(e1)
    print(f"{name:28s} {mod.event_elapsed_ms(e0, e1) * 1e3 / iters:8.2f} us")
timeit("fwd", lambda: mod.pool_linear_forward(x, w, b, pooled, y, N, C, HW, P, 1e-3))
timeit("bwd all", lambda: mod.pool_linear_backward(dy, pooled, w, dx, dw, db, N, C, HW, P, 0.25))
timeit("bwd dx only", lambda: mod.pool_linear_backward(dy, pooled, w, dx, None, None, N, C, HW, P, 0.25))
timeit("bwd dw only", lambda: mod.pool_linear_backward(dy, pooled, w, None, dw, None, N, C, HW, P, 0.25))
timeit("bwd db only", lambda: mod.pool_linear_backward(dy, pooled, w, None, None, db, N, C, HW, P, 0.25))
timeit("bwd none", lambda: mod.pool_linear_backward(dy, pooled, w, None, None, None, N, C, HW, P, 0.25))
# round 6: head + loss (+ the per-image half of their backward pass) from one launch, and the sums over the images
if hasattr(mod, "head_ce_forward"):
    tgt = mod.Array(N * P); mod.from_numpy(np.eye(P, dtype=np.float32)[rng.integers(0, P, N)], tgt)
    loss, dl, rowloss = mod.Array(1), mod.Array(N * P), mod.Array(N)
    timeit("head_ce (one launch)", lambda: mod.head_ce_forward(x, w, b, tgt, pooled, y, loss, dl, dx, rowloss, N, C, HW, P, 1e-3, 0.25))
    timeit("head_ce without dx", lambda: mod.head_ce_forward(x, w, b, tgt, pooled, y, loss, dl, None, rowloss, N, C, HW, P, 1e-3, 0.25))
    timeit("head_grads (own launch)", lambda: mod.head_grads(dl, pooled, dw, db, N, C, P))
    timeit("ce fwd+grad (old)", lambda: mod.softmax_ce_forward_grad(y, tgt, loss, dl, N, P))
    print("loss", mod.to_numpy(loss, (1,), (1,), 0))
