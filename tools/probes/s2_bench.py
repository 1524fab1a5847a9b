#!/usr/bin/env python3
"""Timing of the stride-2 3x3 convolution + shortcut pair launch against the two separate launches (batch 256)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepflows_amd import backend_api
dev = backend_api.Device("hip"); mod = dev.mod
rng = np.random.default_rng(0)
N = 256
def arr(shape):
    h = mod.Array(int(np.prod(shape))); mod.from_numpy(rng.standard_normal(shape).astype(np.float32), h); return h
def timeit(fn, iters=100):
    for _ in range(5): fn()
    e0, e1 = mod.event(), mod.event()
    mod.event_record(e0)
    for _ in range(iters): fn()
    mod.event_record(e1)
    return mod.event_elapsed_ms(e0, e1) * 1e3 / iters
for C, H, O in ((32, 16, 64), (64, 8, 128), (128, 4, 256)):
    x, w, wd = arr((N, C, H, H)), arr((O, C, 3, 3)), arr((O, C, 1, 1))
    y, yd = mod.Array(N * O * H * H // 4), mod.Array(N * O * H * H // 4)
    p1, p2 = mod.Array(512 * O * 3), mod.Array(512 * O * 3)
    a = timeit(lambda: mod.conv2d_forward_s2_pair(x, w, wd, y, yd, N, C, H, H, O, p1, p2, 512))
    b = timeit(lambda: mod.conv2d_forward_stats(x, w, None, y, N, C, H, H, O, 3, 2, 1, p1, 512))
    c = timeit(lambda: mod.conv2d_forward_stats(x, wd, None, yd, N, C, H, H, O, 1, 2, 0, p2, 512))
    print(f"C {C} H {H} O {O}: pair {a:.1f} us; 3x3 {b:.1f} + 1x1 {c:.1f} = {b + c:.1f} us")
