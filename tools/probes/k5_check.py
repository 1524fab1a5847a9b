#!/usr/bin/env python3
"""Outputs of the CNN workload's 5x5 convolution (32 -> 64 channels, 16 x 16, batch 256) to a file: run once per
build / switch setting and compare (tools/probes/k5_check.py out.npz; DFHIP_IGEMM_K5=0 ... other.npz)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepflows_amd import backend_api
dev = backend_api.Device("hip"); mod = dev.mod
rng = np.random.default_rng(0)
N, C, H, W, O, k, s, p = 256, 32, 16, 16, 64, 5, 1, 2
x = rng.standard_normal((N, C, H, W)).astype(np.float32)
w = (rng.standard_normal((O, C, k, k)) / 28).astype(np.float32)
dy = rng.standard_normal((N, O, H, W)).astype(np.float32)
def arr(a):
    h = mod.Array(a.size); mod.from_numpy(a, h); return h
xh, wh, dyh = arr(x), arr(w), arr(dy)
geo = (N, C, H, W, O, k, s, p)
y, dxo, dxa, dw = mod.Array(N * O * H * W), mod.Array(x.size), mod.Array(x.size), mod.Array(w.size)
part = mod.Array(512 * O * 3)
rows = mod.conv2d_forward_stats(xh, wh, None, y, *geo, part, 512)
mod.conv2d_dgrad(dyh, wh, dxo, *geo, mode="overwrite")
mod.conv2d_dgrad(dyh, wh, dxa, *geo, mode="accumulate")
mod.conv2d_wgrad(xh, dyh, dw, *geo)
back = lambda h: mod.to_numpy(h, (h.size,), (1,), 0)
np.savez(sys.argv[1], y=back(y), dxo=back(dxo), dxa=back(dxa), dw=back(dw), rows=rows, part=back(part)[:max(rows, 1) * O * 3])
print("rows", rows)
