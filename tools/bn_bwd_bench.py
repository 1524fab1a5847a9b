"""Timing of the overwrite-mode input-gradient kernels with and without the BatchNorm-backward forms (a pending backward
on the dY operand, partial rows for the layers the output feeds) next to the stand-alone BatchNorm backward they replace,
on the ResNet-18 (CIFAR variant) shapes at batch 256: HIP events on the library stream around `iters` back-to-back
launches (launch boundaries included).  Run as  python3 tools/bn_bwd_bench.py  (under rocprofv3: the same, after `--`)."""
import argparse
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepflows_amd import backend_api  # noqa: E402

LAYERS = [("l1", 32, 16, 16, 32, 3, 1, 1), ("l2", 64, 8, 8, 64, 3, 1, 1), ("l3", 128, 4, 4, 128, 3, 1, 1),
          ("l4", 256, 2, 2, 256, 3, 1, 1)]
PAIRS = [("l2a", 32, 16, 16, 64), ("l3a", 64, 8, 8, 128), ("l4a", 128, 4, 4, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    mod = backend_api.Device("hip").mod
    N = args.n
    rng = np.random.default_rng(0)

    def arr(shape, scale=1.0):
        h = mod.Array(int(np.prod(shape)))
        mod.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32), h)
        return h

    def pending(O, shape, nrows=256):
        p = types.SimpleNamespace()
        p.z, p.rows, p.nrows, p.K, p.kidx = arr(shape), arr((nrows, O, 2)), nrows, 2, 1
        p.mean, p.var, p.gamma, p.eps, p.m = arr((O,)), arr((O,)), arr((O,)), 1e-5, shape[0] * shape[2] * shape[3]
        mod.from_numpy(np.abs(rng.standard_normal(O)).astype(np.float32) + 0.5, p.var)
        p.dgamma, p.dbeta, p.table = mod.Array(O), mod.Array(O), mod.Array(4 * O)
        return p

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = mod.event(), mod.event()
        mod.event_record(e0)
        for _ in range(args.iters):
            fn()
        mod.event_record(e1)
        return mod.event_elapsed_ms(e0, e1) * 1e3 / args.iters

    print(f"{'layer':5s} {'kernel':34s} {'us':>8s}")
    for name, C, H, W, O, k, s, p in LAYERS:
        geo = (N, C, H, W, O, k, s, p)
        oh, ow = H // s, W // s
        dy, w, add, dx = arr((N, O, oh, ow)), arr((O, C, k, k)), arr((N, C, H, W)), mod.Array(N * C * H * W)
        pend = pending(O, (N, O, oh, ow))
        tg = [(arr((N, C, H, W)), arr((C,))), (arr((N, C, H, W)), arr((C,)))]
        rows = mod.Array(512 * C * 3)
        dg, db, dz = mod.Array(O), mod.Array(O), mod.Array(N * O * oh * ow)
        ops = {"dgrad_add (plain)": lambda: mod.conv2d_dgrad_add(dy, w, add, dx, *geo, mode="overwrite"),
               "bn2d_backward (stand-alone)": lambda: mod.bn2d_backward(dy, pend.z, pend.gamma, pend.mean, pend.var, dz, dg, db,
                                                                      N, O, oh * ow, 1e-5)}
        if mod.conv2d_bn_bwd_supported(*geo)[0]:
            ops["dgrad_add + pending"] = lambda: mod.conv2d_dgrad_bn(dy, w, add, dx, *geo, pend, None, None, 0)
            ops["dgrad_add + rows(1)"] = lambda: mod.conv2d_dgrad_bn(dy, w, add, dx, *geo, None, tg[:1], rows, 512)
            ops["dgrad_add + pending + rows(1)"] = lambda: mod.conv2d_dgrad_bn(dy, w, add, dx, *geo, pend, tg[:1], rows, 512)
            ops["dgrad_add + pending + rows(2)"] = lambda: mod.conv2d_dgrad_bn(dy, w, add, dx, *geo, pend, tg, rows, 512)
        for op, fn in ops.items():
            print(f"{name:5s} {op:34s} {timed(fn):8.2f}")
    for name, C, H, W, O in PAIRS:
        dy, dy2, w, w2 = arr((N, O, H // 2, W // 2)), arr((N, O, H // 2, W // 2)), arr((O, C, 3, 3)), arr((O, C, 1, 1))
        dx = mod.Array(N * C * H * W)
        p1, p2 = pending(O, (N, O, H // 2, W // 2)), pending(O, (N, O, H // 2, W // 2))
        tg = [(arr((N, C, H, W)), arr((C,)))]
        rows = mod.Array(512 * C * 2)
        ops = {"pair (plain)": lambda: mod.conv2d_dgrad_pair(dy, w, dy2, w2, None, dx, N, C, H, W, O, mode="overwrite")}
        if mod.conv2d_bn_bwd_supported(N, C, H, W, O, 3, 2, 1, pair=True)[0]:
            ops["pair + rows(1)"] = lambda: mod.conv2d_dgrad_pair_bn(dy, w, dy2, w2, None, dx, N, C, H, W, O, None, None, tg, rows, 512)
            ops["pair + pending x2"] = lambda: mod.conv2d_dgrad_pair_bn(dy, w, dy2, w2, None, dx, N, C, H, W, O, p1, p2, None, None, 0)
            ops["pair + pending x2 + rows(1)"] = lambda: mod.conv2d_dgrad_pair_bn(dy, w, dy2, w2, None, dx, N, C, H, W, O, p1, p2, tg,
                                                                                rows, 512)
        for op, fn in ops.items():
            print(f"{name:5s} {op:34s} {timed(fn):8.2f}")


if __name__ == "__main__":
    main()
