#!/usr/bin/env python3
"""Per-layer timing of the conv kernels on the ResNet-18 (CIFAR variant) shapes at batch 256:
HIP events on the library stream around `iters` back-to-back launches."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepflows_amd import backend_api  # noqa: E402

LAYERS = [("stem", 3, 32, 32, 32, 3, 1, 1), ("l1", 32, 16, 16, 32, 3, 1, 1), ("l2a", 32, 16, 16, 64, 3, 2, 1),
          ("l2", 64, 8, 8, 64, 3, 1, 1), ("l2d", 32, 16, 16, 64, 1, 2, 0), ("l3a", 64, 8, 8, 128, 3, 2, 1),
          ("l3", 128, 4, 4, 128, 3, 1, 1), ("l3d", 64, 8, 8, 128, 1, 2, 0), ("l4a", 128, 4, 4, 256, 3, 2, 1),
          ("l4", 256, 2, 2, 256, 3, 1, 1), ("l4d", 128, 4, 4, 256, 1, 2, 0)]
COUNT = {"stem": 1, "l1": 4, "l2a": 1, "l2": 3, "l2d": 1, "l3a": 1, "l3": 3, "l3d": 1, "l4a": 1, "l4": 3, "l4d": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = backend_api.Device("hip")
    mod = dev.mod
    N = args.n
    rng = np.random.default_rng(0)
    total = {}
    print(f"{'layer':6s} {'op':10s} {'us':>9s} {'TFLOP/s':>9s} {'GFLOP':>8s}")
    for name, C, H, W, O, k, s, p in LAYERS:
        if args.only and name not in args.only.split(","):
            continue
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        def arr(shape):
            h = mod.Array(int(np.prod(shape)))
            mod.from_numpy(rng.standard_normal(shape).astype(np.float32), h)
            return h
        x, w, y = arr((N, C, H, W)), arr((O, C, k, k)), arr((N, O, oh, ow))
        dx, dw = mod.Array(N * C * H * W), mod.Array(O * C * k * k)
        geo = (N, C, H, W, O, k, s, p)
        flops = 2.0 * N * oh * ow * C * k * k * O
        ops = {"fwd": lambda: mod.conv2d_forward(x, w, None, y, *geo),
               "wgrad": lambda: mod.conv2d_wgrad(x, y, dw, *geo),
               "dgrad_ow": lambda: mod.conv2d_dgrad(y, w, dx, *geo, mode="overwrite"),
               "dgrad_acc": lambda: mod.conv2d_dgrad(y, w, dx, *geo, mode="accumulate")}
        for op, fn in ops.items():
            for _ in range(3):
                fn()
            e0, e1 = mod.event(), mod.event()
            mod.event_record(e0)
            for _ in range(args.iters):
                fn()
            mod.event_record(e1)
            us = mod.event_elapsed_ms(e0, e1) * 1e3 / args.iters
            print(f"{name:6s} {op:10s} {us:9.1f} {flops / us / 1e6:9.2f} {flops / 1e9:8.3f}")
            if not (name == "stem" and op.startswith("dgrad")):
                total[op] = total.get(op, 0.0) + us * COUNT[name]
    print("per-step totals (us):", {k: round(v, 1) for k, v in total.items()})


if __name__ == "__main__":
    main()
