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
# test/ResNet.py's ResNet-18 (64 / 128 / 256 / 512 channels on 32 x 32 inputs, no stem pool): --wide
WIDE = [("w_stem", 3, 32, 32, 64, 3, 1, 1), ("w_l1", 64, 32, 32, 64, 3, 1, 1), ("w_l2a", 64, 32, 32, 128, 3, 2, 1),
        ("w_l2", 128, 16, 16, 128, 3, 1, 1), ("w_l2d", 64, 32, 32, 128, 1, 2, 0), ("w_l3a", 128, 16, 16, 256, 3, 2, 1),
        ("w_l3", 256, 8, 8, 256, 3, 1, 1), ("w_l3d", 128, 16, 16, 256, 1, 2, 0), ("w_l4a", 256, 8, 8, 512, 3, 2, 1),
        ("w_l4", 512, 4, 4, 512, 3, 1, 1), ("w_l4d", 256, 8, 8, 512, 1, 2, 0)]
COUNT = {"stem": 1, "l1": 4, "l2a": 1, "l2": 3, "l2d": 1, "l3a": 1, "l3": 3, "l3d": 1, "l4a": 1, "l4": 3, "l4d": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--wide", action="store_true", help="the layer shapes of test/ResNet.py's ResNet-18")
    ap.add_argument("--grouped", action="store_true")
    ap.add_argument("--reverse", action="store_true")
    ap.add_argument("--group-only", default="")
    ap.add_argument("--group-count", type=int, default=0)
    args = ap.parse_args()
    dev = backend_api.Device("hip")
    mod = dev.mod
    N = args.n
    rng = np.random.default_rng(0)
    total = {}
    print(f"{'layer':6s} {'op':10s} {'us':>9s} {'TFLOP/s':>9s} {'GFLOP':>8s}")
    if args.wide:
        COUNT.update({"w_stem": 1, "w_l1": 4, "w_l2a": 1, "w_l2": 3, "w_l2d": 1, "w_l3a": 1, "w_l3": 3, "w_l3d": 1, "w_l4a": 1,
                      "w_l4": 3, "w_l4d": 1})
    for name, C, H, W, O, k, s, p in (WIDE if args.wide else LAYERS):
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
        if k == 3 and s == 1:
            part_s = mod.Array(512 * O * 3)
            if mod.conv2d_forward_stats(x, w, None, y, *geo, part_s, 512) > 0:
                ops["fwd_stats"] = lambda: mod.conv2d_forward_stats(x, w, None, y, *geo, part_s, 512)
        if k == 3 and s == 1 and C == O and hasattr(mod, "conv2d_forward_bn_input"):
            # the same convolution with the BatchNorm of its input applied on load (partials from a producer run)
            part_in, part_out = mod.Array(512 * C * 3), mod.Array(512 * O * 3)
            rows = mod.conv2d_forward_stats(x, w, None, y, *geo, part_in, 512)
            gam, bet, mean, var, aff = arr((C,)), arr((C,)), mod.Array(C), mod.Array(C), mod.Array(2 * C)
            def fwd_bn():
                return mod.conv2d_forward_bn_input(x, part_in, rows, gam, bet, 1e-5, 0.1, mean, var, None, None,
                                                   aff, w, None, y, *geo, part_out, 512)[0]
            if rows > 0 and fwd_bn():          # (only shapes that have the operand path)
                ops["fwd_bn_in"] = fwd_bn
        if k == 3 and s == 2 and p == 1 and hasattr(mod, "conv2d_dgrad_pair"):
            # the block's 1x1 / stride-2 shortcut gradient on the same launch (both read the block input)
            w2, held = arr((O, C, 1, 1)), arr((N, C, H, W))
            pair = lambda: mod.conv2d_dgrad_pair(y, w, y, w2, held, dx, N, C, H, W, O, mode="overwrite")
            if pair():
                ops["dgrad_pair"] = pair
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
            if op in ("fwd_bn_in", "dgrad_pair", "fwd_stats"):
                continue
            if not (name == "stem" and op.startswith("dgrad")):
                total[op] = total.get(op, 0.0) + us * COUNT[name]
    print("per-step totals (us):", {k: round(v, 1) for k, v in total.items()})
    if args.grouped:
        # the weight gradients of one backward sweep (every layer, with multiplicity) as ONE grouped launch
        jobs, flops = [], 0.0
        for name, C, H, W, O, k, s, p in LAYERS:
            if args.group_only and name not in args.group_only.split(","):
                continue
            oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            for _ in range(COUNT[name] if not args.group_count else args.group_count):
                h = [mod.Array(N * C * H * W), mod.Array(N * O * oh * ow), mod.Array(O * C * k * k)]
                mod.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32), h[0])
                mod.from_numpy(rng.standard_normal((N, O, oh, ow)).astype(np.float32), h[1])
                jobs.append((h, (N, C, H, W, O, k, s, p)))
                flops += 2.0 * N * oh * ow * C * k * k * O
        if args.reverse:
            jobs.reverse()                     # the order a backward sweep queues them in
        def sweep():
            mod.wgrad_group_begin()
            for h, geo in jobs:
                mod.conv2d_wgrad(h[0], h[1], h[2], *geo)
            mod.wgrad_group_end()
        for _ in range(3):
            sweep()
        e0, e1 = mod.event(), mod.event()
        mod.event_record(e0)
        for _ in range(args.iters):
            sweep()
        mod.event_record(e1)
        us = mod.event_elapsed_ms(e0, e1) * 1e3 / args.iters
        print(f"grouped wgrad sweep (20 layers incl. stem stand-alone): {us:.1f} us, {flops / us / 1e6:.2f} TFLOP/s")


if __name__ == "__main__":
    main()
