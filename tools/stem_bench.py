"""Back-to-back timing of the stem pool(relu(bn(conv(x)))) at batch 256: the one-node form without the convolution's
output tensor (dfhip_stem.hip) against the two-node form (convolution with partials + fused bn/relu/pool; backward =
dfhip_stem_backward).  HIP events on the library stream around `iters` calls; run under rocprofv3 --kernel-trace for
the per-kernel split:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/stem_bench.py
(the interpreter itself after `--`, never this file through an `env` hop: the profiler's preloaded library has initialised the
GPU by then, and an exec behind that takes the machine down on this pool)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepflows_amd import backend_api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    mod = backend_api.Device("hip").mod
    N, C, H, W, O = args.n, 3, 32, 32, 32
    rng = np.random.default_rng(0)

    def arr(shape, data=None):
        h = mod.Array(int(np.prod(shape)))
        mod.from_numpy((rng.standard_normal(shape) if data is None else data).astype(np.float32), h)
        return h
    x, w = arr((N, C, H, W)), arr((O, C, 3, 3))
    gam, bet = arr((O,), np.abs(rng.standard_normal(O)) + 0.5), arr((O,))
    z, y, g = mod.Array(N * O * H * W), mod.Array(N * O * H * W // 4), arr((N, O, H // 2, W // 2))
    mean, var, rm, rv = mod.Array(O), mod.Array(O), arr((O,)), arr((O,), np.ones(O))
    part = mod.Array(512 * O * 3)
    dw, dg, db = mod.Array(O * C * 9), mod.Array(O), mod.Array(O)
    geo = (N, C, H, W, O, 3, 1, 1)

    def old_fwd():
        rows = mod.conv2d_forward_stats(x, w, None, z, *geo, part, 512)
        mod.bn_relu_maxpool2_forward_partials(z, part, rows, gam, bet, y, mean, var, rm, rv, N, O, H, W, 1e-5, 0.1)

    def old_bwd():
        assert mod.stem_backward(x, z, y, g, gam, bet, mean, var, dw, dg, db, *geo, 1e-5)

    def new_fwd():
        assert mod.stem_forward_noz(x, w, None, gam, bet, y, mean, var, rm, rv, *geo, 1e-5, 0.1)

    def new_bwd():
        assert mod.stem_backward_noz(x, w, None, y, g, gam, bet, mean, var, dw, None, dg, db, *geo, 1e-5)

    for name, fn in (("old_fwd", old_fwd), ("old_bwd", old_bwd), ("new_fwd", new_fwd), ("new_bwd", new_bwd)):
        if args.only and name not in args.only.split(","):
            continue
        for _ in range(3):
            fn()
        e0, e1 = mod.event(), mod.event()
        mod.event_record(e0)
        for _ in range(args.iters):
            fn()
        mod.event_record(e1)
        print(f"{name:8s} {mod.event_elapsed_ms(e0, e1) * 1e3 / args.iters:8.2f} us per call")


if __name__ == "__main__":
    main()
