#!/usr/bin/env python3
"""Per-family roofline table from a bench.py JSON line: usage roofline_table.py <bench.json> > table.md
MFMA-bound families against the fp32 MFMA peak (157.3 TFLOP/s), everything else against HBM (8 TB/s);
algorithmic flops / bytes are the ones ProfScope records, durations are HIP-event times
of the eager per-family pass."""
import json
import sys

PEAK_TF, PEAK_GBS = 157.3, 8000.0
d = json.load(open(sys.argv[1]))
print(f"| family | launches/step | avg us | ms/step | bound | achieved | % of peak |")
print("|---|---|---|---|---|---|---|")
tot = 0.0
for k, v in d["kernels"].items():
    tot += v["ms_per_step"]
    if v.get("bound", "mfma" if v.get("tflops", 0) > 1.0 else "hbm") == "mfma":
        bound, ach, frac = "fp32 MFMA", f"{v['tflops']:.1f} TFLOP/s", v["tflops"] / PEAK_TF
    else:
        ex = f" ({v['executed_tflops']:.1f} TFLOP/s executed)" if "executed_tflops" in v else ""
        bound, ach, frac = "HBM", f"{v['gbs']:.0f} GB/s{ex}", v["gbs"] / PEAK_GBS
    print(f"| {k} | {v['calls_per_step']:.0f} | {v['avg_us']:.1f} | {v['ms_per_step']:.3f} | {bound} | {ach} | {100 * frac:.1f} |")
print(f"\nSum of families {tot:.3f} ms (eager, with event overhead); replayed step {d['ms_per_step']:.3f} ms = "
      f"{d['value']:.0f} images/s.")
