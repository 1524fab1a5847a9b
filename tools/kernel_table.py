#!/usr/bin/env python3
"""ONE table per measurement set: kernel, family, launches per step, duration, algorithmic and executed fraction of its
roofline, matrix-pipe busy fraction, HBM traffic against the algorithmic bytes.

    python tools/kernel_table.py <bench.json> <pmc_traffic.json> [<kernel_stats.txt>] > table.md

  bench.json        the JSON line of bench.py (its `kernel_table`: launches, the kernels' own start / end stamps from event
                    pairs bound to each dispatch, algorithmic flops / bytes of the family calls a kernel is the first
                    launch of)
  pmc_traffic.json  tools/pmc_summary.py --json of the rocprofv3 --pmc passes of the same command on the same build
  kernel_stats.txt  optional: tools/kernel_stats_txt.py of a rocprofv3 --kernel-trace --stats run (its average per
                    kernel is printed beside bench.py's)
MFMA-bound kernels are priced against 157.3 TFLOP/s (fp32 matrix peak at 2.4 GHz), the others against 8 TB/s."""
import json
import sys

PEAK_TF, PEAK_GBS = 157.3, 8000.0
MFMA_FAMILIES = {"conv2d_fwd_igemm", "conv2d_wgrad_grouped", "conv2d_wgrad_igemm", "conv2d_dgrad_igemm", "gemm_mfma_f32"}


def main():
    bench = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pmc = json.load(open(sys.argv[2]))
    trace = {}
    if len(sys.argv) > 3:
        for line in open(sys.argv[3]):
            parts = line.rstrip("\n").rsplit(None, 4)
            if len(parts) == 5 and parts[1].isdigit():
                trace[parts[0].strip()] = float(parts[3]) / 1e3
    pk = pmc.get("kernels", {})
    print(f"Measurement set: bench `{bench['metric']}` {bench['value']:.0f} {bench['unit']} ({bench['ms_per_step']:.4f} ms per step, "
          f"{bench['config'].get('dispatch')}); counters: git {pmc.get('git')}, kernel sources {pmc.get('csrc_sha16')}, "
          f"{pmc.get('steps_profiled')} steps profiled.\n")
    print("| kernel | family | launches / step | us (bench) | us (rocprof) | alg GFLOP | exec GFLOP | frac alg | frac exec | MFMA busy (in the PMC pass) | alg MB | HBM MB (counters) | traffic / alg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot_us = 0.0
    for r in bench["kernel_table"]:
        k = r["kernel"]
        c = pk.get(k, {})
        us = r["avg_us"]
        tot_us += us * r["launches_per_step"]
        mf = r["family"] in MFMA_FAMILIES
        alg_gf, alg_mb = r["alg_gflop_per_launch"], r["alg_mb_per_launch"]
        ex = c.get("executed_flops_per_launch")
        hbm = (c.get("fetch_bytes_per_launch", 0.0) + c.get("write_bytes_per_launch", 0.0)) if c else None
        if mf and alg_gf > 0:
            fa = f"{alg_gf / us * 1e3 / PEAK_TF:.3f}"                       # GFLOP / us = 1e3 TFLOP/s
            fe = f"{ex / 1e9 / us * 1e3 / PEAK_TF:.3f}" if ex else "-"
        else:
            fa = f"{alg_mb / us / (PEAK_GBS / 1e3):.3f} (HBM)" if alg_mb > 0 else "-"    # MB / us = TB/s
            fe = "-"
        busy = f"{c['mfma_busy']:.3f}" if c.get("mfma_busy") else "-"
        name = k if len(k) <= 64 else k[:61] + "..."
        print(f"| `{name}` | {r['family']} | {r['launches_per_step']:g} | {us:.1f} | {trace.get(k, float('nan')):.1f} | "
              f"{alg_gf:.3f} | {(ex / 1e9 if ex else 0):.3f} | {fa} | {fe} | {busy} | {alg_mb:.2f} | "
              f"{(hbm / 1e6 if hbm else float('nan')):.2f} | {(hbm / 1e6 / alg_mb if hbm and alg_mb > 0 else float('nan')):.2f} |")
    print(f"\nSum of the kernels' own durations: {tot_us / 1e3:.4f} ms per step (eager pass behind the gate); replayed step "
          f"{bench['ms_per_step']:.4f} ms.")
    rf = bench.get("roofline") or {}
    if rf:
        print(f"\nDominant family `{rf.get('kernel')}`: frac (algorithmic flops) {rf.get('frac', float('nan')):.3f}, frac_executed "
              f"{rf.get('frac_executed') or float('nan'):.3f}, mfma_busy (PMC pass) {rf.get('mfma_busy') or float('nan'):.3f}, traffic "
              f"{(rf.get('traffic') or 0) / 1e6:.2f} MB per launch against {rf.get('algorithmic_bytes_per_launch', 0) / 1e6:.2f} MB "
              f"algorithmic; clock under an MFMA-only load {rf.get('mfma_clock_mhz_under_load')} MHz.")


if __name__ == "__main__":
    main()
