"""Summarise rocprofv3 --pmc passes (counter_collection.csv files; separate passes for FETCH_SIZE, WRITE_SIZE and the SQ
counters, --kernel-trace only, as MI355X_MICROARCH.md prescribes).

    python tools/pmc_summary.py <counter_collection.csv> ...                      per-kernel text table
    python tools/pmc_summary.py --json <steps> <out.json> [--workload w] [--dgrad m] <csv> ...
                                                                                  per-family AND per-kernel JSON

Units and corrections
  * FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced
    reads (MI355X_MICROARCH.md, HBM section): the doubled figure is what is compared with a byte count.
  * SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = fp32 MFMA flops EXECUTED (a 32x32x2 instruction = 4 096 flops = 8 MOPS).
  * SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the SIMDs; 64 flops per busy cycle, so busy cycles x 64
    equals the executed flops (the two counters agree to the last digit on this part).
  * MFMA busy fraction of a kernel = busy cycles / (1 024 SIMDs x the kernel's own duration x 2.4 GHz): duration from the
    dispatch's start / end stamps OF THE SAME PASS, the data-sheet clock.  By the line above it equals the executed-flop
    fraction of the 157.3 TFLOP/s peak.  (Round 2 divided by GRBM_GUI_ACTIVE / 8 instead: that counter keeps running
    through a short dispatch's ramp and drain -- 385 us for 232 us of kernels -- and read 1.7x low.)
"""
import csv
import sys
from collections import defaultdict

PEAK_CLOCK_HZ = 2.4e9
SIMDS = 1024


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("dfhip::", "")
    if name.startswith("void "):
        name = name[5:]
    # cut the argument list: the first '(' outside template brackets
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    return name.strip()


def collect(paths):
    """{kernel: {counter: [dispatches, sum]}, '_ns': {counter: summed duration of the dispatches that counter was read on}}"""
    agg = defaultdict(lambda: {"c": defaultdict(lambda: [0, 0.0]), "ns": defaultdict(float)})
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                a = agg[k]["c"][c]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                try:
                    agg[k]["ns"][c] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
    return agg


def kernel_rows(agg):
    rows = {}
    for k, d in agg.items():
        cs = d["c"]
        fe, wr = cs.get("FETCH_SIZE", [0, 0.0]), cs.get("WRITE_SIZE", [0, 0.0])
        busy, mops = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0]), cs.get("SQ_INSTS_VALU_MFMA_MOPS_F32", [0, 0.0])
        ent = {"dispatches": max(v[0] for v in cs.values())}
        if fe[0]:
            ent["fetch_bytes_per_launch"] = 2.0 * fe[1] / fe[0] * 1024.0          # gfx950 correction
        if wr[0]:
            ent["write_bytes_per_launch"] = wr[1] / wr[0] * 1024.0
        if busy[0]:
            ns = d["ns"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            ent["mfma_busy_cycles_per_launch"] = busy[1] / busy[0]
            ent["us_in_sq_pass"] = ns / busy[0] / 1e3
            if ns > 0:
                ent["mfma_busy"] = busy[1] / (SIMDS * ns * 1e-9 * PEAK_CLOCK_HZ)
        if mops[0]:
            ent["executed_flops_per_launch"] = mops[1] / mops[0] * 512.0
        rows[k] = ent
    return rows


def main(paths):
    rows = kernel_rows(collect(paths))
    order = sorted(rows, key=lambda k: -(rows[k].get("fetch_bytes_per_launch", 0) + rows[k].get("write_bytes_per_launch", 0)) *
                   rows[k]["dispatches"])
    print(f"{'kernel':70s} {'calls':>6s} {'FETCH MB (x2)':>14s} {'WRITE MB':>9s} {'exec GFLOP':>11s} {'MFMA busy':>10s} {'us (SQ pass)':>13s}")
    for k in order:
        r = rows[k]
        f = lambda key, sc, fmt: (fmt % (r[key] / sc)) if key in r else "-"
        nm = k if len(k) < 70 else k[:67] + "..."
        print(f"{nm:70s} {r['dispatches']:6d} {f('fetch_bytes_per_launch', 1e6, '%.3f'):>14s} {f('write_bytes_per_launch', 1e6, '%.3f'):>9s} "
              f"{f('executed_flops_per_launch', 1e9, '%.3f'):>11s} {f('mfma_busy', 1.0, '%.3f'):>10s} {f('us_in_sq_pass', 1.0, '%.1f'):>13s}")


FAMILIES = {
    # kernel-name prefix (a trailing $ = exact name) -> the family bench.py reports (dfhip ProfScope names);
    # first match wins
    "wgrad_slab_sum": ("k_wgrad_group_sum",),
    "conv2d_wgrad_grouped_bf16": ("k_wgrad_group_bf16",),
    "conv2d_wgrad_grouped": ("k_wgrad_group",),
    # round 6
    "droplin_ce_fwd_bwd": ("k_head_ce<16, 1, 1, true>", "k_head_ce<16, 2, 1, true>", "k_head_ce<16, 4, 1, true>", "k_head_ce<16, 8, 1, true>"),
    "head_ce_fwd_bwd": ("k_head_ce",),
    "head_grads": ("k_head_grads",),
    "loss_finish": ("k_loss_finish",),
    "mlp_fc1_fwd": ("k_mlp_fc1",),
    "mlp2_ce_fwd": ("k_mlp2_fwd", "k_mlp2_tail"),
    "mlp2_ce_bwd_sgd": ("k_mlp2_bwd<1, true>", "k_mlp2_bwd<2, true>", "k_mlp2_bwd<4, true>"),
    "mlp2_ce_bwd": ("k_mlp2_bwd",),
    "dropout_fwd": ("k_dropout<false>",),
    "dropout_bwd": ("k_dropout<true>",),
    "conv2d_wgrad_igemm": ("k_conv_wgrad_rows", "k_conv_wgrad_v3", "k_conv_wgrad_tap", "k_conv_wgrad<", "k_wgrad_tap_sum",
                           "k_flat_slab_sum", "k_wgrad_stem3", "k_stem_bwd<1, true>", "k_stem_bwd<2, true>",
                           "k_stem_bwd<3, true>", "k_stem_slab_sum"),
    "conv2d_fwd_igemm": ("k_conv3x3_rowmma", "k_conv3x3s2_pair", "k_conv3x3s2_img4_pair", "k_conv_igemm_v4<3, 0", "k_conv_igemm_v4<1, 0",
                         "k_conv_igemm_v4<5, 0", "k_conv_igemm$", "k_conv_rows3", "k_conv1x1s2_fwd", "k_conv3x3_img2_fwd",
                         "k_conv3x3_img2_all", "k_conv3x3_img4_fwd", "k_stem_fwd_sel"),
    "conv2d_dgrad_igemm": ("k_conv_igemm_v4<3, 1", "k_conv_igemm_v4<1, 1", "k_conv_igemm_v4<5, 1"),
    "conv2d_dgrad_overwrite_igemm": ("k_conv_igemm_v4<3, 2", "k_conv_igemm_v4<1, 2", "k_conv_igemm_v4<5, 2", "k_dgrad_ow_s1", "k_dgrad_ow_img"),
    "bn2d_fwd": ("k_bn_apply_partials", "k_bn_finalize_partials", "k_bn_stats", "k_bn_apply", "k_bn_fused_fwd"),
    "pool_linear_fwd": ("k_pool_linear_fwd",),
    "pool_linear_bwd": ("k_pool_linear_bwd",),
    "softmax_ce_fwd": ("k_ce_small", "k_ce_rows", "k_ce_finish"),
    "adam_step": ("k_adam",),
    "sgd_step": ("k_sgd",),
    "bn_relu_pool_fwd": ("k_bn_relu_pool_fwd", "k_sel_bn_relu_fwd"),
    "bn_relu_pool_bwd": ("k_bn_relu_pool_bwd", "k_stem_bwd<1, false>", "k_stem_bwd<2, false>", "k_stem_bwd<3, false>"),
    "bn2d_bwd": ("k_bn_fused_bwd", "k_bn_bwd_stats", "k_bn_bwd_apply"),
    "gemm_mfma_f32": ("k_gemm",),
}


def _family_of(kernel):
    k = short(kernel)
    # kernels that serve the accumulate-mode input gradient through a template flag (`DG`) belong to that family
    if k.startswith("k_dgrad_s2_acc"):
        return "conv2d_dgrad_igemm"
    if k.startswith("k_conv3x3_img2_all<"):          # <C, DG, BNC>
        args = [a.strip() for a in k[len("k_conv3x3_img2_all<"):-1].split(",")]
        if len(args) > 1 and args[1] == "true":
            return "conv2d_dgrad_igemm"
    if k.startswith("k_conv3x3_rowmma<"):
        args = [a.strip() for a in k[len("k_conv3x3_rowmma<"):-1].split(",")]
        if len(args) > 7 and args[7] == "true":      # <LW, C, NT, VDPP, TS, OW, CG, DG, BNI> since round 5
            return "conv2d_dgrad_igemm"
    for fam, prefixes in FAMILIES.items():
        if any((k == pr[:-1]) if pr.endswith("$") else k.startswith(pr) for pr in prefixes):
            return fam
    return None


def _build_identity():
    """(git hash, sha256[:16] of the kernel sources) of the tree the pass was taken on: bench.py quotes a traffic
    file only for the same kernel sources."""
    import glob
    import hashlib
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(root, "deepflows_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(root, "deepflows_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(root, "include", "*.h"))):
        h.update(open(p, "rb").read())
    try:
        git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        git = ""
    if not git:                                   # the GPU box has no .git: tools/grun.sh leaves the hash in .git_head
        try:
            git = open(os.path.join(root, ".git_head")).read().strip()
        except OSError:
            git = os.environ.get("DF_GIT_HEAD", "?")
    return git, h.hexdigest()[:16]


def family_json(paths, steps, out_path, workload="resnet18", dgrad="overwrite"):
    """Per family and per kernel, per step: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE), executed fp32-MFMA flops, matrix-pipe
    busy fraction over the kernels' own durations."""
    import json
    agg = collect(paths)
    krows = kernel_rows(agg)
    fam = defaultdict(lambda: {"hbm_bytes": 0.0, "launches": 0, "busy_cycles": 0.0, "exec_flops": 0.0, "sq_ns": 0.0})
    for k, r in krows.items():
        f = _family_of(k)
        if f is None:
            continue
        t = fam[f]
        n = r["dispatches"]
        t["launches"] += n
        t["hbm_bytes"] += n * (r.get("fetch_bytes_per_launch", 0.0) + r.get("write_bytes_per_launch", 0.0))
        t["busy_cycles"] += n * r.get("mfma_busy_cycles_per_launch", 0.0)
        t["exec_flops"] += n * r.get("executed_flops_per_launch", 0.0)
        t["sq_ns"] += n * r.get("us_in_sq_pass", 0.0) * 1e3
    out = {}
    for f, t in fam.items():
        ent = {"hbm_bytes_per_step": t["hbm_bytes"] / steps, "kernel_launches_per_step": t["launches"] / steps}
        if t["sq_ns"] > 0:
            ent["executed_flops_per_step"] = t["exec_flops"] / steps
            ent["mfma_busy_cycles_per_step"] = t["busy_cycles"] / steps
            ent["kernel_us_per_step_in_sq_pass"] = t["sq_ns"] / steps / 1e3
            ent["mfma_busy"] = t["busy_cycles"] / (SIMDS * t["sq_ns"] * 1e-9 * PEAK_CLOCK_HZ)
        out[f] = ent
    kern = {k: dict(r, launches_per_step=r["dispatches"] / steps, family=_family_of(k)) for k, r in krows.items()}
    git, sha = _build_identity()
    with open(out_path, "w") as f:
        json.dump({"note": "rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + "
                           "SQ_INSTS_VALU_MFMA_MOPS_F32: separate runs, --kernel-trace only); FETCH_SIZE doubled (gfx950 counts "
                           "128-B requests as 64 B, MI355X_MICROARCH.md HBM section); executed flops = MOPS_F32 x 512; mfma_busy "
                           "= busy cycles / (1024 SIMDs x the kernels' own durations in the same pass x 2.4 GHz)",
                   "git": git, "csrc_sha16": sha, "workload": workload, "dgrad": dgrad,
                   "steps_profiled": steps, "families": out, "kernels": kern}, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        # --json <steps> <out.json> [--workload w] [--dgrad m] <csv> ...
        rest = sys.argv[4:]
        kw = {}
        while rest and rest[0] in ("--workload", "--dgrad"):
            kw[rest[0][2:]] = rest[1]
            rest = rest[2:]
        family_json(rest, int(sys.argv[2]), sys.argv[3], **kw)
    else:
        main(sys.argv[1:])
