"""Summarise a rocprofv3 --pmc CSV: per kernel name, dispatches and mean counter value.
usage: python tools/pmc_summary.py <counter_collection.csv> [<more.csv> ...]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads (MI355X_MICROARCH.md, HBM section) — the x2-corrected figure is printed beside it."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    name = name.split("(")[0]
    return name if len(name) < 70 else name[:67] + "..."


def main(paths):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                a = agg[k][c]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    rows = []
    for k, cs in agg.items():
        n = max(v[0] for v in cs.values())
        fe = cs.get("FETCH_SIZE", [0, 0.0])
        wr = cs.get("WRITE_SIZE", [0, 0.0])
        rows.append((fe[1] + wr[1], k, n, fe, wr))
    rows.sort(reverse=True)
    print(f"{'kernel':70s} {'calls':>6s} {'FETCH KiB/call':>15s} {'x2 corrected MB':>16s} {'WRITE KiB/call':>15s} {'WRITE MB':>9s}")
    for _, k, n, fe, wr in rows:
        f = fe[1] / fe[0] if fe[0] else float("nan")
        w = wr[1] / wr[0] if wr[0] else float("nan")
        print(f"{k:70s} {n:6d} {f:15.1f} {2 * f * 1024 / 1e6:16.3f} {w:15.1f} {w * 1024 / 1e6:9.3f}")


FAMILIES = {
    # kernel-name prefix (a trailing $ = exact name) -> the family bench.py reports (dfhip ProfScope names);
    # first match wins
    "wgrad_slab_sum": ("k_wgrad_group_sum",),
    "conv2d_wgrad_grouped": ("k_wgrad_group",),
    "conv2d_wgrad_igemm": ("k_conv_wgrad_rows", "k_conv_wgrad_v3", "k_conv_wgrad_tap", "k_conv_wgrad<", "k_wgrad_tap_sum",
                           "k_flat_slab_sum", "k_wgrad_stem3"),
    "conv2d_fwd_igemm": ("k_conv3x3_rowmma", "k_conv3x3s2_pair", "k_conv3x3s2_img4_pair", "k_conv_igemm_v4<3, 0", "k_conv_igemm_v4<1, 0", "k_conv_igemm$", "k_conv_rows3", "k_conv1x1s2_fwd", "k_conv3x3_img2_fwd", "k_conv3x3_img2_all", "k_conv3x3_img4_fwd"),
    "conv2d_dgrad_igemm": ("k_conv_igemm_v4<3, 1", "k_conv_igemm_v4<1, 1"),
    "conv2d_dgrad_overwrite_igemm": ("k_conv_igemm_v4<3, 2", "k_conv_igemm_v4<1, 2", "k_dgrad_ow_s1", "k_dgrad_ow_img"),
    "bn2d_fwd": ("k_bn_apply_partials", "k_bn_finalize_partials", "k_bn_stats", "k_bn_apply"),
    "pool_linear_fwd": ("k_pool_linear_fwd",),
    "pool_linear_bwd": ("k_pool_linear_bwd",),
    "softmax_ce_fwd": ("k_ce_small", "k_ce_rows", "k_ce_finish"),
    "adam_step": ("k_adam",),
    "bn_relu_pool_fwd": ("k_bn_relu_pool_fwd",),
    "bn_relu_pool_bwd": ("k_bn_relu_pool_bwd",),
    "bn2d_bwd": ("k_bn_fused_bwd", "k_bn_bwd_stats", "k_bn_bwd_apply"),
}


def _family_of(kernel):
    k = short(kernel)
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
    return git or os.environ.get("DF_GIT_HEAD", "?"), h.hexdigest()[:16]


def family_json(paths, steps, out_path, workload="resnet18", dgrad="overwrite"):
    """HBM bytes per step of every family: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; and, when
    the CSVs hold an SQ pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE), the MFMA-busy fraction of the family:
    matrix-pipe busy cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
    import json
    tot = defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "launches": 0, "mfma_busy": 0.0, "gui": 0.0,
                               "mops": 0.0, "sq": 0})
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                fam = _family_of(row["Kernel_Name"])
                if fam is None:
                    continue
                c, v = row["Counter_Name"], float(row["Counter_Value"])
                t = tot[fam]
                if c == "FETCH_SIZE":
                    t["fetch_kib"] += v
                    t["launches"] += 1
                elif c == "WRITE_SIZE":
                    t["write_kib"] += v
                elif c == "SQ_VALU_MFMA_BUSY_CYCLES":
                    t["mfma_busy"] += v
                    t["sq"] += 1
                elif c == "GRBM_GUI_ACTIVE":
                    t["gui"] += v
                elif c == "SQ_INSTS_VALU_MFMA_MOPS_F32":
                    t["mops"] += v
    out = {}
    for fam, t in tot.items():
        ent = {"hbm_bytes_per_step": (2.0 * t["fetch_kib"] + t["write_kib"]) * 1024.0 / steps,
               "fetch_kib_per_step_raw": t["fetch_kib"] / steps, "write_kib_per_step": t["write_kib"] / steps,
               "kernel_launches_per_step": max(t["launches"], t["sq"]) / steps}
        if t["sq"] and t["gui"] > 0:
            ent["mfma_busy_fraction"] = t["mfma_busy"] / (1024.0 * t["gui"] / 8.0)
            ent["mfma_busy_cycles_per_step"] = t["mfma_busy"] / steps
            ent["kernel_cycles_per_step"] = t["gui"] / 8.0 / steps
            ent["mfma_f32_mops_per_step"] = t["mops"] / steps
        out[fam] = ent
    git, sha = _build_identity()
    with open(out_path, "w") as f:
        json.dump({"note": "rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + "
                           "GRBM_GUI_ACTIVE: separate runs, --kernel-trace only); FETCH_SIZE doubled (gfx950 counts "
                           "128-B requests as 64 B, MI355X_MICROARCH.md HBM section); mfma_busy_fraction = matrix-pipe "
                           "busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
                   "git": git, "csrc_sha16": sha, "workload": workload, "dgrad": dgrad,
                   "steps_profiled": steps, "families": out}, f, indent=1)
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
