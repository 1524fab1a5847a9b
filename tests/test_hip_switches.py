"""GPU: every tuning / diagnostic switch of DESIGN §10 (environment variables read once per process: kernel variants that
are compiled in but not taken by default -- the staged kernels behind the lean ones, stand-alone behind grouped launches,
the unfused compositions behind the fused nodes) against the DEFAULT path: one training step of the two convolutional
benchmark jobs (ResNet-18 and CNN, batch 256) per switch in a child process, loss / logits / 62 |grad| sums / the stem's weight
gradient / post-step |param| sums compared with the default child's (which tests/test_hip_whole_step.py pins to the oracle).
A switch only changes which kernels run, i.e. the fp32 summation order: 1e-4 relative."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

OVERWRITE = [
    ("DFHIP_DGRAD_LEAN", "0"), ("DFHIP_DGRAD_IMG2", "0"), ("DFHIP_DGRAD_IMG4", "0"), ("DFHIP_DGRAD_PAIR", "0"),
    ("DFHIP_DGRAD_IMG8", "0"), ("DFHIP_IMG4_TR", "0"), ("DFHIP_IMG4_HALVES", "0"), ("DFHIP_BN_INPUT_COMBINE", "0"), ("DFHIP_STEM_COMBINE_RIDES", "0"), ("DFHIP_IMG2_BN_INPUT", "0"), ("DFHIP_ROWMMA_TS8", "0"), ("DFHIP_BN_BWD_HW4", "0"), ("DFHIP_BN_FWD_HW4", "0"), ("DFHIP_IMG4_COAL", "0"), ("DFHIP_DGRAD_ROW0", "0"), ("DFHIP_DGRAD_ROW0", "1"), ("DFHIP_WGRAD_PAIR", "0"), ("DFHIP_WGRAD_DIRECT", "0"),
    ("DFHIP_WGRAD_GROUP_DIV", "4"), ("DFHIP_WGRAD_ROWS", "0"), ("DFHIP_WGRAD_TAP", "0"), ("DFHIP_WGRAD_DEFER", "0"),
    ("DFHIP_WGRAD_STEM3", "0"), ("DFHIP_WGRAD_KT5", "0"), ("DFHIP_IGEMM_K5", "0"), ("DFHIP_IGEMM_K5_SPEC", "0"),
    ("DFHIP_IMG2_LEAN", "0"), ("DFHIP_SHORTCUT_LEAN", "0"), ("DFHIP_STEM_ROWS", "0"), ("DFHIP_VEC_STORE", "0"),
    ("DFHIP_BN_INPUT", "0"), ("DFHIP_STEM_FUSED_BWD", "0"), ("DF_STEM_NOZ", "0"), ("DFHIP_STEM_ONEPASS", "0"),
    ("DFHIP_XCD_ORDER", "1"), ("DFHIP_BN_BWD_FUSED", "0"), ("DF_BN_BWD_FUSED", "0"), ("DF_BN_BWD_PAIR", "1"),
    ("DF_LAZY_MODULES", "0"), ("DF_LINEAR_RELU", "0"), ("DFHIP_MASK_BYTES", "0"), ("DF_S2_PAIR", "0"),
    ("DF_CONV_BN_STATS", "0"), ("DF_HEAD_FUSED", "0"), ("DF_WGRAD_GROUP", "0"), ("DF_BN_INPUT", "0"),
    ("DF_STEM_FUSED_BWD", "0"), ("DFHIP_DGRAD_BN_NW8", "0"), ("DF_GRAD_IN_BUCKET", "0"),
]
ACCUMULATE = [("DFHIP_DGRAD_IMG2_ACC", "0"), ("DFHIP_DGRAD_S2_ACC", "0"), ("DFHIP_WGRAD_DIRECT", "0")]


def _child(tmp, tag, mode, env_extra):
    out = os.path.join(tmp, tag + ".npz")
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_child.py"), out, mode], env=env,
                       capture_output=True, text=True, timeout=600)
    return tag, r.returncode, r.stderr[-1500:], out


def _run_matrix(tmp_path, mode, switches):
    tmp = str(tmp_path)
    jobs = [("default", {})] + [(f"{k}={v}", {k: v}) for k, v in switches]
    with ThreadPoolExecutor(max_workers=4) as pool:
        done = list(pool.map(lambda j: _child(tmp, j[0].replace("=", "_"), mode, j[1]), jobs))
    failed = [(tag, err) for tag, rc, err, _ in done if rc != 0]
    assert not failed, f"children failed: {failed}"
    base = np.load(done[0][3])
    worst, bad = {}, []
    for (name, _), (tag, _, _, out) in zip(jobs[1:], done[1:]):
        got = np.load(out)
        for k in base.files:
            e = rel_err(got[k], base[k])
            worst[name] = max(worst.get(name, (0.0, ""))[0], e), (k if e >= worst.get(name, (0.0, ""))[0] else worst[name][1])
            # the first layer's weight gradient sits behind ReLU / pooling ties: two CORRECT fp32 evaluations that differ
            # in summation order are up to 1e-3 apart there (tests/test_hip_whole_step.py: the fp32 oracle itself is
            # 8.4e-4 from the float64 evaluation for the CNN's conv1) -- 5e-3 for that quantity, 1e-4 for the rest
            lim = 5e-3 if k.endswith("_first_wgrad") else 1e-4
            if e > lim:
                bad.append((name, k, e))
    print(f"\n{mode}: largest relative difference from the default path per switch")
    for name, (e, k) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
        print(f"  {name:28s} {e:.2e}  ({k})")
    assert not bad, f"switches that change the results: {bad}"


def test_every_switch_computes_what_the_default_path_computes(tmp_path):
    _run_matrix(tmp_path, "overwrite", OVERWRITE)


def test_accumulate_mode_switches(tmp_path):
    _run_matrix(tmp_path, "accumulate", ACCUMULATE)
