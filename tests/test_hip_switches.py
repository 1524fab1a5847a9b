"""GPU: every switch of DESIGN §10 (environment variables read once per process: the unfused compositions behind the fused
nodes, stand-alone behind grouped launches) against the DEFAULT path: one training step of the two convolutional
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

# Round 5: every switch whose variant DESIGN records as slower is gone together with the code only it reached; what is left
# selects a composition the front end also reaches by itself for other models (unfused module calls, stand-alone launches,
# the packed exchange bucket) or is a diagnostic of this round (weight gradients on a second stream / on CU-masked streams).
OVERWRITE = [
    ("DF_LAZY_MODULES", "0"), ("DF_WGRAD_GROUP", "0"), ("DF_BN_BWD_FUSED", "0"), ("DFHIP_BN_BWD_FUSED", "0"),
    ("DF_GRAD_IN_BUCKET", "0"), ("DFHIP_WGRAD_GROUP_DIV", "4"), ("DFHIP_WGRAD_SIDE", "6"),
    # (round 6: head + loss + the head's backward as the three launches of rounds 1-5)
    ("DF_HEAD_CE", "0"),
    # (round 6: Dropout, the classifier behind it and the loss as three launches each way -- the CNN job)
    ("DF_DROPLIN_CE", "0"),
    # (round 5's diagnostic: the backward chain on 128 CUs, every four queued weight gradients on the other 128)
    (("DFHIP_OVERLAP_CUS", "128"), ("DFHIP_WGRAD_SIDE", "4")),
]
ACCUMULATE = [("DF_LAZY_MODULES", "0"), ("DF_WGRAD_GROUP", "0")]


def _child(tmp, tag, mode, env_extra):
    out = os.path.join(tmp, tag + ".npz")
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_child.py"), out, mode], env=env,
                       capture_output=True, text=True, timeout=600)
    return tag, r.returncode, r.stderr[-1500:], out


def _run_matrix(tmp_path, mode, switches):
    tmp = str(tmp_path)
    # (a switch is (name, value) or a tuple of such pairs that belong together)
    def env_of(sw):
        return dict(sw) if isinstance(sw[0], tuple) else {sw[0]: sw[1]}
    jobs = [("default", {})] + [(" ".join(f"{k}={v}" for k, v in env_of(sw).items()), env_of(sw)) for sw in switches]
    with ThreadPoolExecutor(max_workers=4) as pool:
        done = list(pool.map(lambda j: _child(tmp, j[0].replace("=", "_").replace(" ", "+"), mode, j[1]), jobs))
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
