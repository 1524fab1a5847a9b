"""Child process of tests/test_hip_switches.py: one training step of the two convolutional benchmark jobs on the hip device
under whatever tuning / diagnostic switches the environment carries, results to an .npz (argv[1]).  argv[2] = dgrad mode."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from deepflows_amd import backend_api
    from deepflows_amd.tensor import Graph
    out, mode = sys.argv[1], sys.argv[2]
    dev = backend_api.Device("hip")
    dev.mod.conv_dgrad_mode = mode
    res = {}
    for wl, batch in (("resnet18", 256), ("cnn", 256)):
        model, opt, crit, xt, yt = bench.build_job(wl, dev, 0, batch)
        outputs = model(xt)
        loss = crit(outputs, yt)
        opt.zero_grad()
        loss.backward()
        params = list(opt.params)
        res[wl + "_loss"] = loss.numpy().astype(np.float64)
        res[wl + "_logits"] = outputs.numpy().astype(np.float64)
        res[wl + "_grad_abs_sum"] = np.array([float(np.abs(p.grad.numpy().astype(np.float64)).sum()) for p in params])
        res[wl + "_first_wgrad"] = params[0].grad.numpy().astype(np.float64)
        opt.step()
        res[wl + "_param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in params])
        Graph.free_graph_all()
    np.savez(out, **res)


if __name__ == "__main__":
    main()
