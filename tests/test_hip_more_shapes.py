"""GPU: shapes outside the benchmarked models that reach kernel instantiations nothing else launched (round 5:
tools/kernel_coverage.py lists them) -- the cross-entropy kernels for more than 16 classes and for batches beyond the
one-workgroup form, the 128-wide GEMM tiles, the tiled 2-D transposition behind `compact` -- against NumPy."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def _arr(mod, a):
    h = mod.Array(a.size)
    mod.from_numpy(np.ascontiguousarray(a, dtype=np.float32), h)
    return h


def _back(mod, h, shape):
    return mod.to_numpy(h, (h.size,), (1,), 0).reshape(shape)


@pytest.mark.parametrize("rows,cols", [(64, 20), (256, 32), (5000, 10), (300, 100), (4097, 7)])
def test_cross_entropy_beyond_the_small_kernel(hip_device, rows, cols):
    """nn/functional.py:104-115 through F.cross_entropy: loss and d loss / d logits for class counts 17 .. 32 (k_ce_small<., 32>),
    more than 32 classes or more than 4 096 rows (k_ce_rows + k_ce_finish, gradient through the primitive path)"""
    from deepflows_amd import nn
    from deepflows_amd.tensor import Graph, Tensor
    rng = np.random.default_rng(rows + cols)
    logits = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
    target = R.smooth_onehot(rng.integers(0, cols, rows), cols, 0.05).astype(np.float32)
    xt = Tensor(logits, device=hip_device, requires_grad=True)
    loss = nn.functional.cross_entropy(xt, Tensor(target, device=hip_device))
    loss.backward()
    want = R.cross_entropy(logits.astype(np.float64), target.astype(np.float64))
    z = logits.astype(np.float64) - logits.astype(np.float64).max(axis=1, keepdims=True)
    sm = np.exp(z) / np.exp(z).sum(axis=1, keepdims=True)
    dwant = (sm * target.astype(np.float64).sum(axis=1, keepdims=True) - target) / rows
    assert_close(loss.numpy().reshape(-1), np.asarray(want).reshape(-1), 1e-5, f"cross entropy {rows} x {cols}")
    assert_close(xt.grad.numpy(), dwant, 1e-4, f"cross entropy gradient {rows} x {cols}")
    Graph.free_graph_all()


@pytest.mark.parametrize("m,k,p,ta,tb", [(4096, 64, 2048, 0, 0), (4096, 96, 2048, 1, 0), (2048, 64, 4096, 0, 1), (3000, 70, 2100, 1, 1),
                                          (65536, 64, 64, 0, 0), (66000, 48, 40, 1, 0), (65536, 64, 64, 0, 1)])
def test_gemm_on_the_128_wide_tiles(hip_device, m, k, p, ta, tb):
    """dfhip_gemm (backend_tensor.py:612-622 + the transposed operands of tensor.py:699-716): outputs of >= 512 tiles of 128 x 128
    and tall outputs of at most 64 columns take k_gemm<2, 2> / k_gemm<2, 1> (the benchmarked models never do)"""
    mod = hip_device.mod
    rng = np.random.default_rng(m + p)
    a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    b = rng.standard_normal((p, k) if tb else (k, p)).astype(np.float32)
    bias = rng.standard_normal(p).astype(np.float32)
    out = mod.Array(m * p)
    mod.gemm(_arr(mod, a), _arr(mod, b), _arr(mod, bias), out, m, k, p, ta, tb)
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    assert_close(_back(mod, out, (m, p)), want, 1e-5, f"gemm {m} x {k} x {p} ta {ta} tb {tb}")


@pytest.mark.parametrize("rows,cols", [(64, 48), (33, 9), (1000, 257), (32, 8)])
def test_compact_of_a_transposed_matrix(hip_device, rows, cols):
    """BackendTensor.compact of `x.T` (backend_tensor.py:310-318): the tiled transposition instead of the strided gather"""
    from deepflows_amd.tensor import Tensor
    rng = np.random.default_rng(rows * cols)
    x = rng.standard_normal((cols, rows)).astype(np.float32)
    t = Tensor(x, device=hip_device)
    got = t.data.permute((1, 0)).compact().numpy()
    assert np.array_equal(got, x.T)


@pytest.mark.parametrize("geo", [(256, 32, 64, 16, 5, 2), (256, 64, 128, 8, 3, 1), (64, 32, 32, 4, 3, 1)])
def test_bias_gradient_rides_on_the_sweeps_slab_sum(hip_device, cpu_oracle, geo):
    """A convolution WITH a bias inside a backward sweep (the CNN workload's conv2 / conv3, test/CNN_CIFAR10_cuda.py:66-72): the
    bias gradient -- dY summed over images and pixels, tensor.py:462-483 -- is formed by workgroups of the group's slab-sum launch
    (no `reduce_to_shape` launch) when the layer's weight gradient is queued in the group"""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    N, C, O, H, k, pad = geo
    rng = np.random.default_rng(N + C + O)
    x_np = rng.standard_normal((N, C, H, H)).astype(np.float32)
    w_np = (rng.standard_normal((O, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b_np = rng.standard_normal((1, O, 1, 1)).astype(np.float32)
    w2_np = (rng.standard_normal((O, O, 3, 3)) / np.sqrt(O * 9)).astype(np.float32)
    res = {}
    for name, dev in (("hip", hip_device), ("cpu", cpu_oracle)):
        x = Tensor(x_np, device=dev, requires_grad=True)
        conv = nn.Conv2d(C, O, kernel_size=k, padding=pad, device=dev)
        conv.weight.data = Tensor(w_np, device=dev).data
        conv.bias.data = Tensor(b_np, device=dev).data
        conv2 = nn.Conv2d(O, O, kernel_size=3, padding=1, bias=False, device=dev)     # (a second queued layer: a group)
        conv2.weight.data = Tensor(w2_np, device=dev).data
        if name == "hip":
            dev.mod.prof_reset(); dev.mod.prof_enable(True)
        y = conv2(conv(x))
        loss = tensor.sum(y * y) * (1.0 / y.numpy().size)
        loss.backward()
        if name == "hip":
            dev.mod.synchronize(); dev.mod.prof_enable(False)
            fams = {kk: v["calls"] for kk, v in dev.mod.prof_read().items() if v["calls"]}
        res[name] = (conv.bias.grad.numpy().copy(), conv.weight.grad.numpy().copy(), conv2.weight.grad.numpy().copy())
        Graph.free_graph_all()
    for a, b, what in zip(res["hip"], res["cpu"], ("db", "dW", "dW2")):
        assert_close(a, b, 1e-4, what)
    if "conv2d_wgrad_grouped" in fams:                # (the layer went into a group: its bias gradient rode along)
        assert "reduce_to_shape" not in fams, fams
