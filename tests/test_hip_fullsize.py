"""GPU parity at the BASELINE's full sizes (ResNet-18 CIFAR variant, batch 256): every conv
layer shape forward / weight gradient / input gradient (both modes), BatchNorm2d, MaxPool2d and
Adam, against the NumPy oracle (oracle/ref_ops.py), plus size-independent properties."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

N = 256
LAYERS = {"stem": (3, 32, 32, 32, 3, 1, 1), "l1": (32, 16, 16, 32, 3, 1, 1), "l2a": (32, 16, 16, 64, 3, 2, 1),
          "l2": (64, 8, 8, 64, 3, 1, 1), "l2d": (32, 16, 16, 64, 1, 2, 0), "l3a": (64, 8, 8, 128, 3, 2, 1),
          "l3": (128, 4, 4, 128, 3, 1, 1), "l3d": (64, 8, 8, 128, 1, 2, 0), "l4a": (128, 4, 4, 256, 3, 2, 1),
          "l4": (256, 2, 2, 256, 3, 1, 1), "l4d": (128, 4, 4, 256, 1, 2, 0),
          # CNN-CIFAR10 (test/CNN_CIFAR10_cuda.py:61-108): conv1 5x5 3->32 on 32x32, conv2 5x5 32->64 on 16x16, conv3 3x3 64->128 on 8x8
          "cnn1": (3, 32, 32, 32, 5, 1, 2), "cnn2": (32, 16, 16, 64, 5, 1, 2), "cnn3": (64, 8, 8, 128, 3, 1, 1)}


def _arr(mod, a):
    h = mod.Array(a.size)
    mod.from_numpy(a, h)
    return h


def _back(mod, h, shape):
    return mod.to_numpy(h, (h.size,), (1,), 0).reshape(shape)


# Shapes that reach kernel instantiations the benchmarked models do not (tools/kernel_coverage.py, round 5: the whole-image
# forms of the 4 x 4 input gradients that larger batches take, the 64- and 128-channel 4 x 4 / 2 x 2 layers, 512 output
# channels on 2 x 2 images): (batch, C, H, W, O, k, stride, pad)
OTHER = {"l3_n512": (512, 128, 4, 4, 128, 3, 1, 1), "l4a_n512": (512, 128, 4, 4, 256, 3, 2, 1), "c64_4x4": (256, 64, 4, 4, 64, 3, 1, 1),
         "c128_2x2": (64, 128, 2, 2, 128, 3, 1, 1), "c256_512_2x2": (64, 256, 2, 2, 512, 3, 1, 1), "l2_n64": (64, 64, 8, 8, 64, 3, 1, 1),
         "l1_n32": (32, 32, 16, 16, 32, 3, 1, 1)}


@pytest.mark.parametrize("name", list(OTHER))
def test_conv_other_sizes_the_dispatch_reaches(hip_device, name):
    _conv_layer(hip_device, name, OTHER[name][0], OTHER[name][1:])


@pytest.mark.parametrize("name", list(LAYERS))
def test_conv_layer_full_size(hip_device, name):
    _conv_layer(hip_device, name, N, LAYERS[name])   # (every layer of both benchmarked models at the benchmarked batch)


def assert_close_elementwise(got, want, cond, what, tol=1e-5, floor=1e-6):
    """Every element against ITS OWN scale (under max|a-b| / max|b| a wrong value in a small-magnitude
    region -- the border taps of a near-zero gradient plane -- passes).  `cond` is the element's condition, the same
    contraction evaluated on the operands' absolute values (sum_k |a_k| |b_k| >= |want|): an fp32 dot product of K terms in
    any order is within ~sqrt(K) 6e-8 of it, so |got - want| <= 1e-5 * max(cond, floor * max cond) holds for every correct
    evaluation whatever cancels inside the element, and a value that is wrong by as little as 1e-5 of its region's magnitude
    fails -- wherever the region lies (the absolute floor is 1e-6 of the tensor's largest condition)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    cond = np.asarray(cond, np.float64)
    scale = np.maximum(cond, floor * cond.max())
    r = np.abs(got - want) / scale
    i = np.unravel_index(int(np.argmax(r)), r.shape)
    assert r[i] <= tol, f"{what}: element {i}: got {got[i]:.6e} want {want[i]:.6e} (condition {cond[i]:.3e}): {r[i]:.3e} > {tol:g}"


R18_CONVS = ["stem", "l1", "l2a", "l2", "l2d", "l3a", "l3", "l3d", "l4a", "l4", "l4d"]


@pytest.mark.parametrize("name", R18_CONVS)
def test_conv_layer_full_size_elementwise_with_small_magnitude_regions(hip_device, name):
    """The eleven ResNet-18 convolution shapes at batch 256 on operands WITH small-magnitude regions (every fourth image
    x 1e-3, every eighth output channel's filters x 1e-3, the border ring of dY x 1e-3, two source channels x 1e-3), each
    result checked element by element against its own condition number."""
    mod = hip_device.mod
    C, H, W, O, k, s, p = LAYERS[name]
    n = N
    oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    rng = np.random.default_rng(17 + len(name) + C + O)
    x = rng.standard_normal((n, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    dy = rng.standard_normal((n, O, oh, ow)).astype(np.float32)
    small = np.float32(1e-3)
    x[1::4] *= small
    x[:, C // 2:C // 2 + 2] *= small
    w[3::8] *= small
    dy[2::4] *= small
    if oh > 2:
        ring = np.ones((oh, ow), np.float32) * small
        ring[1:-1, 1:-1] = 1
        dy *= ring
    geo = (n, C, H, W, O, k, s, p)
    xh, wh, dyh = _arr(mod, x), _arr(mod, w), _arr(mod, dy)
    ax, aw, ady = np.abs(x), np.abs(w), np.abs(dy)
    yh = mod.Array(n * O * oh * ow)
    mod.conv2d_forward(xh, wh, None, yh, *geo)
    assert_close_elementwise(_back(mod, yh, (n, O, oh, ow)), R.conv2d_forward(x, w, None, s, p),
                             R.conv2d_forward(ax, aw, None, s, p), name + " forward")
    dwh = mod.Array(w.size)
    mod.conv2d_wgrad(xh, dyh, dwh, *geo)
    assert_close_elementwise(_back(mod, dwh, w.shape), R.conv2d_wgrad(x, dy, k, s, p), R.conv2d_wgrad(ax, ady, k, s, p),
                             name + " wgrad")
    dxh = mod.Array(x.size)
    for mode in ("overwrite", "accumulate"):
        mod.conv2d_dgrad(dyh, wh, dxh, *geo, mode=mode)
        assert_close_elementwise(_back(mod, dxh, x.shape), R.conv2d_dgrad(dy, w, x.shape, s, p, mode),
                                 R.conv2d_dgrad(ady, aw, x.shape, s, p, mode), f"{name} dgrad {mode}")


def _conv_layer(hip_device, name, n, layer):
    mod = hip_device.mod
    C, H, W, O, k, s, p = layer
    oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    rng = np.random.default_rng(hash(name) % 1000)
    x = rng.standard_normal((n, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    dy = rng.standard_normal((n, O, oh, ow)).astype(np.float32)
    geo = (n, C, H, W, O, k, s, p)
    xh, wh, dyh = _arr(mod, x), _arr(mod, w), _arr(mod, dy)
    yh = mod.Array(n * O * oh * ow)
    mod.conv2d_forward(xh, wh, None, yh, *geo)
    assert_close(_back(mod, yh, (n, O, oh, ow)), R.conv2d_forward(x, w, None, s, p), 1e-4, name + " forward")
    dwh = mod.Array(w.size)
    mod.conv2d_wgrad(xh, dyh, dwh, *geo)
    assert_close(_back(mod, dwh, w.shape), R.conv2d_wgrad(x, dy, k, s, p), 1e-4, name + " wgrad")
    dxh = mod.Array(x.size)
    for mode in ("overwrite", "accumulate"):
        mod.conv2d_dgrad(dyh, wh, dxh, *geo, mode=mode)
        assert_close(_back(mod, dxh, x.shape), R.conv2d_dgrad(dy, w, x.shape, s, p, mode), 1e-4,
                     f"{name} dgrad {mode}")


def test_conv_linearity_and_bias(hip_device):
    """conv(a*x1 + x2) == a*conv(x1) + conv(x2) (no bias); bias adds per output channel."""
    mod = hip_device.mod
    C, H, W, O, k, s, p = LAYERS["l2"]
    geo = (N, C, H, W, O, k, s, p)
    rng = np.random.default_rng(5)
    x1 = rng.standard_normal((N, C, H, W)).astype(np.float32)
    x2 = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, k, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    out = []
    for xx, bias in ((x1, None), (x2, None), (np.float32(2.0) * x1 + x2, None), (x1, b)):
        yh = mod.Array(N * O * H * W)
        mod.conv2d_forward(_arr(mod, xx), _arr(mod, w), _arr(mod, bias) if bias is not None else None, yh, *geo)
        out.append(_back(mod, yh, (N, O, H, W)))
    assert_close(out[2], np.float32(2.0) * out[0] + out[1], 1e-4, "linearity")
    assert_close(out[3], out[0] + b.reshape(1, O, 1, 1), 1e-6, "bias epilogue")


@pytest.mark.parametrize("shape", [(256, 32, 32, 32), (256, 64, 8, 8), (256, 256, 2, 2),
                                   # 2 x 2 planes, eight channels per workgroup: a ragged batch, the largest batch and the
                                   # smallest channel count that form takes, one image more than it takes
                                   (80, 256, 2, 2), (512, 8, 2, 2), (513, 16, 2, 2)])
def test_batchnorm_full_size(hip_device, shape):
    mod = hip_device.mod
    n, c, h, w = shape
    rng = np.random.default_rng(c)
    x = (rng.standard_normal(shape) * 1.7 + 0.3).astype(np.float32)
    gam = rng.standard_normal((1, c, 1, 1)).astype(np.float32)
    bet = rng.standard_normal((1, c, 1, 1)).astype(np.float32)
    dy = rng.standard_normal(shape).astype(np.float32)
    xh, yh = _arr(mod, x), mod.Array(x.size)
    mean, var = mod.Array(c), mod.Array(c)
    rm, rv = _arr(mod, np.zeros(c, np.float32)), _arr(mod, np.ones(c, np.float32))
    mod.bn2d_forward_train(xh, _arr(mod, gam), _arr(mod, bet), yh, mean, var, rm, rv, n, c, h * w, 1e-5, 0.1)
    y_ref, m_ref, v_ref = R.batchnorm2d_train(x, gam, bet)
    assert_close(_back(mod, yh, shape), y_ref, 1e-4, "bn y")
    assert_close(_back(mod, mean, (1, c, 1, 1)), m_ref, 1e-4, "bn mean")
    assert_close(_back(mod, var, (1, c, 1, 1)), v_ref, 1e-4, "bn var")
    assert_close(_back(mod, rv, (1, c, 1, 1)), 0.9 + 0.1 * v_ref, 1e-4, "running var")
    dx, dg, db = mod.Array(x.size), mod.Array(c), mod.Array(c)
    mod.bn2d_backward(_arr(mod, dy), xh, _arr(mod, gam), mean, var, dx, dg, db, n, c, h * w, 1e-5)
    dx_ref, dg_ref, db_ref = R.batchnorm2d_backward(dy, x, gam, m_ref, v_ref)
    assert_close(_back(mod, dx, shape), dx_ref, 1e-4, "bn dx")
    assert_close(_back(mod, dg, (1, c, 1, 1)), dg_ref, 1e-4, "bn dgamma")
    assert_close(_back(mod, db, (1, c, 1, 1)), db_ref, 1e-4, "bn dbeta")


def test_maxpool_full_size(hip_device):
    mod = hip_device.mod
    rng = np.random.default_rng(3)
    x = rng.integers(-4, 5, size=(256, 32, 32, 32)).astype(np.float32)   # many ties
    dy = rng.standard_normal((256, 32, 16, 16)).astype(np.float32)
    xh, yh, dxh = _arr(mod, x), mod.Array(256 * 32 * 256), mod.Array(x.size)
    mod.maxpool2d_forward(xh, yh, 256 * 32, 32, 32, 2)
    y = _back(mod, yh, (256, 32, 16, 16))
    assert np.array_equal(y, R.maxpool2d_forward(x, 2))
    mod.maxpool2d_backward(xh, yh, _arr(mod, dy), dxh, 256 * 32, 32, 32, 2)
    assert np.array_equal(_back(mod, dxh, x.shape), R.maxpool2d_backward(x, y, dy, 2))


def test_adam_full_parameter_set(hip_device):
    """All 62 ResNet-18 tensors (2 797 610 elements) in one multi-tensor launch, 2 steps."""
    mod = hip_device.mod
    rng = np.random.default_rng(11)
    sizes = [864, 32, 32] + [9216, 32, 32] * 4 + [18432, 64, 64, 2048, 64, 64, 36864, 64, 64] + [36864, 64, 64] * 2 + \
            [73728, 128, 128, 8192, 128, 128] + [147456, 128, 128] * 3 + [294912, 256, 256, 32768, 256, 256] + \
            [589824, 256, 256] * 3 + [2560, 10]
    ps = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    vs = [np.zeros(n, np.float32) for n in sizes]
    ss = [np.zeros(n, np.float32) for n in sizes]
    ph, vh, sh = [_arr(mod, a) for a in ps], [_arr(mod, a) for a in vs], [_arr(mod, a) for a in ss]
    for t in (1, 2):
        gs = [rng.standard_normal(n).astype(np.float32) for n in sizes]
        mod.adam_step(ph, [_arr(mod, g) for g in gs], vh, sh, 1e-3, 0.9, 0.999, 1e-8, 5e-4, t)
        for i in range(len(sizes)):
            ps[i], vs[i], ss[i] = R.adam_step(ps[i], gs[i], vs[i], ss[i], t, 1e-3, wd=5e-4)
    for i in (0, 3, 20, 40, 58, 60, 61):
        assert_close(_back(mod, ph[i], (sizes[i],)), ps[i], 1e-5, f"adam param {i}")


def test_elementwise_checksum_at_scale(hip_device):
    """a + b, a * b and relu on 8.4 M elements (the stem activation): exact against NumPy."""
    mod = hip_device.mod
    rng = np.random.default_rng(1)
    n = 256 * 32 * 32 * 32 + 3              # not a multiple of 4: exercises the scalar tail
    a = rng.standard_normal(n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    ah, bh, oh_ = _arr(mod, a), _arr(mod, b), mod.Array(n)
    mod.ewise_add(ah, bh, oh_)
    assert np.array_equal(_back(mod, oh_, (n,)), a + b)
    mod.ewise_mul(ah, bh, oh_)
    assert np.array_equal(_back(mod, oh_, (n,)), a * b)
    mod.relu_forward(ah, oh_)
    assert np.array_equal(_back(mod, oh_, (n,)), np.maximum(a, 0))
    s = mod.Array(1)
    mod.reduce_sum(ah, s, n)
    assert abs(float(_back(mod, s, (1,))[0]) - float(a.astype(np.float64).sum())) < 1e-4 * np.abs(a).sum()


def test_grouped_wgrad_all_layers(hip_device):
    """dfhip_wgrad_group_begin/_end: the weight gradients of every ResNet-18 layer shape queued and
    run as one launch equal the stand-alone launches (fp32 summation order differs) and the oracle;
    a call that touches a queued dw (here: reading it back) flushes the queue by itself."""
    mod = hip_device.mod
    if not mod.wgrad_grouping:
        pytest.skip("DF_WGRAD_GROUP=0")
    rng = np.random.default_rng(11)
    jobs = []
    for name, (C, H, W, O, k, s, p) in LAYERS.items():
        n = N                                   # (every layer of both benchmarked models at the benchmarked batch)
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = rng.standard_normal((n, C, H, W)).astype(np.float32)
        dy = rng.standard_normal((n, O, oh, ow)).astype(np.float32)
        geo = (n, C, H, W, O, k, s, p)
        xh, dyh = _arr(mod, x), _arr(mod, dy)
        alone = mod.Array(O * C * k * k)
        mod.conv2d_wgrad(xh, dyh, alone, *geo)
        jobs.append((name, x, dy, geo, xh, dyh, _back(mod, alone, (O, C, k, k)).copy()))
    assert mod.wgrad_group_begin()
    outs = []
    for name, x, dy, geo, xh, dyh, _ in jobs:
        dwh = mod.Array(geo[4] * geo[1] * geo[5] * geo[5])
        mod.conv2d_wgrad(xh, dyh, dwh, *geo)
        outs.append(dwh)
    mod.wgrad_group_end()
    for (name, x, dy, geo, xh, dyh, alone), dwh in zip(jobs, outs):
        got = _back(mod, dwh, alone.shape)
        assert_close(got, alone, 2e-5, name + " grouped vs stand-alone")
        if name in ("l1", "l3", "l4d", "l2a"):
            assert_close(got, R.conv2d_wgrad(x, dy, geo[5], geo[6], geo[7]), 1e-4, name + " grouped vs oracle")
    # a second group, flushed implicitly by the read-back of one of its results
    assert mod.wgrad_group_begin()
    name, x, dy, geo, xh, dyh, alone = jobs[1]
    dwh = mod.Array(alone.size)
    mod.conv2d_wgrad(xh, dyh, dwh, *geo)
    got = _back(mod, dwh, alone.shape)              # to_numpy touches dw -> flush
    mod.wgrad_group_end()
    assert_close(got, alone, 2e-5, "implicit flush")


def test_grouped_wgrad_small_batches(hip_device):
    """ragged / small batches through the grouped path (N = 2, 6, 10: fewer groups than slabs)."""
    mod = hip_device.mod
    if not mod.wgrad_grouping:
        pytest.skip("DF_WGRAD_GROUP=0")
    rng = np.random.default_rng(12)
    for n in (2, 6, 10):
        assert mod.wgrad_group_begin()
        held = []
        for name in ("l1", "l2a", "l3", "l4", "l3d"):
            C, H, W, O, k, s, p = LAYERS[name]
            oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            x = rng.standard_normal((n, C, H, W)).astype(np.float32)
            dy = rng.standard_normal((n, O, oh, ow)).astype(np.float32)
            xh, dyh, dwh = _arr(mod, x), _arr(mod, dy), mod.Array(O * C * k * k)
            mod.conv2d_wgrad(xh, dyh, dwh, n, C, H, W, O, k, s, p)
            held.append((name, x, dy, k, s, p, xh, dyh, dwh, (O, C, k, k)))
        mod.wgrad_group_end()
        for name, x, dy, k, s, p, xh, dyh, dwh, shape in held:
            assert_close(_back(mod, dwh, shape), R.conv2d_wgrad(x, dy, k, s, p), 1e-4, f"{name} n={n} grouped")


@pytest.mark.parametrize("n", [256, 48, 40, 24, 6])
def test_grouped_wgrad_with_shortcuts_that_share_their_blocks_input(hip_device, n):
    """In a residual block with a convolution shortcut both the 3x3 / stride-2 convolution and the 1x1 / stride-2 shortcut
    read the block input: queued in one group with the SAME x array, the shortcut's weight gradient rides on the
    convolution's workgroups (a tenth accumulator, wgrad_rows_direct<PAIR>).  Every layer kind of the grouped body in one
    group, against the oracle, at the benchmarked batch and at batches where the units of several image pairs do / do
    not divide the batch (48: all kinds direct; 40, 24: the 8-pair kinds fall back to the staged body; 6: all staged)."""
    mod = hip_device.mod
    if not mod.wgrad_grouping:
        pytest.skip("DF_WGRAD_GROUP=0")
    rng = np.random.default_rng(29)
    assert mod.wgrad_group_begin()
    held = []
    shared = {}
    for name in ("l1", "l2", "l3", "l4", "l2a", "l2d", "l3a", "l3d", "l4a", "l4d"):
        C, H, W, O, k, s, p = LAYERS[name]
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        key = (C, H, W) if s == 2 else name
        if key not in shared:
            x = rng.standard_normal((n, C, H, W)).astype(np.float32)
            shared[key] = (x, _arr(mod, x))
        x, xh = shared[key]
        dy = rng.standard_normal((n, O, oh, ow)).astype(np.float32)
        dyh, dwh = _arr(mod, dy), mod.Array(O * C * k * k)
        mod.conv2d_wgrad(xh, dyh, dwh, n, C, H, W, O, k, s, p)
        held.append((name, x, dy, k, s, p, xh, dyh, dwh, (O, C, k, k)))
    mod.wgrad_group_end()
    for name, x, dy, k, s, p, xh, dyh, dwh, shape in held:
        assert_close(_back(mod, dwh, shape), R.conv2d_wgrad(x, dy, k, s, p), 1e-4, f"{name} n={n} grouped, shared input")


@pytest.mark.parametrize("name,bias", [("l1", False), ("l2", True), ("l3", False), ("l1", True)])
def test_register_operand_forward_with_partials(hip_device, name, bias):
    """3x3 / stride 1 forward on the register-operand kernel (dfhip_conv_rows.hip) at batch 256: the output
    against the oracle, and the BatchNorm partials it leaves behind (count, mean, centred second moment per
    workgroup and channel) combined with Chan's formula against the mean / biased variance of the output."""
    mod = hip_device.mod
    C, H, W, O, k, s, p = LAYERS[name]
    rng = np.random.default_rng(41)
    x = (rng.standard_normal((N, C, H, W)) + 0.25).astype(np.float32)
    w = (rng.standard_normal((O, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32) if bias else None
    yh, ph = mod.Array(N * O * H * W), mod.Array(512 * O * 3)
    rows = mod.conv2d_forward_stats(_arr(mod, x), _arr(mod, w), _arr(mod, b) if bias else None, yh,
                                    N, C, H, W, O, k, s, p, ph, 512)
    y_ref = R.conv2d_forward(x, w, b, s, p)
    assert_close(_back(mod, yh, y_ref.shape), y_ref, 1e-4, name + " forward")
    assert rows > 0
    part = _back(mod, ph, (512, O, 3))[:rows].astype(np.float64)
    cnt, mu, m2 = np.zeros(O), np.zeros(O), np.zeros(O)
    for r in range(rows):
        nb, mb, qb = part[r, :, 0], part[r, :, 1], part[r, :, 2]
        tot = cnt + nb
        d = mb - mu
        mu = mu + d * nb / tot
        m2 = m2 + qb + d * d * cnt * nb / tot
        cnt = tot
    assert np.all(cnt == N * H * W)
    y64 = y_ref.astype(np.float64)
    assert_close(mu, y64.mean(axis=(0, 2, 3)), 1e-5, "partials mean")
    assert_close(m2 / cnt, y64.var(axis=(0, 2, 3)), 1e-5, "partials variance")


@pytest.mark.parametrize("n", [64, 32, 6])
def test_grouped_wgrad_of_32_wide_planes(hip_device, n):
    """The 64-channel 32 x 32 layers of the wide ResNet-18 (test/ResNet.py) and of VGG16 in a weight-gradient group: 3x3 /
    stride 1, the block's 3x3 / stride 2 with its 1x1 / stride-2 shortcut on the same input, beside 16-wide layers --
    32-wide planes exist in the grouped body with operands from global memory only (batch 6: too few workgroups for a
    group, every layer takes the stand-alone paths)."""
    mod = hip_device.mod
    if not mod.wgrad_grouping:
        pytest.skip("DF_WGRAD_GROUP=0")
    rng = np.random.default_rng(31)
    shapes = [(64, 32, 32, 64, 3, 1, 1), (64, 32, 32, 128, 3, 2, 1), (64, 32, 32, 128, 1, 2, 0), (128, 16, 16, 128, 3, 1, 1),
              (32, 32, 32, 64, 3, 1, 1)]
    assert mod.wgrad_group_begin()
    held, shared = [], {}
    for C, H, W, O, k, s, p in shapes:
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        key = (C, H, W, s == 2)
        if key not in shared:
            x = rng.standard_normal((n, C, H, W)).astype(np.float32)
            shared[key] = (x, _arr(mod, x))
        x, xh = shared[key]
        dy = rng.standard_normal((n, O, oh, ow)).astype(np.float32)
        dyh, dwh = _arr(mod, dy), mod.Array(O * C * k * k)
        mod.conv2d_wgrad(xh, dyh, dwh, n, C, H, W, O, k, s, p)
        held.append((x, dy, k, s, p, xh, dyh, dwh, (O, C, k, k)))
    mod.wgrad_group_end()
    for x, dy, k, s, p, xh, dyh, dwh, shape in held:
        assert_close(_back(mod, dwh, shape), R.conv2d_wgrad(x, dy, k, s, p), 1e-4, f"{shape} n={n} grouped, 32-wide planes")


@pytest.mark.parametrize("C,H,O,n,bias", [(64, 32, 128, 64, False), (128, 16, 256, 128, True), (256, 8, 512, 256, False),
                                           (32, 16, 64, 256, True), (64, 8, 96, 256, False)])
def test_shortcut_convolution_with_partials_and_several_channel_tiles(hip_device, C, H, O, n, bias):
    """1x1 / stride 2 forward (k_conv1x1s2_fwd: a workgroup computes several 32-channel tiles from one load of its pixel
    operand; the wide ResNet-18's shapes, the 256-channel form, a channel count that is not a multiple of 32): the output
    against the oracle and the BatchNorm partials of every channel tile combined with Chan's formula."""
    mod = hip_device.mod
    rng = np.random.default_rng(43)
    oh = H // 2
    x = (rng.standard_normal((n, C, H, H)) + 0.25).astype(np.float32)
    w = (rng.standard_normal((O, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32) if bias else None
    cap = 2048
    yh, ph = mod.Array(n * O * oh * oh), mod.Array(cap * O * 3)
    rows = mod.conv2d_forward_stats(_arr(mod, x), _arr(mod, w), _arr(mod, b) if bias else None, yh,
                                    n, C, H, H, O, 1, 2, 0, ph, cap)
    y_ref = R.conv2d_forward(x, w, b, 2, 0)
    assert_close(_back(mod, yh, y_ref.shape), y_ref, 1e-4, "shortcut forward")
    assert rows > 0
    part = _back(mod, ph, (cap, O, 3))[:rows].astype(np.float64)
    cnt, mu, m2 = np.zeros(O), np.zeros(O), np.zeros(O)
    for r in range(rows):
        nb, mb, qb = part[r, :, 0], part[r, :, 1], part[r, :, 2]
        tot = cnt + nb
        d = mb - mu
        mu = mu + d * nb / np.maximum(tot, 1)
        m2 = m2 + qb + d * d * cnt * nb / np.maximum(tot, 1)
        cnt = tot
    assert np.all(cnt == n * oh * oh)
    y64 = y_ref.astype(np.float64)
    assert_close(mu, y64.mean(axis=(0, 2, 3)), 1e-5, "partials mean")
    assert_close(m2 / cnt, y64.var(axis=(0, 2, 3)), 1e-5, "partials variance")


@pytest.mark.parametrize("name", ["l1", "l2", "l3", "l4", "l4_128_to_256", "l4_256_to_128"])
def test_register_operand_accumulate_dgrad_with_held_gradient(hip_device, name):
    """The mathematically correct input gradient of the 3x3 / stride 1 layers on the register-operand kernel (weights
    transposed and flipped while they are staged; 2 x 2 images: the forward's all-positions kernel with the channel counts
    swapped, also where they differ), with and without the gradient the tensor already holds."""
    mod = hip_device.mod
    C, H, W, O, k, s, p = {"l4_128_to_256": (128, 2, 2, 256, 3, 1, 1), "l4_256_to_128": (256, 2, 2, 128, 3, 1, 1)}.get(
        name) or LAYERS[name]
    rng = np.random.default_rng(17)
    dy = rng.standard_normal((N, O, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, k, k)) / np.sqrt(O * k * k)).astype(np.float32)
    held = rng.standard_normal((N, C, H, W)).astype(np.float32)
    want = R.conv2d_dgrad(dy, w, (N, C, H, W), s, p, "accumulate")
    dyh, wh, dxh = _arr(mod, dy), _arr(mod, w), mod.Array(N * C * H * W)
    mod.conv2d_dgrad(dyh, wh, dxh, N, C, H, W, O, k, s, p, mode="accumulate")
    assert_close(_back(mod, dxh, want.shape), want, 1e-4, name + " accumulate dgrad")
    mod.conv2d_dgrad_add(dyh, wh, _arr(mod, held), dxh, N, C, H, W, O, k, s, p, mode="accumulate")
    assert_close(_back(mod, dxh, want.shape), want + held, 1e-4, name + " accumulate dgrad + held gradient")


@pytest.mark.parametrize("name", ["l2a", "l3a", "l4a"])
def test_accumulate_dgrad_of_the_stride2_layers_with_shortcut_and_held_gradient(hip_device, name):
    """The complete input gradient of a block's 3x3 / stride-2 convolution at batch 256 -- alone, with the gradient the tensor
    already holds, and together with the block's 1x1 / stride-2 shortcut (k_dgrad_s2_acc: only the (pixel, tap) pairs that
    exist are multiplied) -- against the oracle's sums."""
    mod = hip_device.mod
    C, H, W, O, k, s, p = LAYERS[name]
    rng = np.random.default_rng(19)
    oh, ow = H // 2, W // 2
    dy = rng.standard_normal((N, O, oh, ow)).astype(np.float32)
    dy2 = rng.standard_normal((N, O, oh, ow)).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) / np.sqrt(O * 9)).astype(np.float32)
    w2 = (rng.standard_normal((O, C, 1, 1)) / np.sqrt(O)).astype(np.float32)
    held = rng.standard_normal((N, C, H, W)).astype(np.float32)
    shape = (N, C, H, W)
    want3 = R.conv2d_dgrad(dy, w, shape, 2, 1, "accumulate")
    want1 = R.conv2d_dgrad(dy2, w2, shape, 2, 0, "accumulate")
    dyh, dy2h, wh, w2h, heldh, dxh = _arr(mod, dy), _arr(mod, dy2), _arr(mod, w), _arr(mod, w2), _arr(mod, held), mod.Array(held.size)
    mod.conv2d_dgrad(dyh, wh, dxh, N, C, H, W, O, 3, 2, 1, mode="accumulate")
    assert_close(_back(mod, dxh, shape), want3, 1e-4, name + " accumulate dgrad")
    mod.conv2d_dgrad_add(dyh, wh, heldh, dxh, N, C, H, W, O, 3, 2, 1, mode="accumulate")
    assert_close(_back(mod, dxh, shape), want3 + held, 1e-4, name + " accumulate dgrad + held gradient")
    assert mod.conv2d_dgrad_pair(dyh, wh, dy2h, w2h, heldh, dxh, N, C, H, W, O, mode="accumulate")
    assert_close(_back(mod, dxh, shape), want3 + want1 + held, 1e-4, name + " accumulate dgrad + shortcut + held gradient")
    assert mod.conv2d_dgrad_pair(dyh, wh, dy2h, w2h, None, dxh, N, C, H, W, O, mode="accumulate")
    assert_close(_back(mod, dxh, shape), want3 + want1, 1e-4, name + " accumulate dgrad + shortcut")
