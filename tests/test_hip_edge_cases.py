"""GPU: protocol edge cases of the hip device — empty arrays, ragged tails, error classes,
8-dimension limit, views with offsets / negative strides / stride 0, pool reuse."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _arr(mod, a):
    h = mod.Array(a.size)
    mod.from_numpy(a, h)
    return h


def _back(mod, h):
    return mod.to_numpy(h, (h.size,), (1,), 0)


def test_empty_and_ragged_sizes(hip_device):
    mod = hip_device.mod
    e = mod.Array(0)
    mod.fill(e, 1.0)
    mod.ewise_add(e, e, e)
    assert mod.to_numpy(e, (0,), (1,), 0).shape == (0,)
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 5, 63, 64, 65, 255, 257, 1023, 4097):
        a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        out = mod.Array(n)
        mod.ewise_mul(_arr(mod, a), _arr(mod, b), out)
        assert np.array_equal(_back(mod, out), a * b), n
        mod.scalar_maximum(_arr(mod, a), 0.25, out)
        assert np.array_equal(_back(mod, out), np.maximum(a, np.float32(0.25))), n
        s = mod.Array(1)
        mod.reduce_max(_arr(mod, a), s, n)
        assert _back(mod, s)[0] == a.max()


def test_error_classes_match_the_cuda_module(hip_device):
    mod = hip_device.mod
    a, out = mod.Array(4), mod.Array(5)
    with pytest.raises(ValueError):
        mod.ewise_add(a, a, out)                       # size mismatch (.cu:232-236)
    with pytest.raises(ValueError):
        mod.compact(a, mod.Array(1), (1,) * 9, (1,) * 9, 0)     # > 8 dims (.cu:113-118)
    with pytest.raises(ValueError):
        mod.compact(a, mod.Array(3), (2, 2), (2, 1), 0)          # view size != out size
    with pytest.raises(IndexError):
        mod.to_numpy(a, (5,), (1,), 0)                           # view exceeds the array
    with pytest.raises(ValueError):
        mod.from_numpy(np.zeros(3, np.float32), a)
    with pytest.raises(ValueError):
        mod.matmul(a, a, out, 2, 2, 2)
    with pytest.raises(ValueError):
        mod.conv2d_forward(a, a, None, a, 1, 1, 2, 2, 1, 5, 1, 0)   # kernel larger than input


def test_numpy_value_semantics(hip_device):
    mod = hip_device.mod
    a = _arr(mod, np.array([1.0, -2.0, 0.0, np.inf], np.float32))
    out = mod.Array(4)
    mod.scalar_div(a, 0.0, out)                                  # NumPy device: inf/nan, no exception
    got = _back(mod, out)
    assert np.isposinf(got[0]) and np.isneginf(got[1]) and np.isnan(got[2])
    mod.ewise_maximum(a, _arr(mod, np.array([np.nan, 1.0, 0.0, 0.0], np.float32)), out)
    got = _back(mod, out)
    assert np.isnan(got[0]) and got[1] == 1.0                    # np.maximum propagates NaN
    mod.scalar_power(a, 2.0, out)
    assert np.array_equal(_back(mod, out)[:3], np.array([1.0, 4.0, 0.0], np.float32))


def test_views_offsets_negative_and_zero_strides(hip_device):
    mod = hip_device.mod
    base = np.arange(128, dtype=np.float32)
    h = _arr(mod, base)
    cases = [((2, 3, 4), (60, 20, 5), 3), ((5, 4), (-24, 1), 100), ((3, 2, 2, 5), (0, 60, 0, 1), 7),
             ((1, 1, 6), (0, 0, -2), 20), ((2, 2, 2, 2, 2, 2, 2), (64, 32, 16, 8, 4, 2, 1), 0)]
    for shp, st, off in cases:
        want = np.lib.stride_tricks.as_strided(base[off:] if all(s >= 0 for s in st) else base,
                                               shape=shp, strides=tuple(4 * s for s in st)) \
            if all(s >= 0 for s in st) else None
        if want is None:
            idx = np.indices(shp)
            want = base[off + sum(idx[d] * st[d] for d in range(len(shp)))]
        out = mod.Array(int(np.prod(shp)))
        mod.compact(h, out, shp, st, off)
        assert np.array_equal(_back(mod, out).reshape(shp), want), (shp, st, off)
        assert np.array_equal(mod.to_numpy(h, shp, st, off), want)
    # np.int64 extents (they appear after permute(), backend_tensor.py:405-406)
    out = mod.Array(12)
    mod.compact(h, out, (np.int64(3), np.int64(4)), (np.int64(1), np.int64(3)), np.int64(0))
    assert np.array_equal(_back(mod, out).reshape(3, 4), base[:12].reshape(4, 3).T)


def test_pool_reuses_blocks_without_new_hipmalloc(hip_device):
    mod = hip_device.mod
    for _ in range(3):
        hs = [mod.Array(1000 + i) for i in range(50)]
        del hs
    before = mod.pool_stats()["hipmalloc_calls"]
    for _ in range(5):
        hs = [mod.Array(1000 + i) for i in range(50)]
        del hs
    assert mod.pool_stats()["hipmalloc_calls"] == before


def test_transposed_gemm_operands_and_bias(hip_device):
    mod = hip_device.mod
    rng = np.random.default_rng(2)
    for (m, k, p) in ((128, 784, 256), (256, 256, 10), (37, 19, 5), (1, 300, 1), (300, 1, 7),
                      (70, 65, 33), (5, 1000, 3), (33, 4096, 2)):   # the last three: single-launch small-GEMM kernel
        A = rng.standard_normal((m, k)).astype(np.float32)
        B = rng.standard_normal((k, p)).astype(np.float32)
        bias = rng.standard_normal(p).astype(np.float32)
        want = A @ B
        for ta in (False, True):
            for tb in (False, True):
                a = _arr(mod, np.ascontiguousarray(A.T if ta else A))
                b = _arr(mod, np.ascontiguousarray(B.T if tb else B))
                out = mod.Array(m * p)
                mod.gemm(a, b, None, out, m, k, p, ta, tb)
                got = _back(mod, out).reshape(m, p)
                assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (m, k, p, ta, tb)
        out = mod.Array(m * p)
        mod.gemm(_arr(mod, A), _arr(mod, B), _arr(mod, bias), out, m, k, p, False, False)
        assert np.abs(_back(mod, out).reshape(m, p) - (want + bias)).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_split_k_gemm_long_reduction(hip_device):
    """im2col-style A^T.dC: small output, 65 536-long reduction (primitive-path wgrad)."""
    mod = hip_device.mod
    rng = np.random.default_rng(4)
    A = rng.standard_normal((65536, 27)).astype(np.float32)
    D = rng.standard_normal((65536, 32)).astype(np.float32)
    out = mod.Array(27 * 32)
    mod.gemm(_arr(mod, A), _arr(mod, D), None, out, 27, 65536, 32, True, False)
    want = A.astype(np.float64).T @ D.astype(np.float64)
    assert np.abs(_back(mod, out).reshape(27, 32) - want).max() <= 1e-4 * np.abs(want).max()


def test_deferred_slab_sum_is_invisible(hip_device):
    """The slab sum of a weight gradient is deferred onto the next weight-gradient launch; every
    way of looking at dW in between must still see the finished gradient."""
    from oracle import ref_ops
    mod = hip_device.mod
    rng = np.random.default_rng(21)
    shapes = [(8, 32, 16, 16, 32, 3, 1, 1), (8, 64, 8, 8, 64, 3, 1, 1), (8, 32, 16, 16, 64, 1, 2, 0)]
    want, arrs = [], []
    for (N, C, H, W, O, k, s, p) in shapes:
        oh = (H + 2 * p - k) // s + 1
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        dy = rng.standard_normal((N, O, oh, oh)).astype(np.float32)
        want.append(ref_ops.conv2d_wgrad(x, dy, k, s, p))
        arrs.append((_arr(mod, x), _arr(mod, dy), mod.Array(O * C * k * k), (N, C, H, W, O, k, s, p)))
    # back to back (each launch takes the previous sum along), then read in reverse order
    for xa, da, dw, geo in arrs:
        mod.conv2d_wgrad(xa, da, dw, *geo)
    for (xa, da, dw, geo), w in reversed(list(zip(arrs, want))):
        got = _back(mod, dw).reshape(w.shape)
        assert np.abs(got - w).max() <= 1e-4 * np.abs(w).max()
    # a consumer on the device right after the launch (ewise on dW), and free + reuse of dW
    xa, da, dw, geo = arrs[0]
    mod.conv2d_wgrad(xa, da, dw, *geo)
    out = mod.Array(dw.size)
    mod.scalar_mul(dw, 2.0, out)
    assert np.abs(_back(mod, out).reshape(want[0].shape) - 2 * want[0]).max() <= 2e-4 * np.abs(want[0]).max()       # (1e-4 of the doubled tensor)
    mod.conv2d_wgrad(xa, da, dw, *geo)
    del dw
    z = mod.Array(want[0].size)
    mod.fill(z, 0.0)
    assert not _back(mod, z).any()


def test_stem_backward_inside_a_weight_gradient_group(hip_device):
    """Inside a weight-gradient group the one-pass stem backward leaves its last step (totals over the slabs -> dw, dgamma,
    dbeta) to the group's slab-sum launch (dfhip_stem_combine.h).  Same bits as outside a group, and a device consumer of
    one of the pending arrays before the group ends sees the finished values."""
    mod = hip_device.mod
    rng = np.random.default_rng(77)
    N, C, H, W, O = 32, 3, 32, 32, 32
    x = _arr(mod, rng.standard_normal((N, C, H, W)).astype(np.float32))
    w = _arr(mod, (0.2 * rng.standard_normal((O, C, 3, 3))).astype(np.float32))
    gamma = _arr(mod, (1.0 + 0.1 * rng.standard_normal(O)).astype(np.float32))
    beta = _arr(mod, (0.1 * rng.standard_normal(O)).astype(np.float32))
    y, mean, var = mod.Array(N * O * 16 * 16), mod.Array(O), mod.Array(O)
    assert mod.stem_forward_noz(x, w, None, gamma, beta, y, mean, var, None, None, N, C, H, W, O, 3, 1, 1, 1e-5, 0.1)
    g = _arr(mod, rng.standard_normal((N, O, 16, 16)).astype(np.float32))
    # a queued weight gradient for the group to carry
    xa = _arr(mod, rng.standard_normal((N, 32, 16, 16)).astype(np.float32))
    da = _arr(mod, rng.standard_normal((N, 32, 16, 16)).astype(np.float32))
    geo = (N, 32, 16, 16, 32, 3, 1, 1)

    def run(grouped, touch):
        dw, dgamma, dbeta, dwq = mod.Array(O * C * 9), mod.Array(O), mod.Array(O), mod.Array(32 * 32 * 9)
        doubled = None
        if grouped:
            mod.wgrad_group_begin()
        mod.conv2d_wgrad(xa, da, dwq, *geo)
        assert mod.stem_backward_noz(x, w, None, y, g, gamma, beta, mean, var, dw, None, dgamma, dbeta, N, C, H, W, O, 3, 1, 1, 1e-5)
        if touch:                                   # a consumer of a pending array in front of the group's end
            doubled = mod.Array(O)
            mod.scalar_mul(dgamma, 2.0, doubled)
        if grouped:
            mod.wgrad_group_end()
        out = [_back(mod, a).copy() for a in (dw, dgamma, dbeta, dwq)]
        return out, (_back(mod, doubled).copy() if doubled is not None else None)

    ref, _ = run(False, False)
    for touch in (False, True):
        got, doubled = run(True, touch)
        for a, b, name in zip(got, ref, ("dw", "dgamma", "dbeta", "queued dw")):
            if name == "queued dw":                 # (a grouped launch plans its workgroups differently: another summation order)
                assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max(), name
            else:
                assert np.array_equal(a, b), f"{name} differs inside a group (touch={touch})"
        if touch:
            assert np.array_equal(doubled, 2.0 * ref[1])


def test_dgrad_add_epilogue(hip_device):
    from oracle import ref_ops
    mod = hip_device.mod
    rng = np.random.default_rng(22)
    for (N, C, H, W, O, k, s, p) in [(8, 32, 16, 16, 32, 3, 1, 1), (6, 64, 8, 8, 128, 3, 2, 1), (4, 3, 12, 12, 8, 3, 1, 1),
                                     (5, 32, 16, 16, 64, 1, 2, 0)]:
        oh = (H + 2 * p - k) // s + 1
        dy = rng.standard_normal((N, O, oh, oh)).astype(np.float32)
        w = rng.standard_normal((O, C, k, k)).astype(np.float32)
        add = rng.standard_normal((N, C, H, W)).astype(np.float32)
        for mode in ("overwrite", "accumulate"):
            want = ref_ops.conv2d_dgrad(dy, w, (N, C, H, W), s, p, mode) + add
            dx = mod.Array(add.size)
            mod.conv2d_dgrad_add(_arr(mod, dy), _arr(mod, w), _arr(mod, add), dx, N, C, H, W, O, k, s, p, mode=mode)
            got = _back(mod, dx).reshape(want.shape)
            assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (N, C, H, W, O, k, s, p, mode)


@pytest.mark.parametrize("C,H,O,N,with_bias", [(1, 32, 8, 3, False), (2, 5, 40, 2, True), (3, 32, 32, 5, True),
                                               (4, 7, 33, 1, False),
                                               # more rows than waves: a wave walks 2-3 rows (read-ahead loop), ragged end
                                               (3, 31, 40, 70, True), (3, 32, 32, 160, False)])
def test_few_channel_stem_kernel(hip_device, C, H, O, N, with_bias):
    """k_conv_rows3: 3x3 / stride 1 / pad 1 on 32-wide images with 1-4 source channels (the ResNet stem),
    ragged output-channel tiles, odd heights, bias, batches where a wave walks several rows."""
    from conftest import assert_close
    from oracle import ref_ops as R
    mod = hip_device.mod
    rng = np.random.default_rng(100 + C)
    x = rng.standard_normal((N, C, H, 32)).astype(np.float32)
    w = rng.standard_normal((O, C, 3, 3)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32) if with_bias else None
    xh, wh = mod.Array(x.size), mod.Array(w.size)
    mod.from_numpy(x, xh)
    mod.from_numpy(w, wh)
    bh = None
    if with_bias:
        bh = mod.Array(O)
        mod.from_numpy(b, bh)
    yh = mod.Array(N * O * H * 32)
    mod.conv2d_forward(xh, wh, bh, yh, N, C, H, 32, O, 3, 1, 1)
    got = mod.to_numpy(yh, (yh.size,), (1,), 0).reshape(N, O, H, 32)
    want = R.conv2d_forward(x, w, b.reshape(1, O, 1, 1) if with_bias else None, 1, 1)
    assert_close(got, want, 1e-4, f"stem kernel C={C} H={H} O={O}")


@pytest.mark.parametrize("C,H,O,N", [(3, 32, 32, 40), (2, 40, 16, 9), (3, 64, 8, 300), (1, 6, 5, 2), (3, 32, 20, 530)])
def test_stem_kernel_partials_bands_and_walking_workgroups(hip_device, C, H, O, N):
    """k_conv_rows3 with its BatchNorm partials: images higher than one band (H > 32: two bands with a shared halo row),
    more bands than workgroups (a workgroup re-stages its tile and keeps its shifted sums across items), a band of
    fewer rows than waves, ragged channel tiles; the partial rows combined with Chan's formula against the output's
    mean / biased variance, with a channel mean far from zero (bias 3.0: the shifted sums must not cancel)."""
    from conftest import assert_close
    from oracle import ref_ops as R
    mod = hip_device.mod
    rng = np.random.default_rng(7 * C + H + N)
    x = rng.standard_normal((N, C, H, 32)).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) * 0.3).astype(np.float32)
    b = np.full(O, 3.0, np.float32)
    yh, ph = mod.Array(N * O * H * 32), mod.Array(512 * O * 3)
    rows = mod.conv2d_forward_stats(_arr(mod, x), _arr(mod, w), _arr(mod, b), yh, N, C, H, 32, O, 3, 1, 1, ph, 512)
    y_ref = R.conv2d_forward(x, w, b.reshape(1, O, 1, 1), 1, 1)
    assert_close(_back(mod, yh).reshape(y_ref.shape), y_ref, 1e-4, "stem forward")
    assert 0 < rows <= 512
    part = _back(mod, ph).reshape(512, O, 3)[:rows].astype(np.float64)
    cnt, mu, m2 = np.zeros(O), np.zeros(O), np.zeros(O)
    for r in range(rows):
        nb, mb, qb = part[r, :, 0], part[r, :, 1], part[r, :, 2]
        tot = cnt + nb
        d = mb - mu
        mu = mu + d * nb / np.maximum(tot, 1)
        m2 = m2 + qb + d * d * cnt * nb / np.maximum(tot, 1)
        cnt = tot
    assert np.all(cnt == N * H * 32)
    y64 = y_ref.astype(np.float64)
    assert_close(mu, y64.mean(axis=(0, 2, 3)), 1e-5, "partials mean")
    assert_close(m2 / cnt, y64.var(axis=(0, 2, 3)), 2e-5, "partials variance")


@pytest.mark.parametrize("N,C,HW,O,with_bias", [(256, 128, 4, 48, True), (128, 128, 4, 128, False), (64, 256, 2, 80, True),
                                                (32, 128, 2, 256, True)])
def test_image_per_lane_forward_kernels(hip_device, N, C, HW, O, with_bias):
    """k_conv3x3_img4_fwd / k_conv3x3_img2_all (3x3 / stride 1 / pad 1 on 4 x 4 and 2 x 2 images, an image on the lane,
    padding taps not multiplied): output against the oracle with a bias and ragged channel-tile counts, and the
    BatchNorm partials (count, mean, centred second moment per workgroup and channel) against the output's statistics."""
    from conftest import assert_close
    from oracle import ref_ops as R
    mod = hip_device.mod
    rng = np.random.default_rng(N + C + O)
    x = (rng.standard_normal((N, C, HW, HW)) + 0.5).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32) if with_bias else None
    yh, ph = mod.Array(N * O * HW * HW), mod.Array(512 * O * 3)
    rows = mod.conv2d_forward_stats(_arr(mod, x), _arr(mod, w), _arr(mod, b) if with_bias else None, yh,
                                    N, C, HW, HW, O, 3, 1, 1, ph, 512)
    y_ref = R.conv2d_forward(x, w, b.reshape(1, O, 1, 1) if with_bias else None, 1, 1)
    assert_close(_back(mod, yh).reshape(y_ref.shape), y_ref, 1e-4, "image-per-lane forward")
    assert rows > 0
    part = _back(mod, ph).reshape(512, O, 3)[:rows].astype(np.float64)
    cnt = part[:, :, 0].sum(axis=0)
    mu = (part[:, :, 0] * part[:, :, 1]).sum(axis=0) / cnt
    m2 = (part[:, :, 2] + part[:, :, 0] * (part[:, :, 1] - mu) ** 2).sum(axis=0)
    assert np.all(cnt == N * HW * HW)
    y64 = y_ref.astype(np.float64)
    assert_close(mu, y64.mean(axis=(0, 2, 3)), 1e-5, "partials mean")
    assert_close(m2 / cnt, y64.var(axis=(0, 2, 3)), 2e-5, "partials variance")


@pytest.mark.parametrize("C,H,O,N", [(3, 32, 32, 16), (1, 8, 5, 3), (2, 2, 32, 300), (3, 30, 17, 7)])
def test_stem_weight_gradient_kernel(hip_device, C, H, O, N):
    """k_wgrad_stem3 stand-alone (dY given): fewer images than CUs, more images than workgroups (a workgroup walks
    several), fewer row pairs than waves, ragged output channels and (channel, tap) columns."""
    from conftest import assert_close
    from oracle import ref_ops as R
    mod = hip_device.mod
    rng = np.random.default_rng(31 + C + H)
    x = rng.standard_normal((N, C, H, 32)).astype(np.float32)
    dy = rng.standard_normal((N, O, H, 32)).astype(np.float32)
    dw = mod.Array(O * C * 9)
    mod.conv2d_wgrad(_arr(mod, x), _arr(mod, dy), dw, N, C, H, 32, O, 3, 1, 1)
    want = R.conv2d_wgrad(x, dy, 3, 1, 1)
    assert_close(_back(mod, dw).reshape(want.shape), want, 1e-4, f"stem wgrad C={C} H={H} O={O} N={N}")


@pytest.mark.parametrize("N,C,O,H,W,k,s,p", [
    (3, 32, 32, 4, 8, 3, 1, 1), (2, 64, 64, 8, 8, 3, 1, 1), (1, 96, 32, 2, 16, 3, 1, 1), (5, 32, 64, 16, 16, 3, 1, 1),
    (2, 32, 32, 32, 32, 3, 1, 1), (8, 128, 128, 4, 4, 3, 1, 1), (16, 64, 256, 2, 2, 3, 1, 1), (6, 32, 128, 4, 4, 3, 1, 1),
    (4, 32, 32, 2, 4, 3, 1, 1),
    (16, 128, 128, 4, 4, 3, 1, 1), (32, 48, 64, 4, 4, 3, 1, 1), (16, 128, 256, 4, 4, 3, 2, 1), (32, 48, 128, 4, 4, 3, 2, 1),
    (2, 32, 64, 16, 16, 3, 2, 1), (4, 64, 128, 8, 8, 3, 2, 1), (8, 128, 256, 4, 4, 3, 2, 1), (2, 32, 32, 4, 8, 3, 2, 1),
    (2, 32, 64, 16, 16, 1, 2, 0), (4, 64, 128, 8, 8, 1, 2, 0), (8, 128, 256, 4, 4, 1, 2, 0), (16, 32, 32, 2, 2, 1, 2, 0)])
def test_lean_overwrite_dgrad(hip_device, N, C, O, H, W, k, s, p):
    """k_dgrad_ow_s1 / _ks / _img2 / _img4: the overwrite-mode input gradient as the 1x1 convolution it is (3x3 stride 1:
    shifted, edge-clamped dY, four border classes; 3x3 stride 2: up-sampled dY, four parity classes;
    1x1 stride 2: even pixels only), with and without the held gradient."""
    from conftest import assert_close
    from oracle import ref_ops as R
    mod = hip_device.mod
    rng = np.random.default_rng(N * 1000 + C + O + 7 * k + s)
    oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = rng.standard_normal((N, O, oh, ow)).astype(np.float32)
    w = rng.standard_normal((O, C, k, k)).astype(np.float32)
    held = rng.standard_normal((N, C, H, W)).astype(np.float32)
    want = R.conv2d_dgrad(dy, w, (N, C, H, W), s, p, "overwrite")
    dyh, wh, hh = _arr(mod, dy), _arr(mod, w), _arr(mod, held)
    dx = mod.Array(N * C * H * W)
    mod.conv2d_dgrad(dyh, wh, dx, N, C, H, W, O, k, s, p, mode="overwrite")
    assert_close(_back(mod, dx).reshape(N, C, H, W), want, 1e-4, "lean dgrad")
    dx2 = mod.Array(N * C * H * W)
    mod.conv2d_dgrad_add(dyh, wh, hh, dx2, N, C, H, W, O, k, s, p, mode="overwrite")
    assert_close(_back(mod, dx2).reshape(N, C, H, W), want + held, 1e-4, "lean dgrad + held gradient")


@pytest.mark.parametrize("m,k,p", [(256, 256, 10), (37, 19, 5), (128, 784, 256), (5, 70, 33)])
def test_linear_backward_one_call(hip_device, m, k, p):
    """dfhip_linear_backward: dx, dw, db of y = x.W + b in one call (one launch for small layers), any output optional."""
    from conftest import assert_close
    mod = hip_device.mod
    rng = np.random.default_rng(m + k + p)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((k, p)).astype(np.float32)
    dy = rng.standard_normal((m, p)).astype(np.float32)
    xh, wh, dyh = _arr(mod, x), _arr(mod, w), _arr(mod, dy)
    dx, dw, db = mod.Array(m * k), mod.Array(k * p), mod.Array(p)
    mod.linear_backward(xh, wh, dyh, dx, dw, db, m, k, p)
    assert_close(_back(mod, dx).reshape(m, k), dy @ w.T, 1e-4, "dx")
    assert_close(_back(mod, dw).reshape(k, p), x.T @ dy, 1e-4, "dw")
    assert_close(_back(mod, db), dy.sum(0), 1e-4, "db")
    dw2 = mod.Array(k * p)
    mod.linear_backward(xh, wh, dyh, None, dw2, None, m, k, p)     # frozen input / no bias
    assert_close(_back(mod, dw2).reshape(k, p), x.T @ dy, 1e-4, "dw only")
