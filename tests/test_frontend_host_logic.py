"""CPU: host logic of the front-end mirror (views, broadcasting, registration quirks, schedulers)
on the oracle device."""
import math

import numpy as np
import pytest

from deepflows_amd import backend_api, nn, optim, tensor, workloads
from deepflows_amd.backend import BackendTensor
from deepflows_amd.tensor import Graph, Tensor


def test_views_never_copy_and_compact_materialises(cpu_oracle):
    a = BackendTensor(np.arange(24, dtype=np.float32).reshape(2, 3, 4), device=cpu_oracle)
    p = a.permute((2, 0, 1))
    assert p._handle is a._handle and not p.is_compact()
    assert np.array_equal(p.numpy(), np.arange(24).reshape(2, 3, 4).transpose(2, 0, 1))
    c = p.compact()
    assert c.is_compact() and c._handle is not a._handle
    s = a[0:2, 1:3, 0:4:2]
    assert s.shape == (2, 2, 2) and np.array_equal(s.numpy(), np.arange(24).reshape(2, 3, 4)[:, 1:3, ::2])
    # a contiguous slice of a larger buffer is NOT compact (handle.size differs)
    assert not a[0:1, 0:3, 0:4].is_compact()
    with pytest.raises(ValueError):
        p.reshape((24,))
    assert a.reshape((-1, 4)).shape == (6, 4) and a.reshape((6, -1)).shape == (6, 4)


def test_only_other_is_broadcast(cpu_oracle):
    x = BackendTensor(np.ones((2, 3), np.float32), device=cpu_oracle)
    b = BackendTensor(np.arange(3, dtype=np.float32).reshape(1, 3), device=cpu_oracle)
    assert np.array_equal((x + b).numpy(), np.ones((2, 3)) + np.arange(3))
    with pytest.raises(AssertionError):
        b + x                                    # self is never broadcast (backend_tensor.py:533-542)


def test_mean_divides_by_total_count(cpu_oracle):
    a = BackendTensor(np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2), device=cpu_oracle)
    m = a.mean(axis=2)
    assert np.allclose(m.numpy(), np.arange(24).reshape(2, 3, 2, 2).sum(2) / 24.0)


def test_setitem_scatters_into_the_view(cpu_oracle):
    a = backend_api.zeros((3, 4), device=cpu_oracle)
    a[1:3, 1:3] = BackendTensor(np.array([[1, 2], [3, 4]], np.float32), device=cpu_oracle)
    want = np.zeros((3, 4), np.float32)
    want[1:3, 1:3] = [[1, 2], [3, 4]]
    assert np.array_equal(a.numpy(), want)


def test_plain_lists_are_not_registered(cpu_oracle):
    np.random.seed(0)
    ref_style = workloads.ResNet18(10, device=cpu_oracle, register_blocks=False)
    full = workloads.ResNet18(10, device=cpu_oracle, register_blocks=True)
    p5 = list(ref_style.parameters())
    assert len(p5) == 5 and sum(p.size for p in p5) == 3498
    p62 = list(full.parameters())
    assert len(p62) == 62 and sum(p.size for p in p62) == 2797610
    assert len(workloads.all_block_parameters(ref_style)) == 62


def test_train_eval_toggle_global_grad_flag(cpu_oracle):
    from deepflows_amd.autograd import is_grad_enable
    m = nn.Linear(3, 2, device=cpu_oracle)
    m.eval()
    assert not is_grad_enable()
    y = m(Tensor(np.ones((1, 3), np.float32), device=cpu_oracle))
    assert not y.requires_grad and y not in Graph.node_list
    m.train()
    assert is_grad_enable()


def test_backward_only_from_1d_and_free_graph_keeps_leaves(cpu_oracle):
    w = Tensor(np.ones((2, 2), np.float32), device=cpu_oracle, requires_grad=True)
    y = w * 2.0
    with pytest.raises(ValueError):
        y.backward()
    loss = tensor.sum(y)
    loss.backward()
    assert np.array_equal(w.grad.numpy(), np.full((2, 2), 2.0, np.float32))
    assert Graph.node_list == [w]


def test_schedulers_match_closed_forms(cpu_oracle):
    class Opt:
        lr = 0.1
    o = Opt()
    s = optim.WarmupCosineLR(o, warmup_epochs=5, T_max=20, eta_min=1e-5)
    lrs = []
    for _ in range(8):
        s.step()
        lrs.append(o.lr)
    assert lrs[0] == 0.0 and abs(lrs[5] - 0.1) < 1e-12
    t = 1
    assert abs(lrs[6] - (1e-5 + (0.1 - 1e-5) * (1 + math.cos(math.pi * t / 20)) / 2)) < 1e-12
    o2 = Opt()
    s2 = optim.StepLR(o2, step_size=2, gamma=0.5)
    out = []
    for _ in range(5):
        s2.step()
        out.append(o2.lr)
    assert out == [0.1, 0.1, 0.05, 0.05, 0.025]


def test_schedulers_reproduce_the_reference_sequences():
    """25 epochs of every schedule against the learning rates the reference's own classes produced
    (optim/scheduler.py:13-61 run by oracle/gen_f3_golden.py): equal to the last bit, same expression order."""
    from conftest import load_golden
    g = load_golden("g11_nn_surface.npz")

    class Opt:
        def __init__(self, lr):
            self.lr = lr

    def run(make, lr):
        o = Opt(lr)
        s = make(o)
        seq = []
        for _ in range(25):
            s.step()
            seq.append(o.lr)
        return np.array(seq, dtype=np.float64)

    assert np.array_equal(run(lambda o: optim.StepLR(o, step_size=3, gamma=0.5), 0.1), g["lr_step"])
    assert np.array_equal(run(lambda o: optim.CosineAnnealingLR(o, T_max=7, eta_min=1e-5), 0.05), g["lr_cosine"])
    assert np.array_equal(run(lambda o: optim.WarmupCosineLR(o, warmup_epochs=2, T_max=10, eta_min=1e-5), 1e-3),
                          g["lr_warmcos"])
    assert np.array_equal(run(lambda o: optim.WarmupCosineLR(o, warmup_epochs=4, T_max=6, base_lr=0.02,
                                                             warmup_start_lr=1e-4, eta_min=0.0), 7.0),
                          g["lr_warmcos_explicit"])
    assert np.array_equal(run(lambda o: optim.WarmupCosineLR(o, warmup_epochs=0, T_max=5), 0.3), g["lr_warmcos_nowarm"])
    # an optimizer without a learning rate is left alone
    class NoLr:
        pass
    for make in (lambda o: optim.StepLR(o, 1), lambda o: optim.CosineAnnealingLR(o, 3),
                 lambda o: optim.WarmupCosineLR(o, 1, 3)):
        o = NoLr()
        s = make(o)
        s.step(); s.step()
        assert not hasattr(o, "lr") and s.last_epoch == 1


def test_state_dict_roundtrip_excludes_bn_running_stats(cpu_oracle):
    np.random.seed(1)
    m = workloads.CIFAR10_CNN(10, device=cpu_oracle)
    sd = m.state_dict()
    assert "bn1.weight" in sd and "bn1.running_mean" not in sd     # plain attributes in the reference
    np.random.seed(2)
    m2 = workloads.CIFAR10_CNN(10, device=cpu_oracle)
    m2.load_state_dict({k: v.numpy() for k, v in sd.items()})
    for k, v in m2.state_dict().items():
        assert np.array_equal(v.numpy(), sd[k].numpy()), k


def test_dropout_mask_equals_numpy_binomial():
    """Dropout draws its mask with a vectorised threshold instead of np.random.binomial; values and
    generator state must match the reference's call (dropout.py:15-20) exactly."""
    import numpy as np
    from deepflows_amd.nn.modules.dropout import bernoulli_mask
    for keep in (0.5, 0.1, 0.3, 0.7, 0.9, 0.25, 0.8, 0.999, 0.001):
        for seed in range(4):
            np.random.seed(seed)
            a = np.random.binomial(1, keep, (37, 129)).astype(np.float32)
            sa = np.random.get_state()
            np.random.seed(seed)
            b = bernoulli_mask(keep, (37, 129))
            sb = np.random.get_state()
            assert np.array_equal(a, b), (keep, seed)
            assert sa[2] == sb[2] and np.array_equal(sa[1], sb[1]), (keep, seed)


def test_engine_skips_a_contribution_a_sibling_node_already_delivered(cpu_oracle):
    """A fused node may return None from grad_fn for a parent whose gradient another node's launch already carried
    (the shortcut convolution of a residual block): the sweep leaves the parent's gradient as it is -- with and
    without a gradient already held."""
    from deepflows_amd.nn.functional import _FusedNode

    class _Delivers(_FusedNode):
        fuses_accumulation = True

        def __init__(self, x, factor):
            self.x, self.factor = x, factor
            super().__init__(x.data * 1.0, [x])

        def grad_fn(self, node, grad, accumulated=None):
            if self.factor is None:
                return None                                  # "already delivered"
            out = grad * self.factor
            return out if accumulated is None else out + accumulated

    x = Tensor(np.arange(4, dtype=np.float32), requires_grad=True, device=cpu_oracle)
    xin = x * 1.0
    a = _Delivers(xin, 2.0)            # created first: visited last
    b = _Delivers(xin, None)           # visited first, returns None with no gradient held
    c = _Delivers(xin, 3.0)
    d = _Delivers(xin, None)           # returns None while a gradient is already held (the accumulation branch)
    loss = tensor.sum(a + b + c + d)
    loss.backward()
    assert np.allclose(x.grad.numpy(), 5.0 * np.ones(4))


def test_large_dropout_masks_continue_numpys_stream_in_c():
    """Masks of 8 192 elements and more come from libdfhip's host helper (dfhip_host_bernoulli_mask: the MT19937
    stream of NumPy's global generator continued in vectorised C): same values as np.random.binomial(1, keep, n) and
    the same generator state afterwards, from even and odd stream positions and across block boundaries."""
    import numpy as np
    from deepflows_amd import _lib
    from deepflows_amd.nn.modules.dropout import _continue_numpy_stream, bernoulli_mask
    if not _lib.library_present():
        pytest.skip("libdfhip.so is not built")
    lib = _lib.load()
    try:
        for keep in (0.5, 0.3, 0.8):
            for n in (8192, 100001, 524288):
                for pre, odd in ((0, False), (1, False), (311, True), (312, True), (625, False)):
                    def prep():
                        np.random.seed(11)
                        if pre:
                            np.random.random_sample(pre)
                        if odd:
                            np.random.randint(0, 1000)       # one 32-bit word: the stream position becomes odd
                    prep()
                    a = np.random.binomial(1, keep, n).astype(np.float32)
                    nxt_a = np.random.random_sample(5)
                    # every vector width the library carries (plain C, AVX2, AVX-512: a level the host lacks is clamped)
                    for isa in (0, 1, 2):
                        lib.dfhip_host_rng_isa(isa)
                        prep()
                        b = _continue_numpy_stream(keep, (n,))
                        assert b is not None
                        nxt_b = np.random.random_sample(5)
                        assert np.array_equal(a, b), (keep, n, pre, odd, isa)
                        assert np.array_equal(nxt_a, nxt_b), (keep, n, pre, odd, isa)
    finally:
        lib.dfhip_host_rng_isa(-1)
    # the ends of the range: keep = 1 (every element kept), keep = 0
    for keep in (1.0, 0.0):
        np.random.seed(5)
        a = np.random.binomial(1, keep, 10000).astype(np.float32)
        np.random.seed(5)
        assert np.array_equal(a, _continue_numpy_stream(keep, (10000,))), keep
    np.random.seed(2)
    a = np.random.binomial(1, 0.5, (256, 2048)).astype(np.float32)
    np.random.seed(2)
    assert np.array_equal(a, bernoulli_mask(0.5, (256, 2048)))


def test_backward_brackets_the_sweep_with_the_wgrad_group(cpu_oracle):
    """Tensor.backward() opens a device's weight-gradient group before the reverse sweep and closes it
    after it -- also when a grad_fn raises -- and leaves devices without the hooks alone."""
    calls = []
    mod = cpu_oracle.mod
    mod.wgrad_group_begin = lambda: calls.append("begin") or True
    mod.wgrad_group_end = lambda: calls.append("end")
    try:
        w = Tensor(np.ones((3, 2), np.float32), requires_grad=True, device=cpu_oracle)
        x = Tensor(np.arange(6, dtype=np.float32).reshape(2, 3), device=cpu_oracle)
        loss = tensor.sum(x @ w)
        loss.backward()
        assert calls == ["begin", "end"]
        assert np.allclose(w.grad.numpy(), np.arange(6).reshape(2, 3).sum(0)[:, None] * np.ones((1, 2)))
        # a failing grad_fn still closes the group
        calls.clear()
        w2 = Tensor(np.ones((3, 2), np.float32), requires_grad=True, device=cpu_oracle)
        y = x @ w2
        loss2 = tensor.sum(y)

        def boom(*a, **k):
            raise RuntimeError("grad_fn failed")
        y.grad_fn = boom
        with pytest.raises(RuntimeError):
            loss2.backward()
        assert calls == ["begin", "end"]
        Graph.free_graph_all()                     # the failed sweep left its tape behind
        # begin() returning False (grouping switched off): end is not called
        calls.clear()
        mod.wgrad_group_begin = lambda: calls.append("begin") and False
        w3 = Tensor(np.ones((3, 2), np.float32), requires_grad=True, device=cpu_oracle)
        tensor.sum(x @ w3).backward()
        assert calls == ["begin"]
    finally:
        del mod.wgrad_group_begin, mod.wgrad_group_end
        Graph.free_graph_all()


def test_device_modules_follow_the_references_protocol():
    """The 26 methods of the reference's device module (names, parameter names, parameter order) as the REFERENCE defines
    them -- tests/golden/protocol_signatures.json, written by oracle/gen_protocol_signatures.py from
    DeepFlows/backend/backend_tensor.py:64-172 (inspect.signature of NumpyBackend) and the m.def names of
    ndarray_backend_cuda.cu:515-717 -- against the hip device module and the oracle's NumPy device."""
    import inspect
    import json
    import os
    from conftest import GOLDEN
    from deepflows_amd.hip_device import HipModule
    from oracle.numpy_device import NumpyDevice
    ref = json.load(open(os.path.join(GOLDEN, "protocol_signatures.json")))
    assert len(ref["numpy_backend"]) == 26
    for cls in (HipModule, NumpyDevice):
        for name, params in ref["numpy_backend"].items():
            fn = getattr(cls, name, None)
            assert fn is not None, f"{cls.__name__} lacks protocol method {name}"
            sig = inspect.signature(fn)
            mine = [p for p in sig.parameters if p != "self"]
            if cls is HipModule:
                assert mine[:len(params)] == params, f"{cls.__name__}.{name}{tuple(mine)}: the reference has {tuple(params)}"
            else:
                assert len(mine) >= len(params), f"{cls.__name__}.{name} takes fewer arguments than the reference's"
            extra = list(sig.parameters.values())[1 + len(params):]
            assert all(p.default is not inspect.Parameter.empty for p in extra), f"{cls.__name__}.{name}: extra required arguments"
    # every function the reference's CUDA extension exports exists on the hip module under the same name
    for name in ref["cuda_module"]["functions"]:
        assert hasattr(HipModule, name), f"HipModule lacks {name} (exported by the reference's CUDA module)"
    assert ref["cuda_module"]["classes"] == ["Array"] and hasattr(HipModule, "Array")
