"""CPU: host-side code of libdfhip.so that runs WITHOUT a GPU -- the argument checks of the C entry points, the error-string
plumbing, the (empty) pool statistics, group / capture bookkeeping that fails cleanly when no HIP device is visible.  Under
tools/asan_check.sh these run against the AddressSanitizer build of the same sources (DFHIP_LIBRARY); in the ordinary CPU
suite they pin the error behaviour of the boundary: a call on a box without a GPU returns DFHIP_RUNTIME_ERROR /
DFHIP_INVALID_ARGUMENT with a message, it never crashes and never computes."""
import ctypes

import pytest

from deepflows_amd import _lib


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def test_calls_without_a_device_fail_with_a_message(lib):
    n = ctypes.c_int(-1)
    st = lib.dfhip_device_count(ctypes.byref(n))
    if st == 0 and n.value > 0:
        pytest.skip("a GPU is visible: this test is about the box without one")
    out = ctypes.c_void_p()
    st = lib.dfhip_malloc(1024, ctypes.byref(out))
    assert st != 0 and out.value in (None, 0)
    msg = lib.dfhip_last_error().decode()
    assert "HIP" in msg or "device" in msg, msg
    with pytest.raises((RuntimeError, ValueError)):
        _lib.check(st)


def test_argument_checks_come_before_any_device_work(lib):
    # null pointers / impossible shapes are refused by the entry points' own checks or by the missing device -- either way a
    # status and a message, no crash
    for call in (lambda: lib.dfhip_conv2d_forward(None, None, None, None, 1, 1, 1, 1, 1, 3, 1, 1),
                 lambda: lib.dfhip_conv2d_dgrad_bn(None, None, None, None, 256, 32, 16, 16, 32, 3, 1, 1, None, None),
                 lambda: lib.dfhip_bn_bwd_table(None, 32),
                 lambda: lib.dfhip_upload_release(7),
                 lambda: lib.dfhip_comm_check(),
                 lambda: lib.dfhip_graph_end(None) if False else lib.dfhip_wgrad_group_end(),
                 lambda: lib.dfhip_allreduce_sum(None, 0)):
        st = call()
        assert isinstance(st, int)
        if st != 0:
            assert lib.dfhip_last_error()          # a message is there


def test_bn_descriptor_layout_matches_the_header(lib):
    """sizeof / field offsets of the ctypes mirrors of dfhip_bn_bwd_in / dfhip_bn_stats_out as the C compiler lays them out
    (pointers 8-aligned, ints 4): a mismatch would make the kernels read garbage descriptors"""
    b, o = _lib.BnBwdIn, _lib.BnStatsOut
    assert b.z.offset == 0 and b.rows.offset == 8 and b.nrows.offset == 16 and b.K.offset == 20 and b.kidx.offset == 24
    assert b.mean.offset == 32 and b.var.offset == 40 and b.gamma.offset == 48 and b.eps.offset == 56 and b.inv_m.offset == 60
    assert b.dgamma.offset == 64 and b.dbeta.offset == 72 and b.table.offset == 80 and ctypes.sizeof(b) == 88
    assert o.n.offset == 0 and o.z.offset == 8 and o.mean.offset == 24 and o.rows.offset == 40 and o.cap_rows.offset == 48
    assert o.nrows.offset == 52 and ctypes.sizeof(o) == 56
