"""ctypes loader for libdfhip.so (the C ABI declared in include/dfhip.h).

The library is built in-tree (deepflows_amd/csrc/Makefile -> deepflows_amd/libdfhip.so).
There is NO fallback: if the shared object is missing or no HIP device is visible, asking
for the `hip` device raises — the product path never computes on the CPU.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFHIP_LIBRARY: a diagnostic build of the same library (tools/probes/build/*), never a different implementation
LIB_PATH = os.environ.get("DFHIP_LIBRARY") or os.path.join(_HERE, "libdfhip.so")

_lib = None

# status -> exception class, the classes pybind11 raised for the reference's CUDA module
# (std::invalid_argument -> ValueError, out_of_range -> IndexError, runtime_error ->
# RuntimeError, bad_alloc -> MemoryError; SURVEY §8b "Errors")
_EXC = {1: ValueError, 2: IndexError, 3: RuntimeError, 4: MemoryError}

P = c_void_p
I64P = POINTER(c_int64)
PP = POINTER(c_void_p)


class BnBwdIn(ctypes.Structure):
    """dfhip_bn_bwd_in (include/dfhip.h): a BatchNorm2d backward pass applied on the consumer's operand loads."""
    _fields_ = [("z", P), ("rows", P), ("nrows", c_int), ("K", c_int), ("kidx", c_int), ("mean", P), ("var", P),
                ("gamma", P), ("eps", c_float), ("inv_m", c_float), ("dgamma", P), ("dbeta", P), ("table", P)]


class BnStatsOut(ctypes.Structure):
    """dfhip_bn_stats_out: the partial rows a producer leaves behind for the BatchNorm layers its output feeds."""
    _fields_ = [("n", c_int), ("z", P * 2), ("mean", P * 2), ("rows", P), ("cap_rows", c_int), ("nrows", c_int)]


_SIGS = {
    "dfhip_init": [c_int],
    "dfhip_host_bernoulli_mask": [P, POINTER(c_int), c_double, c_int64, P],
    "dfhip_upload_bernoulli_mask": [P, POINTER(c_int), c_double, c_int64, P],
    "dfhip_upload_bernoulli_masks": [P, POINTER(c_int), c_int, P, P, P],
    "dfhip_host_rng_isa": [c_int],
    "dfhip_device_count": [POINTER(c_int)],
    "dfhip_sync": [],
    "dfhip_device_name": [c_char_p, c_int],
    "dfhip_malloc": [c_int64, POINTER(c_void_p)],
    "dfhip_free": [P],
    "dfhip_pool_stats": [I64P, I64P, I64P],
    "dfhip_pool_trim": [],
    "dfhip_h2d": [P, P, c_int64],
    "dfhip_d2h": [P, P, c_int64],
    "dfhip_d2d": [P, P, c_int64],
    "dfhip_fill": [P, c_int64, c_float],
    "dfhip_compact": [P, P, c_int64, c_int, I64P, I64P, c_int64],
    "dfhip_ewise_setitem": [P, P, c_int64, c_int, I64P, I64P, c_int64],
    "dfhip_scalar_setitem": [c_int64, c_float, P, c_int, I64P, I64P, c_int64],
    "dfhip_matmul": [P, P, P, c_int64, c_int64, c_int64],
    "dfhip_reduce_sum": [P, P, c_int64, c_int64],
    "dfhip_reduce_max": [P, P, c_int64, c_int64],
    "dfhip_gemm": [P, P, P, P, c_int64, c_int64, c_int64, c_int, c_int],
    "dfhip_linear_backward": [P, P, P, P, P, P, c_int64, c_int64, c_int64],
    "dfhip_linear_relu_forward": [P, P, P, P, P, c_int64, c_int64, c_int64, POINTER(c_int)],
    "dfhip_linear_relu_backward": [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, POINTER(c_int)],
    "dfhip_conv2d_forward": [P, P, P, P] + [c_int] * 8,
    "dfhip_conv2d_forward_stats": [P, P, P, P] + [c_int] * 8 + [P, c_int, POINTER(c_int)],
    "dfhip_bn2d_forward_partials": [P, P, c_int] + [P] * 8 + [c_int, c_int, c_int, c_float, c_float],
    "dfhip_conv2d_wgrad": [P, P, P] + [c_int] * 8,
    "dfhip_conv2d_wgrad_affine": [P, P, P, P] + [c_int] * 8,
    "dfhip_conv2d_forward_bn_input": [P, P, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P, P] + [c_int] * 8 +
                                     [P, c_int, POINTER(c_int), POINTER(c_int)],
    "dfhip_wgrad_group_begin": [],
    "dfhip_wgrad_group_end": [],
    "dfhip_conv2d_dgrad": [P, P, P] + [c_int] * 9,
    "dfhip_conv2d_dgrad_add": [P, P, P, P] + [c_int] * 9,
    "dfhip_conv2d_dgrad_pair": [P, P, P, P, P, P] + [c_int] * 6 + [POINTER(c_int)],
    "dfhip_conv2d_bn_bwd_supported": [c_int] * 9 + [POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    "dfhip_conv2d_dgrad_bn": [P, P, P, P] + [c_int] * 8 + [POINTER(BnBwdIn), POINTER(BnStatsOut)],
    "dfhip_conv2d_dgrad_pair_bn": [P, P, P, P, P, P] + [c_int] * 5 + [POINTER(BnBwdIn), POINTER(BnBwdIn), POINTER(BnStatsOut)],
    "dfhip_conv2d_wgrad_bn": [P, P, P, P, P, P] + [c_int] * 8 + [POINTER(c_int)],
    "dfhip_bn_bwd_table": [POINTER(BnBwdIn), c_int],
    "dfhip_bn2d_backward_apply_table": [P, P, P, P, c_int, c_int, c_int],
    "dfhip_conv2d_forward_s2_pair": [P, P, P, P, P] + [c_int] * 5 + [P, P, c_int, POINTER(c_int), POINTER(c_int)],
    "dfhip_bn2d_forward_train": [P] * 8 + [c_int, c_int, c_int, c_float, c_float],
    "dfhip_bn2d_forward_eval": [P] * 6 + [c_int, c_int, c_int, c_float],
    "dfhip_bn2d_backward": [P] * 8 + [c_int, c_int, c_int, c_float],
    "dfhip_bn2d_forward_partials_pair": ([P, P, c_int] + [P] * 6 + [c_float, c_float]) * 2 + [P, c_int, c_int, c_int],
    "dfhip_bn2d_backward_pair": [P] + ([P] * 7 + [c_float]) * 2 + [c_int, c_int, c_int],
    "dfhip_bn_relu_maxpool2_forward": [P] * 8 + [c_int] * 4 + [c_float, c_float],
    "dfhip_bn_relu_maxpool2_backward": [P] * 10 + [c_int] * 4 + [c_float],
    "dfhip_bn_relu_maxpool2_forward_partials": [P, P, c_int] + [P] * 7 + [c_int] * 4 + [c_float, c_float],
    "dfhip_stem_backward": [P] * 11 + [c_int] * 8 + [c_float, POINTER(c_int)],
    "dfhip_stem_forward_noz": [P] * 10 + [c_int] * 8 + [c_float, c_float, POINTER(c_int)],
    "dfhip_stem_backward_noz": [P] * 13 + [c_int] * 8 + [c_float, POINTER(c_int)],
    "dfhip_upload_async": [P, P, c_int64, c_int],
    "dfhip_upload_wait": [c_int],
    "dfhip_upload_release": [c_int],
    "dfhip_augment_batch": [P, P, P, c_int, c_int, c_int, c_int, c_int],
    "dfhip_onehot_smooth": [P, P, c_int, c_int, c_double],
    "dfhip_maxpool2d_forward": [P, P, c_int, c_int, c_int, c_int],
    "dfhip_maxpool2d_backward": [P, P, P, P, c_int, c_int, c_int, c_int],
    "dfhip_relu_forward": [P, P, c_int64],
    "dfhip_relu_backward": [P, P, P, P, c_int64],
    "dfhip_softmax_ce_forward": [P, P, P, c_int, c_int],
    "dfhip_softmax_ce_backward": [P, P, P, P, c_int, c_int],
    "dfhip_softmax_ce_forward_grad": [P, P, P, P, c_int, c_int, POINTER(c_int)],
    "dfhip_pool_linear_forward": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float],
    "dfhip_pool_linear_backward": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float],
    "dfhip_head_ce_forward": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P],
    "dfhip_droplin_ce_forward": [P, P, c_float, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P],
    "dfhip_head_grads": [P, P, P, P, P, P, c_int, c_int, c_int],
    "dfhip_loss_finish": [P, P, c_int],
    "dfhip_dropout_apply": [P, P, P, c_int64, c_float, c_int],
    "dfhip_set_conv_matmul_dtype": [c_int],
    "dfhip_conv_matmul_dtype": [],
    "dfhip_mlp2_ce_forward": [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "dfhip_mlp2_ce_backward": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int],
    "dfhip_reduce_to_shape": [P, P, c_int, I64P, c_int, I64P],
    "dfhip_reduce_axis": [P, P, c_int64, c_int64, c_int64, c_int, c_float],
    "dfhip_broadcast_axis": [P, P, c_int64, c_int64, c_int64, c_float],
    "dfhip_axpy": [P, P, c_float, c_int64],
    "dfhip_adam_step": [c_int, PP, PP, PP, PP, I64P, c_double, c_double, c_double, c_double,
                        c_double, c_int, c_double],
    "dfhip_adam_hyper": [P, c_double, c_double, c_double, c_double, c_double, c_int, c_double],
    "dfhip_adam_hyper_table": [P, c_int, c_double, c_double, c_double, c_double, c_double, c_int, c_double],
    "dfhip_adam_hyper_index": [P, POINTER(c_int), POINTER(c_int)],
    "dfhip_adam_step_dev": [c_int, PP, PP, PP, PP, I64P, P],
    "dfhip_sgd_step": [c_int, PP, PP, PP, I64P, c_double, c_double, c_double, c_int, c_double],
    "dfhip_sgd_hyper": [P, c_double, c_double, c_double, c_int, c_double],
    "dfhip_sgd_step_dev": [c_int, PP, PP, PP, I64P, P],
    "dfhip_pack": [c_int, PP, I64P, P],
    "dfhip_unpack": [c_int, PP, I64P, P],
    "dfhip_comm_unique_id": [c_char_p],
    "dfhip_comm_init": [c_int, c_int, c_char_p],
    "dfhip_allreduce_sum": [P, c_int64],
    "dfhip_allreduce_sum_async": [P, c_int64],
    "dfhip_comm_wait": [],
    "dfhip_comm_check": [],
    "dfhip_comm_allreduce_max": [POINTER(c_float)],
    "dfhip_comm_rccl_path": [c_char_p, c_int],
    "dfhip_comm_count": [POINTER(c_int), POINTER(c_int)],
    "dfhip_clock_probe": [c_double, POINTER(c_double), POINTER(c_double)],
    "dfhip_comm_destroy": [],
    "dfhip_graph_begin": [],
    "dfhip_graph_end": [POINTER(c_void_p)],
    "dfhip_graph_launch": [P],
    "dfhip_graph_destroy": [P],
    "dfhip_event_create": [POINTER(c_void_p)],
    "dfhip_event_record": [P],
    "dfhip_event_elapsed_ms": [P, P, POINTER(c_float)],
    "dfhip_event_destroy": [P],
    "dfhip_prof_enable": [c_int],
    "dfhip_prof_reset": [],
    "dfhip_prof_gate_close": [c_int],
    "dfhip_prof_gate_open": [],
}
for _n in ("add", "mul", "div", "maximum", "eq", "ge"):
    _SIGS["dfhip_ewise_" + _n] = [P, P, P, c_int64]
for _n in ("add", "mul", "div", "power", "maximum", "eq", "ge"):
    _SIGS["dfhip_scalar_" + _n] = [P, c_float, P, c_int64]
for _n in ("log", "exp", "tanh"):
    _SIGS["dfhip_ewise_" + _n] = [P, P, c_int64]

# every symbol include/dfhip.h declares (tests check the .so exports all of them)
ALL_SYMBOLS = sorted(list(_SIGS) + ["dfhip_last_error", "dfhip_stream", "dfhip_prof_read", "dfhip_prof_read_kernels"])


def library_present():
    return os.path.exists(LIB_PATH)


def load():
    """dlopen libdfhip.so and attach argument types. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C deepflows_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    lib.dfhip_last_error.argtypes = []
    lib.dfhip_last_error.restype = c_char_p
    lib.dfhip_stream.argtypes = []
    lib.dfhip_stream.restype = c_void_p
    lib.dfhip_prof_read.argtypes = [c_int, POINTER(c_char_p), POINTER(c_double), I64P,
                                    POINTER(c_double), POINTER(c_double), POINTER(c_double), I64P]
    lib.dfhip_prof_read.restype = c_int
    lib.dfhip_prof_read_kernels.argtypes = [c_int, POINTER(c_char_p), POINTER(c_char_p), POINTER(c_double), I64P,
                                            POINTER(c_double), POINTER(c_double)]
    lib.dfhip_prof_read_kernels.restype = c_int
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = _lib.dfhip_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, RuntimeError)(msg)
