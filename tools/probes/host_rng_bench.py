import ctypes, time, numpy as np, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from deepflows_amd import _lib
lib = _lib.load()
np.random.seed(0)
st = np.random.get_state()
key = np.array(st[1], dtype=np.uint32, copy=True)
n = 524288
ref = {}
for keep in (0.75, 0.5, 0.3):
    np.random.seed(0); ref[keep] = np.random.binomial(1, keep, n).astype(np.float32)
for isa in (0, 1, 2):
    print("isa in effect", lib.dfhip_host_rng_isa(isa))
    out = np.empty(n, np.float32)
    for keep in (0.75, 0.5, 0.3):
        best = 1e9
        for r in range(30):
            pos = ctypes.c_int(int(st[2]))
            k = key.copy()
            t = time.perf_counter()
            lib.dfhip_host_bernoulli_mask(k.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), keep, n, out.ctypes.data_as(ctypes.c_void_p))
            best = min(best, time.perf_counter() - t)
        print(keep, "best %.3f ms for %d" % (best * 1e3, n), np.array_equal(out, ref[keep]))
