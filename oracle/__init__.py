"""TEST INFRASTRUCTURE ONLY — the CPU oracle for the DeepFlows hot path.

Nothing under oracle/ is imported by the product package `deepflows_amd`; only tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py may import it, and only as the
checker, never as the thing measured or shipped.

Pinning: the reference holds no golden vectors for this path beyond three trivial known-answer
checks (test/test_cuda.py:36-99).  The oracle is therefore pinned against outputs of the
reference itself: oracle/gen_golden.py imports /root/reference (in the build container only),
runs it on seeded inputs and writes tests/golden/*.npz; tests/test_oracle_golden.py checks
this oracle against every one of those fixtures.
"""
