"""Which kernel instantiations of libdfhip.so does anything launch?  (round 5 prune)

    DFHIP_KERNEL_LOG=/tmp/k.log python -m pytest tests -m gpu -q ...; DFHIP_KERNEL_LOG=/tmp/k.log python bench.py ...
    python tools/kernel_coverage.py /tmp/k.log [more logs]

compares the mangled names the runs appended (dfhip_runtime.hip: note_kernel) with the `__device_stub__` symbols of the shared
object: every instantiation nobody launched is printed (demangled), grouped by kernel template."""
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    so = os.path.join(ROOT, "deepflows_amd", "libdfhip.so")
    syms = subprocess.run(["nm", so], capture_output=True, text=True, check=True).stdout.split("\n")
    stubs = set()
    for line in syms:
        parts = line.split()
        if len(parts) == 3 and "__device_stub__" in parts[2]:
            stubs.add(parts[2])
    launched = set()
    for path in sys.argv[1:]:
        for line in open(path):
            launched.add(line.strip())

    def kernel_of_stub(stub):
        # _ZN12_GLOBAL__N_131__device_stub__k_fooILi1EEEvPf -> the kernel's own mangled name differs only by the stub prefix:
        # compare on the demangled text instead
        return stub

    dem = lambda names: dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
    ds, dl = dem(sorted(stubs)), dem(sorted(launched))
    norm = lambda t: t.replace("__device_stub__", "").replace("(anonymous namespace)::", "").replace("void ", "").strip()
    have = {norm(v): k for k, v in ds.items()}
    used = {norm(v) for v in dl.values()}
    dead = sorted(n for n in have if n not in used)
    by = collections.defaultdict(list)
    for n in dead:
        by[n.split("<")[0].split("(")[0]].append(n)
    print(f"{len(have)} instantiations in the library, {len(used & set(have))} launched, {len(dead)} never launched")
    for k in sorted(by):
        print(f"\n{k}: {len(by[k])} never launched")
        for n in by[k]:
            print("   ", n[:200])


if __name__ == "__main__":
    main()
