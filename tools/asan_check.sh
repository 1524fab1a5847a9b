#!/bin/bash
# Host-side AddressSanitizer run of the library-facing CPU tests (build container only: there is no GPU here and GPU ASan is
# not available on the pool).  Builds deepflows_amd/csrc/build_asan/libdfhip_asan.so (`make asan`) and runs the tests that
# load the library -- loader / exported symbols / header-binding agreement, the protocol and host-logic tests, the launcher
# and rendezvous tests -- with DFHIP_LIBRARY pointing at it and the clang ASan runtime preloaded.
set -e
cd "$(dirname "$0")/.."
make -C deepflows_amd/csrc asan -j8 >/dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export DFHIP_LIBRARY="$PWD/deepflows_amd/csrc/build_asan/libdfhip_asan.so"
# (python itself is not instrumented: leak reports of the interpreter are noise; new/delete mismatches, overflows and
# use-after-free inside the library abort the run)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0
LD_PRELOAD="$RT" python -m pytest tests/test_cabi_exports.py tests/test_frontend_host_logic.py tests/test_launch.py \
    tests/test_rendezvous.py tests/test_asan_host_paths.py -q -x -m "not gpu" "$@"
