#!/bin/bash
# Build a diagnostic variant of libdfhip.so into tools/probes/build/<name>/ with extra compiler flags:
#   tools/build_variant.sh stamps -DDFHIP_ROWS_STAMPS ;  DFHIP_LIBRARY=tools/probes/build/stamps/libdfhip.so python ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/probes/build/$name
mkdir -p $out
cd $root/deepflows_amd/csrc
# ONLY=<file stem> picks the translation unit that gets the flags (default dfhip_conv_rows); ALL=1 rebuilds every one
only=${ONLY:-dfhip_conv_rows}
for f in dfhip_runtime dfhip_hostrng dfhip_prims dfhip_gemm dfhip_conv dfhip_wgrad dfhip_dgrad_ow dfhip_conv_rows dfhip_conv_s2 dfhip_stem dfhip_fused dfhip_mlp; do
  if [ "$f" = "$only" ] || [ ! -f build/$f.o ] || [ -n "$ALL" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed "$@" -c $f.hip -o $out/$f.o
  else
    cp build/$f.o $out/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfhip.so $out/*.o -ldl
echo built $out/libdfhip.so
