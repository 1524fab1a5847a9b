#!/bin/bash
# build the library first (a stale libdfhip.so travels to the GPU box as it is), then hand the command to gpurun
set -e
make -C "$(dirname "$0")/../deepflows_amd/csrc" -j8 >/dev/null
exec /usr/local/graft/bin/gpurun "$@"
