#!/bin/bash
# build the library first (a stale libdfhip.so travels to the GPU box as it is), then hand the command to gpurun
set -e
make -C "$(dirname "$0")/../deepflows_amd/csrc" -j8 >/dev/null
# the snapshot has no .git: leave the commit (and whether the tree is dirty) where the tools on the box find it
(cd "$(dirname "$0")/.." && echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- . ':!.git_head' || echo +dirty)" > .git_head)
exec /usr/local/graft/bin/gpurun "$@"
