#!/bin/bash
# dev/build_ab_lib.sh <commit> <name>: the library of another commit as dev/ab_libs/libpgalign_<name>.so, for an A/B on ONE box
# (PGA_LIB=$PWD/dev/ab_libs/libpgalign_<name>.so; boxes differ by 5-10 %: only runs of the same gpurun call compare).  Remove dev/ab_libs afterwards.
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/ab_wt && git worktree add -f /tmp/ab_wt "$1" -q
make -C /tmp/ab_wt/pangraph_amd/csrc -j16 ../libpgalign.so > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
mkdir -p dev/ab_libs && cp /tmp/ab_wt/pangraph_amd/libpgalign.so dev/ab_libs/libpgalign_$2.so
git worktree remove --force /tmp/ab_wt
ls -la dev/ab_libs
