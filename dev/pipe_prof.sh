#!/bin/bash
# the workgroup pipeline's cycle split per wave (build with -DPP_PROF into a scratch library, one junk extension and one full-length one)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
cd pangraph_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -I../../include -DPP_PROF -x hip -c pga_ksw_pipe.hip -o /tmp/pga_ksw_pipe_prof.o || exit 1
objs=$(ls *.o | grep -v pga_ksw_pipe.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpgalign_prof.so $objs /tmp/pga_ksw_pipe_prof.o -lpthread || exit 1
cd ../..
for full in 0.0 1.0; do
  echo "== full=$full"; PGA_LIB=/tmp/libpgalign_prof.so PGA_BSTRIPS=off PGA_PIPE=force python dev/dp_probe.py 9900 1 1500 0x40 $full 2>&1 | grep -E "pipe prof|rep 2" | tail -10
done
