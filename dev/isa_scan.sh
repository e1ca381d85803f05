#!/bin/bash
# static scan of the gfx950 ISA of every kernel for loads that are waited for at once (global_load / buffer_load directly followed by
# s_waitcnt vmcnt(0)): where a source loop means to keep several loads in flight, such a pair is the sign that the compiler sank a load whose
# value is only used under a condition into a branch of its own (pga_sort_wave.h: rs_pin).  Dependent loads show up too: read the hits.
# usage: dev/isa_scan.sh [min_hits=3]   (needs hipcc; no GPU)
MIN=${1:-3}
cd "$(dirname "$0")/../pangraph_amd/csrc" || exit 1
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I. -I../../include -x hip -S --cuda-device-only -o /tmp/isa_scan.s $f 2>/dev/null || continue
  awk -v F=$f -v MIN=$MIN '/^_Z[A-Za-z0-9_]*:/ {name=$1} /global_load|buffer_load/{g=NR; t[name]++} /s_waitcnt vmcnt\(0\)/{ if (NR-g<=1 && g>0) c[name]++ }
    END{for (k in t) if (c[k]>=MIN && k ~ /^_ZN3pga/) { n=k; sub(/^_ZN3pga[0-9]*/,"",n); printf "%-20s %-40s %3d of %3d loads waited for at once\n", F, substr(n,1,40), c[k], t[k]} }' /tmp/isa_scan.s
done | sort -k3 -n -r
