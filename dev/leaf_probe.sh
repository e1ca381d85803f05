#!/bin/bash
# ONE leaf batch (56 whole-genome pairs, 0.58 Gbp) alone on the device under PGA_VERBOSE (+ PGA_CHAIN_PROF): stage times, the chain kernel's phase clocks, DP classes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
PGA_VERBOSE=1 PGA_CHAIN_PROF=${CHAIN_PROF:-1} python bench.py --leaf-only --slots 1 --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --no-resident-rate --no-parity-check --cap-gbp 0.6 2>&1 | grep -E "n_seq=|chain:|chain stage|backtrack|plans on|regions\+plans|round 0: (probes|host)|dp class|rounds done|CIGAR fin" | tail -${TAILN:-60} | cut -c1-260
