"""one long banded end extension through pga_stage_extd2 (timing probe for the DP kernels; PGA_VERBOSE=1 prints the class times)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import stagebind as sb
from pangraph_amd import batch

rng = np.random.default_rng(5)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 9900
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
flag = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x40
full = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0    # fraction of problems whose homology runs to the end (the others z-drop)
jobs = []
for i in range(n):
    t = rng.integers(0, 4, L).astype(np.uint8)
    q = t.copy()
    m = rng.random(L) < 0.01
    q[m] = (q[m] + 1) % 4
    if i >= full * n:
        q[700:] = rng.integers(0, 4, L - 700)
    jobs.append((q, t, w, 400, -1, flag))
dll = batch.lib()
for rep in range(3):
    t0 = time.time()
    out = sb.product_extd2(dll, jobs, 1, 9, 1, 16, 2, 41, 1)
    print("rep", rep, "%.1f ms" % ((time.time() - t0) * 1e3), "zdropped", out[0]["zdropped"], "n_cigar", len(out[0]["cigar"]), flush=True)
