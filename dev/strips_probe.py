"""Timing of ONE large unbanded problem (the 10 kb x 10 kb gap fills across rearrangements): first pass (approximate) and exact second pass,
through the wave strips (default) or the workgroup strips (PGA_OLD_STRIPS=1).  PGA_VERBOSE=1 prints the class times."""
import sys, os, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
faulthandler.dump_traceback_later(240, exit=True)
import numpy as np
import stagebind as sb
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq, mutate
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
rng = np.random.default_rng(7)
def job(L, fl, kind):
    t = random_seq(rng, L)
    q = random_seq(rng, L) if kind == "junk" else mutate(rng, t, snp=0.01, indel=0.001)
    if kind == "half": q = np.concatenate([q[:L // 2], random_seq(rng, L // 2)])
    return (sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), 150001, 200, -1, fl)
for L in (10000, 4000, 2048):
    for kind in ("junk", "half", "related"):
        for fl, name in ((0x08, "approx"), (0, "exact")):
            jobs = [job(L, fl, kind)]
            sb.product_extd2(gpu.dll, jobs, 1, 9, 1, 16, 2, 41, 1)
            best = 1e9
            for rep in range(3):
                t0 = time.time(); r = sb.product_extd2(gpu.dll, jobs, 1, 9, 1, 16, 2, 41, 1); best = min(best, time.time() - t0)
            print(f"L={L} {kind:8s} {name:6s}: {best*1e3:7.2f} ms wall; zdropped={r[0]['zdropped']} n_cigar={len(r[0]['cigar'])} score={r[0]['score']}", flush=True)
