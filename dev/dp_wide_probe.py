"""Timing probe for the wide DP kernel on early z-dropping end extensions (class 3 of dp_run)."""
import sys, os, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
faulthandler.dump_traceback_later(240, exit=True)
import numpy as np
import stagebind as sb
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
rng = np.random.default_rng(7)
def job(L, w, fl, hom=100):
    t = random_seq(rng, L); q = random_seq(rng, L); q[:hom] = t[:hom]
    return (sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), w, 200, -1, fl)
for (L, w, fl, n) in [(9600, 2873, 0x40, 1), (9600, 2873, 0x40, 256), (9600, 2873, 0x40 | 0x08 | 0x10, 256), (4800, 2873, 0x40, 256), (2000, 1501, 0x40, 256), (9600, 2873, 0x40, 1024)]:
    jobs = [job(L, w, fl) for _ in range(min(n, 64))]
    jobs = (jobs * ((n + len(jobs) - 1) // len(jobs)))[:n]
    sb.product_extd2(gpu.dll, jobs[:1], 1, 9, 1, 16, 2, 41, 1)
    t0 = time.time(); r = sb.product_extd2(gpu.dll, jobs, 1, 9, 1, 16, 2, 41, 1); dt = time.time() - t0
    print(f"L={L} w={w} flag={fl:#x} n={n}: {dt*1e3:.1f} ms wall; zdropped={sum(x['zdropped'] for x in r)} max_q~{np.mean([x['max_q'] for x in r]):.0f}", flush=True)
