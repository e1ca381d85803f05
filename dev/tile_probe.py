"""Timing probe for the register-resident tile kernel (class 0 of dp_run): N first-pass gap fills of ~210 x 210."""
import sys, os, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
faulthandler.dump_traceback_later(300, exit=True)
import numpy as np
import stagebind as sb
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq, mutate
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
rng = np.random.default_rng(11)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
base = []
for _ in range(512):
    L = int(rng.integers(200, 232))
    t = random_seq(rng, L); q = mutate(rng, t, snp=float(os.environ.get("SNP", "0.03")), indel=float(os.environ.get("INDEL", "0.003")))
    base.append((sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), 150001, 200, -1, 0x08))
jobs = (base * (N // len(base) + 1))[:N]
sb.product_extd2(gpu.dll, jobs[:1000], 1, 9, 1, 16, 2, 41, 1)
t0 = time.time(); r = sb.product_extd2(gpu.dll, jobs, 1, 9, 1, 16, 2, 41, 1); dt = time.time() - t0
print(f"{N} tiles: {dt*1e3:.1f} ms wall incl. host packing")
