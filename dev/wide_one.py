"""One large unbanded exact-mode problem through the workgroup DP kernel (profiling target)."""
import sys, os, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
faulthandler.dump_traceback_later(240, exit=True)
import numpy as np
import stagebind as sb
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq, mutate
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
rng = np.random.default_rng(3)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
fl = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
t = random_seq(rng, L); q = mutate(rng, t, snp=0.02, indel=0.002)
job = (sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), 150001, -1, -1, fl)
sb.product_extd2(gpu.dll, [job], 1, 9, 1, 16, 2, 41, 1)
t0 = time.time(); r = sb.product_extd2(gpu.dll, [job] * int(os.environ.get("NJOBS", "1")), 1, 9, 1, 16, 2, 41, 1); dt = time.time() - t0
print(f"L={L} flag={fl:#x}: {dt*1e3:.1f} ms wall, score {r[0]['score']}, n_cigar {len(r[0]['cigar'])}")
