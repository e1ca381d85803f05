"""One long banded end extension that runs to the end (the last tail round of the bench), through the workgroup kernel: L w flag"""
import sys, os, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
faulthandler.dump_traceback_later(60, exit=True)
import numpy as np
import stagebind as sb
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq, mutate
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
rng = np.random.default_rng(3)
L = int(sys.argv[1]); w = int(sys.argv[2]); fl = int(sys.argv[3], 0)
t = random_seq(rng, L); q = mutate(rng, t, snp=0.02, indel=0.002)
job = (sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), w, 200, -1, fl)
sb.product_extd2(gpu.dll, [job], 1, 9, 1, 16, 2, 41, 1)
t0 = time.time(); r = sb.product_extd2(gpu.dll, [job], 1, 9, 1, 16, 2, 41, 1); dt = time.time() - t0
print(f"L={L} w={w} flag={fl:#x}: {dt*1e3:.1f} ms wall, score {r[0]['score']} max {r[0]['max']} zdropped {r[0]['zdropped']}", flush=True)
