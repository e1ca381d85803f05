import sys, os, faulthandler
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
import stagebind as sb
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq, mutate
gpu = Mm2Lib('pangraph_amd/libpgalign.so'); ref = Mm2Lib('oracle/_ref/libmm2ref.so')
rng = np.random.default_rng(1)
mat = sb.simple_mat(1,9,1)
cases = [(int(a), int(b), w, zd, eb, fl) for (a,b,w,zd,eb,fl) in [(50,50,150001,200,-1,0x08),(50,60,150001,200,-1,0),(207,207,150001,200,-1,0x08),(207,215,150001,200,-1,0),(300,280,1501,200,-1,0x40),(100,90,1501,200,-1,0x40|0x02|0x80),(400,500,150001,200,-1,0x08),(64,64,150001,200,-1,0x08),(65,1,150001,200,-1,0x08),(1,65,150001,200,-1,0)]]
which = sys.argv[1:] and [int(x) for x in sys.argv[1:]] or range(len(cases))
for ci in which:
    L,Lq,w,zd,eb,fl = cases[ci]
    t = random_seq(rng, L); q = mutate(rng, t, snp=0.03, indel=0.01)[:Lq]
    if len(q) < Lq: q = np.concatenate([q, random_seq(rng, Lq-len(q))])
    qn, tn = sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode())
    print('case', ci, L, Lq, w, hex(fl), flush=True)
    g = sb.product_extd2(gpu.dll, [(qn,tn,w,zd,eb,fl)], 1,9,1,16,2,41,1)[0]
    e = sb.ref_extd2(ref.dll, qn, tn, mat, 16,2,41,1, w, zd, eb, fl)
    keys = ["zdropped","reach_end","cigar","score"] + ([] if fl&8 else ["max","max_q","max_t","mqe","mqe_t","mte","mte_q"])
    bad = [k for k in keys if g[k]!=e[k]]
    print('  ', 'OK' if not bad else ('DIFF '+str(bad)+' got '+str({k:g[k] for k in bad})+' exp '+str({k:e[k] for k in bad})), flush=True)
