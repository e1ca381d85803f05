import sys, time, os
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT)
os.environ['PGA_VERBOSE']='1'
from pangraph_amd.mm2ffi import *
from pangraph_amd.synth import evolve_population
ref = Mm2Lib('oracle/_ref/libmm2ref.so')
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
def run(tag, seqs, names, check=True, **kw):
    t1=time.time(); b = gpu.align_all(seqs,names,**kw); t2=time.time()
    msg = ''
    if check:
        t0=time.time(); a = ref.align_all(seqs,names,**kw); t3=time.time()
        msg = ('MATCH' if [x.key() for x in a]==[x.key() for x in b] else 'DIFF') + ' ref %.2fs'%(t3-t0)
    print(tag, len(b), msg, 'gpu %.2fs'%(t2-t1), flush=True)
L = int(sys.argv[1]) if len(sys.argv)>1 else 1000000
seqs = evolve_population(3, 2, L, snp=0.01, indel=0.001, n_inv=2, n_ins=4, n_del=2, max_event=30000)
run('warm', seqs[:2], ['1','2'], check=False)
run('2x%d'%L, seqs, ['1','2'])
