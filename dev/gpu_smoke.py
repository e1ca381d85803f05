import sys, time, gzip, re
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT)
from pangraph_amd.mm2ffi import *
from pangraph_amd.synth import evolve_population
def read_fa(path):
    op = gzip.open if path.endswith('.gz') else open
    names, seqs = [], []
    with op(path, 'rt') as f:
        for line in f:
            line=line.strip()
            if not line: continue
            if line.startswith('>'): names.append(line[1:].split()[0]); seqs.append([])
            else: seqs[-1].append(line)
    return names, [''.join(s) for s in seqs]
ref = Mm2Lib('oracle/_ref/libmm2ref.so')
gpu = Mm2Lib('pangraph_amd/libpgalign.so')
def cmp(tag, seqs, names, **kw):
    t0=time.time(); a = ref.align_all(seqs,names,**kw); t1=time.time()
    b = gpu.align_all(seqs,names,**kw); t2=time.time()
    ka=[x.key() for x in a]; kb=[x.key() for x in b]
    print(tag, len(a), len(b), 'MATCH' if ka==kb else 'DIFF', 'ref %.2fs gpu %.2fs'%(t1-t0,t2-t1), flush=True)
    if ka!=kb:
        for i,(x,y) in enumerate(zip(ka,kb)):
            if x!=y: print(' ref',i,x[:14], x[14][:80]); print(' gpu',i,y[:14], y[14][:80]); break
seqs = evolve_population(1, 2, 3000, snp=0.02, indel=0.002, n_inv=0, n_ins=0, n_del=0)
cmp('tiny2x3k', seqs, ['10','9'])
cmp('tiny2x3k_k10', seqs, ['10','9'], sensitivity=20, kmer_length=10)
seqs = evolve_population(2, 4, 50000, snp=0.01, max_event=5000)
cmp('4x50k', seqs, [str(100+i) for i in range(4)])
names, seqs = read_fa('tests/golden/plasmids.fa.gz')
names = [str(1000+i*7919) for i in range(len(names))]
cmp('plasmids4', seqs[:4], names[:4])
cmp('plasmids15', seqs, names)
