"""SURVEY 8(f)-1 at merge scale: the member sequences of many blocks re-aligned onto their consensus on the GPU (pga_map_variations),
the CPU restatement (oracle/pgo_mapvar.c, one core) on a sample for comparison and as the check.
usage: dev/f1_bench.py [n_blocks] [members] [block_len] [band_width]"""
import sys, os, time, json, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import mapvarbind as mb

if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    members = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    bw = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    rng = np.random.default_rng(20260928)
    dll = C.CDLL(os.path.join(ROOT, "pangraph_amd", "libpgalign.so"))
    odll = C.CDLL(os.path.join(ROOT, "oracle", "libpgoracle.so"))
    jobs = []
    for b in range(nb):
        ref = mb.random_seq(rng, int(L * rng.uniform(0.5, 1.5)))
        arr = np.frombuffer(ref.encode(), dtype=np.uint8)
        for m in range(members):                                         # cheap mutation: SNPs + a few short indels (vectorised)
            a = arr.copy()
            k = rng.random(len(a)) < 0.01
            a[k] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(k.sum()))]
            s = a.tobytes().decode()
            for _ in range(int(rng.integers(0, 4))):
                p = int(rng.integers(0, len(s))); n = int(rng.integers(1, bw))
                s = s[:p] + s[p + n:] if rng.random() < 0.5 else s[:p] + mb.random_seq(rng, n) + s[p:]
            jobs.append((ref, s, 0, bw))
    bases = sum(len(j[1]) for j in jobs)
    mb.product_map_variations(dll, jobs[:64])                            # warm-up (allocator, module load)
    t0 = time.time(); got = mb.product_map_variations(dll, jobs); t_gpu = time.time() - t0
    full = os.environ.get("F1_FULL_CHECK") == "1"                        # every job against the CPU restatement (one core: minutes)
    idx = list(range(len(jobs))) if full else list(range(0, len(jobs), max(1, len(jobs) // 200)))
    t0 = time.time(); exp = [mb.oracle_map_variations(odll, *jobs[i]) for i in idx]; t_cpu = time.time() - t0
    keys = ("status", "score", "attempts", "hit_boundary", "subs", "dels", "inss")
    same = all({k: got[i][k] for k in keys} == {k: e[k] for k in keys} for i, e in zip(idx, exp))
    cpu_bases = sum(len(jobs[i][1]) for i in idx)
    print(json.dumps(dict(jobs=len(jobs), member_Mbp=bases / 1e6, gpu_s=round(t_gpu, 3), gpu_Mbp_s=round(bases / 1e6 / t_gpu, 1), gpu_jobs_s=round(len(jobs) / t_gpu),
                          cpu_sample_jobs=len(idx), n_identical=sum({k: got[i][k] for k in keys} == {k: e[k] for k in keys} for i, e in zip(idx, exp)), cpu_s=round(t_cpu, 3), cpu_Mbp_s_1core=round(cpu_bases / 1e6 / t_cpu, 2), identical_on_sample=same,
                          retried=sum(g["attempts"] > 1 for g in got), note="gpu_s includes the ctypes packing / unpacking of the test binding")))
