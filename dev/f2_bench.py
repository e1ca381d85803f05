"""SURVEY 8(f)-2 at the BASELINE size: split + filter of the match list of every wave of the C5 build on the GPU (pga_result_filter), timed
per wave; the CPU restatement (oracle/pgo_filter.c) on the same lists for comparison and as the check.  usage: dev/f2_bench.py [n_genomes] [length]"""
import sys, os, time, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from pangraph_amd import levels, batch
from pangraph_amd.batch import pga_match_t
import filterbind as fb

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
    pop = levels.Population(20260928, n, L)
    dll = batch.lib()
    odll = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "libpgoracle.so"))
    dll.pga_result_filter.restype = C.c_int
    dll.pga_result_filter.argtypes = [C.c_void_p, C.POINTER(fb.FilterParams), C.POINTER(C.c_void_p)]
    dll.pga_result_n_matches.restype = C.c_int64; dll.pga_result_n_matches.argtypes = [C.c_void_p]
    dll.pga_result_matches.restype = C.POINTER(pga_match_t); dll.pga_result_matches.argtypes = [C.c_void_p]
    dll.pga_result_cigars.restype = C.POINTER(C.c_uint32); dll.pga_result_cigars.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    odll.pgo_split_filter.restype = C.c_int64
    odll.pgo_split_filter.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.POINTER(pga_match_t)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
    fp = fb.FilterParams(100, 3, 100.0, 10.0)
    tot_in = tot_out = 0; t_gpu = t_cpu = 0.0; ops_in = 0; same = True
    for label, groups, names in pop.build_waves():
        rb = batch.ResidentBatch(batch.PreparedBatch(groups, names))
        res = rb.align(want_raw=True)
        h = res._handle
        n_in = dll.pga_result_n_matches(h)
        nops = C.c_uint64(0); cg = dll.pga_result_cigars(h, C.byref(nops))
        out = C.c_void_p()
        t0 = time.time()
        assert dll.pga_result_filter(h, C.byref(fp), C.byref(out)) == 0
        t_gpu += time.time() - t0
        n_out = dll.pga_result_n_matches(out)
        om = C.POINTER(pga_match_t)(); oc = C.POINTER(C.c_uint32)(); on = C.c_uint64(0)
        t0 = time.time()
        n_ref = odll.pgo_split_filter(n_in, dll.pga_result_matches(h), cg, 100, 100.0, 10.0, 3, C.byref(om), C.byref(oc), C.byref(on))
        t_cpu += time.time() - t0
        if n_ref != n_out:
            same = False
        else:
            gm = dll.pga_result_matches(out)
            a = np.ctypeslib.as_array(C.cast(gm, C.POINTER(C.c_uint8)), shape=(n_out * C.sizeof(pga_match_t),)) if n_out else np.zeros(0, np.uint8)
            b = np.ctypeslib.as_array(C.cast(om, C.POINTER(C.c_uint8)), shape=(n_out * C.sizeof(pga_match_t),)) if n_out else np.zeros(0, np.uint8)
            gn = C.c_uint64(0); gc = dll.pga_result_cigars(out, C.byref(gn))
            same &= bool((a == b).all()) and gn.value == on.value and all(gc[i] == oc[i] for i in range(0, on.value, max(1, on.value // 2000)))
        tot_in += n_in; tot_out += n_out; ops_in += nops.value
        dll.pga_result_free(out); res.close(); rb.close()
    print(json.dumps({"row": "8(f)-2 split + filter", "genomes": n, "matches_in": tot_in, "cigar_ops_in": ops_in, "accepted_out": tot_out, "gpu_s_all_waves": round(t_gpu, 3),
                      "cpu_port_s_all_waves": round(t_cpu, 3), "identical_to_cpu_port": same}))
