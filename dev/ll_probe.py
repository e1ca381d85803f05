"""ksw_ll_i16 problems alone on the device: wall time of one launch (host path included) against the threads per group (PGA_LL_NT) and the groups per
problem (PGA_LL_GROUPS) of k_ll_multi (pga_ll.hip); G = 1 is the single-workgroup kernel.   gpurun -- 'python dev/ll_probe.py'"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes, numpy as np
import stagebind as sb
from pangraph_amd.synth import random_seq
dll = ctypes.CDLL(os.path.join(ROOT, "pangraph_amd", "libpgalign.so"))
rng = np.random.default_rng(1)
for (lq, lt, n) in ((10000, 10000, 1), (10000, 10000, 7), (6000, 6000, 1), (3500, 3500, 4), (10000, 10000, 40)):
    jobs = [(sb.nt4(random_seq(rng, lq).tobytes().decode()), sb.nt4(random_seq(rng, lt).tobytes().decode()), 0, 0, -1, 0x8000) for _ in range(n)]
    for nt in ("64", "256"):
        os.environ["PGA_LL_NT"] = nt
        line = f"{n} x {lq} x {lt}, {nt} threads per group:"
        for g in ("1", "16", "24", "32", "48", "64"):
            os.environ["PGA_LL_GROUPS"] = g
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); sb.product_extd2(dll, jobs, 1, 9, 1, 16, 2, 16, 2); best = min(best, time.perf_counter() - t0)
            line += f"  G={g}: {best * 1e3:6.2f}"
        print(line + " ms")
