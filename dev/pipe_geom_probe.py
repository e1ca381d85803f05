"""a few geometries through the workgroup pipeline under PGA_VERBOSE: does it hand any of them back?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import stagebind as sb
from pangraph_amd import batch
rng = np.random.default_rng(3)
def rs(n): return rng.integers(0, 4, n).astype(np.uint8)
jobs = []
t = rs(1073); q = t[400:607].copy(); jobs.append((q, t, 150001, 200, -1, 0))          # short query inside a long target, unbanded exact fill
t = rs(1073); q = rs(207); jobs.append((q, t, 150001, 200, -1, 0))
t = rs(3000); q = t[:1500].copy(); jobs.append((q, t, 1501, 200, -1, 0x40))
t = rs(900); q = rs(5000); jobs.append((q, t, 1501, 200, -1, 0x40))
os.environ["PGA_PIPE"] = "force"; os.environ["PGA_BSTRIPS"] = "off"; os.environ["PGA_VERBOSE"] = "1"
out = sb.product_extd2(batch.lib(), jobs, 1, 9, 1, 16, 2, 41, 1)
for j, o in zip(jobs, out): print(len(j[0]), len(j[1]), j[2], hex(j[5]), "->", o["zdropped"], o["score"], o["max"], len(o["cigar"]))
