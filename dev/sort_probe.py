"""Which digit structure makes the sort replay slow?  Synthetic anchor-position arrays through pga_stage_sort under PGA_VERBOSE."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dll = C.CDLL(os.path.join(ROOT, "pangraph_amd", "libpgalign.so"))
dll.pga_stage_sort.restype = C.c_int
dll.pga_stage_sort.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
if len(sys.argv) > 1:
    raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2)
    off = np.array([0, len(raw)], dtype=np.uint64)
    x = raw[:, 0].copy()
    print("loaded", len(raw), "anchors; strands", np.bincount((x >> np.uint64(63)).astype(np.int64)), "targets", np.unique((x >> np.uint64(32)) & np.uint64(0x7fffffff)), flush=True)
    for rep in range(2):
        xy = raw.copy()
        t = time.time(); dll.pga_stage_sort(1, off.ctypes.data, xy.ctypes.data); print(f"### real array: {(time.time()-t)*1e3:.1f} ms", flush=True)
    # the same keys, target-1 forward-strand part only, positions only
    sel = raw[(x >> np.uint64(63) == 0) & (((x >> np.uint64(32)) & np.uint64(0x7fffffff)) == 1)]
    for lab, arr in (("target 1, forward strand, raw order", sel),):
        off2 = np.array([0, len(arr)], dtype=np.uint64)
        for rep in range(2):
            xy = arr.copy()
            t = time.time(); dll.pga_stage_sort(1, off2.ctypes.data, xy.ctypes.data); print(f"### {lab} ({len(arr)}): {(time.time()-t)*1e3:.1f} ms", flush=True)
    d = (sel[:, 0] & np.uint64(0xffffffff)) >> np.uint64(16)
    ch = np.flatnonzero(np.diff(d.astype(np.int64)) != 0)
    print("digit runs at shift 16:", len(ch) + 1, "first changes at", ch[:30], "digits", d[ch[:30]], flush=True)
    sys.exit(0)
rng = np.random.default_rng(1)
n = 550_000
pos = np.sort(rng.choice(5_200_000, size=n, replace=False)).astype(np.uint64)
def run(label, x):
    xy = np.zeros((len(x), 2), dtype=np.uint64); xy[:, 0] = x; xy[:, 1] = np.arange(len(x), dtype=np.uint64)
    off = np.array([0, len(x)], dtype=np.uint64)
    dll.pga_stage_sort(1, off.ctypes.data, xy.ctypes.data)     # warm-up
    xy[:, 0] = x; xy[:, 1] = np.arange(len(x), dtype=np.uint64)
    t = time.time(); dll.pga_stage_sort(1, off.ctypes.data, xy.ctypes.data); dt = time.time() - t
    assert (np.diff(xy[:, 0].astype(np.int64)) >= 0).all()
    print(f"### {label}: {dt*1e3:.1f} ms", flush=True)
run("sorted", pos)
run("rotated by 40%", np.roll(pos, int(0.4 * n)))
noise = np.roll(pos, int(0.4 * n)).copy(); k = rng.choice(n, size=n // 100, replace=False); noise[k] = rng.integers(0, 5_200_000, size=len(k)).astype(np.uint64)
run("rotated + 1% noise", noise)
d = np.roll(pos, int(0.4 * n)).copy(); j = rng.choice(n - 300, size=8, replace=False)
for a in j: d[a:a + 150] = d[(a + 100000) % (n - 200):(a + 100000) % (n - 200) + 150]      # eight duplicated stretches (equal keys)
run("rotated + 8 duplicated stretches", d)
inv = np.roll(pos, int(0.4 * n)).copy(); inv[100000:130000] = inv[100000:130000][::-1].copy()
run("rotated + one inverted stretch", inv)
