"""The sort-free index build (pangraph_amd/csrc/pga_index_buckets.h: a candidate route, PGA_INDEX_BUCKETS=1) and the sort-free mid_occ statistic
(pga_maxocc_hist.h: PGA_MAXOCC_HIST=1; checked on the key table the index build produced, against the sorted counts, at three fractions) under
dev/emu/hip_emu.h -- the product's kernels, every workgroup as fibers on the host, against a std::map: every minimizer finds, through its key id,
exactly the occurrence words of its (group, hash), ascending (what mm_idx_get returns: packages/minimap2-sys/minimap2/index.c:84-98,252).
CPU-only: this checks the kernels' logic (indexing, barriers, the overflow report), not that they are fast or that hipcc's code is right."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def emu_bins(tmp_path_factory):
    d = tmp_path_factory.mktemp("emu")
    src = os.path.join(ROOT, "tests", "emu", "index_buckets_emu.cpp")
    out = {}
    for tag, extra in (("lds", []), ("global", ["-DPGA_IXB_HB=16"])):
        exe = str(d / f"ixb_{tag}")
        subprocess.run(["g++", "-O1", "-std=c++17", "-DPGA_EMU", "-Wall", "-Werror", "-o", exe, src] + extra, check=True, capture_output=True, text=True)
        out[tag] = exe
    return out


@pytest.mark.parametrize("tag,args,expect", [
    ("lds", ["1", "7", "3000", "2000", "0"], "keys"),                       # a few groups, one or two tiles, buckets of one group side by side
    ("lds", ["2", "5", "120000", "90000", "0"], "keys"),                    # groups of up to 360 k minimizers: hundreds of buckets per group, a dozen tiles
    ("lds", ["6", "40", "2500", "40", "0"], "keys"),                        # few distinct k-mers per group: long lists, buckets near their cap
    ("lds", ["4", "6", "3000", "2000", "2500"], "beyond the histogram"),    # one k-mer 2 500 times in one group: a list of thousands inside one bucket; a count the mid_occ histogram does not resolve
    ("lds", ["5", "6", "20000", "50", "6000"], "overflow reported"),        # ... 6 000 times: the bucket cannot be sorted in LDS, the scan says so
    ("global", ["1", "7", "3000", "2000", "0"], "keys"),                    # 16 histogram bins: tiles span more buckets than the histogram holds
    ("global", ["7", "300", "60", "30", "0"], "keys"),                      # hundreds of small groups, empty ones among them
])
def test_bucket_index_under_emulation(emu_bins, tag, args, expect):
    r = subprocess.run([emu_bins[tag]] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:") and expect in r.stdout, r.stdout + r.stderr


def test_workgroup_sort_under_emulation(tmp_path):
    """pga_wg_sort.h (candidate, PGA_WG_SORT=1: the small sorts of the chaining stage in one launch) against std::stable_sort on the compared key
    bits -- the order rocprim::radix_sort_pairs(..., 0, end_bit) gives: sizes around the powers of two and at the cap, 32- and 64-bit keys, few
    distinct keys (stability), partial bit ranges."""
    exe = str(tmp_path / "wgs")
    subprocess.run(["g++", "-O1", "-std=c++17", "-DPGA_EMU", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "emu", "wg_sort_emu.cpp")], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_emulator_reports_a_divergent_barrier(tmp_path):
    """hip_emu.h itself: threads that leave a kernel while others of the workgroup wait at a barrier are an error, not a silent pass"""
    src = tmp_path / "div.cpp"
    src.write_text('#include "%s"\n__global__ void k(int *p) { if (threadIdx.x & 1) return; __syncthreads(); p[threadIdx.x] = 1; }\n'
                   'int main() { int p[8] = {0}; emu_launch(dim3(1), dim3(8), [&] { k(p); }); return 0; }\n' % os.path.join(ROOT, "dev", "emu", "hip_emu.h"))
    exe = str(tmp_path / "div")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, str(src)], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "wait at a barrier" in r.stderr
