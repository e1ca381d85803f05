"""Two routes of the product that replace rocPRIM sorts -- pga_maxocc_hist.h (mm_idx_cal_max_occ of every group from per-group histograms of the occurrence
counts; packages/minimap2-sys/minimap2/index.c:186-207) and pga_wg_sort.h (the chaining stage's sorts of at most 4 096 pairs in one launch) -- under
dev/emu/hip_emu.h: the product's kernels, every workgroup as fibers on the host, against the sorted counts / std::stable_sort.
CPU-only: this checks the kernels' logic (indexing, barriers, the "unresolved" report), not that they are fast or that hipcc's code is right; on the device
both are held against the rocPRIM routes by tests/test_gpu_zz_candidates.py and, through the records, by every digest test of the suite."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def mo_bin(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "mo")
    subprocess.run(["g++", "-O1", "-std=c++17", "-DPGA_EMU", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "emu", "maxocc_hist_emu.cpp")], check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("args,expect", [
    (["1", "7", "3000", "0"], "keys"),                        # a few groups, empty and tiny ones among them
    (["2", "5", "120000", "0"], "keys"),                      # groups of up to 120 k keys: dozens of workgroups per group
    (["7", "300", "60", "0"], "keys"),                        # hundreds of small groups: workgroups that span many groups
    (["4", "6", "3000", "2500"], "keys"),                     # one count beyond the histogram that is NOT the answer at any fraction
    (["5", "2", "3", "2500"], "beyond the histogram"),        # ... and a group of three keys whose answer it is: unresolved, the caller sorts
])
def test_mid_occ_histograms_under_emulation(mo_bin, args, expect):
    r = subprocess.run([mo_bin] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:") and expect in r.stdout, r.stdout + r.stderr


def test_workgroup_sort_under_emulation(tmp_path):
    """pga_wg_sort.h (the small sorts of the chaining stage in one launch) against std::stable_sort on the compared key
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
