"""Shared fixtures.  Tests marked `gpu` need an MI355X and call the product through its C-ABI; everything else
runs on CPU: the oracle against golden vectors and against the compiled reference, host logic, ABI checks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PRODUCT_SO = os.path.join(ROOT, "pangraph_amd", "libpgalign.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "libpgoracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmm2ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _make(path, *targets):
    subprocess.run(["make", "-C", path, *targets], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle_lib():
    from pangraph_amd.mm2ffi import Mm2Lib
    if not os.path.exists(ORACLE_SO):
        _make(os.path.join(ROOT, "oracle"), "oracle")
    return Mm2Lib(ORACLE_SO)


@pytest.fixture(scope="session")
def ref_lib():
    """The reference's own C, compiled by oracle/Makefile (prebuilt .so on the GPU box)."""
    from pangraph_amd.mm2ffi import Mm2Lib
    if not os.path.exists(REF_SO):
        if os.path.isdir("/root/reference"):
            _make(os.path.join(ROOT, "oracle"), "ref")
        else:
            pytest.skip("oracle/_ref/libmm2ref.so not built and /root/reference absent")
    return Mm2Lib(REF_SO)


@pytest.fixture(scope="session")
def product_so():
    if not os.path.exists(PRODUCT_SO):
        _make(os.path.join(ROOT, "pangraph_amd", "csrc"))
    return PRODUCT_SO


@pytest.fixture(scope="session")
def gpu_lib(product_so):
    """The product.  No fallback: if the HIP library cannot see a device the test FAILS."""
    from pangraph_amd.mm2ffi import Mm2Lib
    import ctypes
    lib = Mm2Lib(product_so)
    lib.dll.pga_device_count.restype = ctypes.c_int
    assert lib.dll.pga_device_count() > 0, "libpgalign.so sees no HIP device"
    return lib
