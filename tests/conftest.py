"""pytest configuration: markers, paths, and one-time builds of the checkers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what the tests load; hipcc cross-compiles here, nothing needs a GPU
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    if not os.path.exists(os.path.join(ROOT, "zsc_amd", "libzsc_hip.so")) or os.path.isdir("/opt/rocm/bin"):
        try:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "zsc_amd", "csrc")], check=True)
        except Exception as exc:  # pragma: no cover
            print("warning: could not (re)build libzsc_hip.so:", exc)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle_py import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libzsc_ref.so not present (needs /root/reference to build)")
    return Reference()


def gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _flush_c_stdio():
    """The compiled reference reports through printf (its own example configuration header,
    /root/reference/test/zsc_test_private.h:73-81): flush C stdio while pytest still captures the test's
    output, so that those lines stay with their test instead of pouring out when the process exits."""
    yield
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
