"""A C program compiled against the REFERENCE'S OWN headers and linked to libzsc_hip.so
(tests/c_caller): the drop-in boundary as a C user sees it -- prototypes, enum widths and
gz_header's layout are checked by a compiler, not by reading."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "c_caller", "caller_ref")


@pytest.mark.gpu
def test_c_caller_built_against_reference_headers():
    if not os.path.exists(BIN):
        pytest.skip("tests/c_caller/caller_ref was not built (needs /root/reference at build time)")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(HERE, "..", "zsc_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.startswith("OK "), (r.returncode, r.stdout, r.stderr[-2000:])
