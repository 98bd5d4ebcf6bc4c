import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def require_ref(name="bgt"):
    """Path of a compiled-reference artefact (oracle/_ref/<name>, built by `make -C oracle ref` in the build container and
    shipped with the snapshot).  Its absence FAILS the test: a parity test that silently disappears is not a pass."""
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path) and os.path.exists("/root/reference/pbwt.c"):      # build container: build the checker
        import subprocess
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(path):
        pytest.fail("oracle/_ref/%s is missing: build it with `make -C oracle ref` where /root/reference exists "
                    "(it is git-ignored but travels to the GPU box with the snapshot)" % name)
    return path


@pytest.fixture(autouse=True)
def _kernel_choice_is_automatic_again():
    """Tests force kernel families through bgth_force_kernels (process-wide): every test starts and ends with the automatic choice."""
    yield
    hip_mod = sys.modules.get("bgt_amd.hip")
    if hip_mod is not None and getattr(hip_mod, "_forced", 0):
        hip_mod.force_kernels(0)
