import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the suite forces the library's failure and fallback paths through the behaviour-changing hooks of include/hrfd_debug.h;
# they are inert in a process that did not ask for them (read once by libhrfd, so set before it is loaded)
os.environ.setdefault("HRFD_DEBUG_HOOKS", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests.reflib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own sources compiled by oracle/Makefile (oracle/_ref).
    Built in the dev container (where /root/reference exists) and shipped
    prebuilt to the GPU box; tests that need it skip when it is absent."""
    from tests import reflib
    if not reflib.have_ref():
        if os.path.isdir("/root/reference/radioDiags"):
            import subprocess
            subprocess.check_call(["make", "-C", reflib.ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return reflib.Ref()
