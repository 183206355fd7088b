import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    # the oracle is the checker; building it is not using it
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], check=True)


@pytest.fixture(scope="session")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dwt_amd

    c = dwt_amd.Context(0)
    yield c
    c.close()


@pytest.fixture
def opts(ctx):
    """Diagnostic switches of the shared context (enum dwtx_option, include/dwtx.h), back to 0 after the test."""
    used = []

    class Opts:
        def set(self, name, value=1):
            used.append(name)
            ctx.set_option(name, value)

    yield Opts()
    for name in used:
        ctx.set_option(name, 0)
