import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (oracle/liboracle.so); built on demand with the committed Makefile."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def h2e_built():
    """libh2e.so must exist (built by __graft_entry__.build()); build it here if hipcc is around."""
    from halo2ecc_s_amd import build as b
    from halo2ecc_s_amd.engine import lib_path
    if not os.path.exists(lib_path()):
        b.build(verbose=False)
    return lib_path()


@pytest.fixture(scope="session")
def engine(h2e_built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from halo2ecc_s_amd import Engine
    return Engine(0)
