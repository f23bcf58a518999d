import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _ensure_built():
    lib = os.path.join(ROOT, "halo2-plonky2-verifier_amd", "libh2w.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call([os.path.join(ROOT, "build.sh")])


@pytest.fixture(scope="session")
def h2w():
    """The product package (hyphenated directory name -> importlib)."""
    _ensure_built()
    mod = importlib.import_module("halo2-plonky2-verifier_amd")
    sys.modules.setdefault("h2w_amd", mod)
    return mod


@pytest.fixture(scope="session")
def h2w_api(h2w):
    return importlib.import_module("halo2-plonky2-verifier_amd.api")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def consts(h2w, oracle):
    ko = oracle.synth_consts(0xC0FFEE)
    kh = h2w.PoseidonConsts.from_buffer_copy(bytes(ko))
    return ko, kh


@pytest.fixture(scope="session")
def published(h2w, oracle):
    """The published Poseidon parameter sets: the oracle's copy from tests/golden/poseidon_published.json, the product's own
    from h2w_poseidon_published (tests/test_poseidon_published.py pins both to published known-answer vectors)."""
    return oracle.published_consts(), h2w.published_consts()
