import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _reload_drnmf_env():
    """libdrnmf reads its DRNMF_* tuning variables ONCE per process (drnmf_reload_env retakes the
    snapshot): a test that flips one must tell the library -- only if it is loaded already."""
    try:
        from drnmf_amd import _capi
    except Exception:
        return
    if _capi._lib is not None:
        _capi._lib.drnmf_reload_env()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with setenv / delenv of a DRNMF_* variable followed by drnmf_reload_env."""
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        orig_set(name, value, *a, **k)
        if name.startswith("DRNMF_"):
            _reload_drnmf_env()

    def delenv(name, *a, **k):
        orig_del(name, *a, **k)
        if name.startswith("DRNMF_"):
            _reload_drnmf_env()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch


@pytest.fixture(autouse=True)
def _drnmf_env_snapshot():
    """Set up first / torn down last: whatever a test's monkeypatch changed has been undone by then."""
    yield
    _reload_drnmf_env()


@pytest.fixture(scope="session", autouse=True)
def _drnmf_matrix_mode_of_the_session():
    """DRNMF_TEST_MATRIX_MODE=bf16x3 runs the WHOLE GPU suite with the frame-parallel products in the
    split-operand mode (include/drnmf.h DRNMF_MATRIX_BF16X3) -- every tolerance unchanged.  Unset: the default
    exact-fp32 mode; tests/test_gpu_x3.py covers the mode either way."""
    mode = os.environ.get("DRNMF_TEST_MATRIX_MODE")
    if mode:
        import torch
        if torch.cuda.is_available():
            from drnmf_amd import ops
            ops.set_matrix_mode(mode, 0)
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_numpy_golden.npz"))


@pytest.fixture
def cell_form(request, monkeypatch):
    """The recurrent cell has two forms (csrc/cell_gram.h): the factored pair of contractions and,
    for small dictionaries, the Gram form.  'auto' leaves the library's rule in charge; 'factored'
    forces the factored kernels (DRNMF_GRAM=0; the monkeypatch fixture above makes libdrnmf re-read it) so that the small
    parity shapes keep exercising them too."""
    if request.param == "factored":
        monkeypatch.setenv("DRNMF_GRAM", "0")
    else:
        monkeypatch.delenv("DRNMF_GRAM", raising=False)
    return request.param
