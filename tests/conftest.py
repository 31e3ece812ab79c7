import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_numpy_golden.npz"))


@pytest.fixture
def cell_form(request, monkeypatch):
    """The recurrent cell has two forms (csrc/cell_gram.h): the factored pair of contractions and,
    for small dictionaries, the Gram form.  'auto' leaves the library's rule in charge; 'factored'
    forces the factored kernels (DRNMF_GRAM=0, read by libdrnmf at every call) so that the small
    parity shapes keep exercising them too."""
    if request.param == "factored":
        monkeypatch.setenv("DRNMF_GRAM", "0")
    else:
        monkeypatch.delenv("DRNMF_GRAM", raising=False)
    return request.param
