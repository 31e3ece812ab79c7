"""Importable alias of the package directory `dr-nmf_amd/` (a hyphen is not a valid module name).

All code lives in ../dr-nmf_amd; this file only redirects the package search path so that
`import drnmf_amd.layers` loads `dr-nmf_amd/layers.py`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dr-nmf_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
