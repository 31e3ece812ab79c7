"""dr-nmf_amd: MI355X (gfx950) implementation of stwisdom/dr-nmf's hot path.

The directory name carries a hyphen, so it is imported through the top-level alias package
`drnmf_amd` (drnmf_amd/__init__.py points its __path__ here):

    from drnmf_amd import layers, ops
"""
__version__ = "0.1.0"
