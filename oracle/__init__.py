"""TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy) of the reference's DR-NMF hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``drnmf_amd``) never does.
"""
