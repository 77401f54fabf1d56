"""CPU oracle for the PVR-embedding + BC hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  The product (`pvr_habitat_amd/`) never imports it and has no CPU fallback: it fails
loudly when the HIP library is missing.
"""
