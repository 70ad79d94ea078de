"""CPU oracle for the volumetric un-projection path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
multiviewhmr_amd/ imports it (tests/test_boundary.py checks that).  See unproject_oracle.c for
the reference file:line map and how parity is pinned.
"""
