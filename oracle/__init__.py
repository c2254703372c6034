"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the predict-and-recompute CG hot path.

Nothing in ``new_cg_variants_amd`` (the product) may import this package.  The
only legitimate users are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``, and there only as the checker / the CPU
line, never as the thing shipped.
"""
