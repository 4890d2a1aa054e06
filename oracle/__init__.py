"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the MoTIF hot path.  It is the checker, never the product:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Pinning status:
  * torch-level network restatement (`motif_ref.py`, `pwc_ref.py`): PINNED against outputs of the
    reference itself, imported in the build container under stubs (`tests/golden/make_golden.py`),
    committed as fixtures under `tests/golden/`.
  * native kernels whose CUDA text is in-repo (soft-splat x3, PWC correlation, DCNv2): restated from
    the kernel text (`native_ref.c`); the reference holds one known-answer test for them (DCN
    zero-offset identity, `models/modules/DCNv2/test.py:32-67`), which is reproduced in
    `tests/test_oracle.py`.  The reference ships no other vector for these kernels.
  * `alt_cuda_corr`: third-party, not vendored, no version pin -> PARITY UNPINNED at that boundary;
    anchored on the in-repo equivalent `CorrBlock` (`models/core/corr.py:8-56`).
"""
