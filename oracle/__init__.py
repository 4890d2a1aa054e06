"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the MoTIF hot path.  It is the checker, never the product:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Pinning status -- what is pinned by RUNNING the reference and what only by READING it:
  * torch-level network restatement (`motif_ref.py`, `pwc_ref.py`): PINNED against outputs of the
    reference itself, imported in the build container under stubs (`tests/golden/make_golden.py`),
    committed as fixtures under `tests/golden/`.  In that import the reference's CUDA-only natives are
    replaced by `native_ref.c` -- the same C file this oracle calls -- so the `max|diff| = 0.0` of
    `restatement_vs_reference*.json` pins the network code AROUND the natives, not the natives.
  * RAFT correlation look-up (`alt_cuda_corr`, third party, binary only, no version pin): PINNED since round 3 by
    reference-run data that involve no oracle code -- `tests/golden/corrblock_16x24.npz` is the output of the
    reference's own pure-torch `CorrBlock` (`models/core/corr.py:8-56`) for queries inside, outside, on integers and
    on half pixels, `tests/golden/raft_corrblock_128x160.npz` the reference RAFT-small run with `alternate_corr=False`
    (`models/core/raft.py:44-45,104`).  `oracle.alt_corr_lookup`, the oracle RAFT, the HIP look-up and the HIP RAFT are
    all tested against them (`tests/test_oracle.py`, `tests/test_kernels_gpu.py`).
  * soft-splat x3 (cupy kernel strings), PWC 81-way correlation (cupy), DCNv2 im2col + GEMM (`src/cuda/*.cu`): restated
    from the kernel TEXT on both sides of every comparison (`native_ref.c`); they cannot be built here (no nvcc / THC).
    The reference holds one known-answer test for them (DCN zero-offset identity,
    `models/modules/DCNv2/test.py:32-67`), reproduced for the oracle and for the HIP kernel; it ships no other
    vector.  These three stages are the ones WITHOUT a reference-derived vector.
"""
