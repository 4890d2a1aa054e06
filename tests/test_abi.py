"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/motif_hip.h
declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from motif_amd.csrc import build
    build.build()
    from motif_amd import _lib
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "motif_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(motif_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from motif_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libmotif_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms, "ctypes binding and header disagree: %s" % (set(_lib.EXPORTS) ^ set(syms))
    assert lib.motif_abi_version() == 9 == _lib.ABI_VERSION


def test_library_is_gfx950_only():
    import subprocess
    so = os.path.join(ROOT, "motif_amd", "libmotif_hip.so")
    out = subprocess.run(["strings", so], capture_output=True, text=True).stdout
    assert "gfx950" in out
    for other in ("gfx942", "gfx90a", "sm_90"):
        assert other not in out


def test_argument_errors_do_not_need_a_gpu(lib):
    import ctypes
    from motif_amd._lib import MotifConvDesc
    d = MotifConvDesc()
    d.N, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW = 1, 8, 8, 6, 0, 4, 3, 3
    d.stride, d.pad, d.dil, d.groups = 1, 1, 1, 4            # 6 channels not divisible by 4 groups
    assert lib.motif_conv2d_packed_size(ctypes.byref(d)) < 0
    d.groups = 2
    assert lib.motif_conv2d_packed_size(ctypes.byref(d)) == 2 * 1 * 36 * 32   # groups*ncg*Kpad*WN, Kpad = 2*T*ceil(3/2) = 36
    assert lib.motif_splat_fwd(None, None, None, None, None, None, None, 1, 1, 1, 1, None) < 0
    assert lib.motif_siren_pack(None, None, (ctypes.c_int * 5)(67, 64, 64, 256, 3), 4, None, None) == 26244


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from motif_amd.models.modules.Ours import LunaTokis
    from motif_amd.data.synthetic import synthetic_sample
    s = synthetic_sample(32, 32, 4, 1)
    with pytest.raises(RuntimeError):
        LunaTokis().eval()(s["LQs"], None, s["time"], s["scale"], use_GT=False, iter=4)


def test_graft_entry_build_checks_the_library_it_built():
    """`__graft_entry__.build()` (the driver's "does it build" step) compiles every kernel source in-tree, loads the library and checks
    its ABI version and exports against the binding -- it once asserted a stale version number and failed for a whole round unnoticed."""
    import __graft_entry__ as g
    g.build()


def test_chain_entry_takes_the_trunk_shapes_and_refuses_the_others(lib):
    """`motif_conv2d_chain_ws_words` is host logic (no GPU): > 0 = the shape runs as one persistent launch, 0 = call the layers one by one."""
    import ctypes
    from motif_amd import _lib

    def words(n, c, h, w, L, mma=7, **kw):
        d = _lib.MotifConvDesc()
        d.N, d.H, d.W, d.C0, d.C1, d.Cout, d.KH, d.KW = n, h, w, c, 0, c, 3, 3
        d.stride, d.pad, d.dil, d.groups, d.pad_mode, d.mma = 1, 1, 1, 1, 0, mma
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.motif_conv2d_chain_ws_words(ctypes.byref(d), L)
    assert words(3, 64, 180, 320, 80) == 64 + 80 * 3 * 23          # ticket + abort + one word per (layer, image, tile row)
    assert words(2, 64, 180, 320, 10) == 64 + 10 * 2 * 23
    assert words(3, 64, 180, 320, 80, mma=6) == 0                   # two-part fp16 form only
    assert words(3, 128, 180, 320, 80) == 0 and words(3, 32, 180, 320, 80) == 0      # 49 .. 64 channels
    assert words(3, 64, 180, 322, 80) == 0                          # W % 4
    assert words(3, 64, 180, 320, 80, stride=2) == 0 and words(3, 64, 180, 320, 80, groups=2) == 0 and words(3, 64, 180, 320, 80, pad_mode=1) == 0
    assert words(3, 64, 180, 320, 300) == 0                         # the layer table lives in LDS
    assert words(16, 64, 540, 960, 80) == 0                         # so does the tile decode table
