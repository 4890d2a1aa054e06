"""ctypes binding of libmotif_hip.so (the C ABI declared in include/motif_hip.h).

The product path has NO fallback: if the shared library cannot be had or a call fails, a RuntimeError is
raised (SURVEY.md §8(b) "Errors").  A missing libmotif_hip.so (fresh checkout: binaries are not in git) is compiled
in-tree with hipcc on first use (`motif_amd/csrc/build.py`, gfx950); `load(build=True)` forces the incremental build.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MOTIF_HIP_LIB") or os.path.join(_HERE, "libmotif_hip.so")   # env: instrumented debug builds
_lib = None


class MotifConvDesc(Structure):
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("C0", c_int), ("C1", c_int),
                ("Cout", c_int), ("KH", c_int), ("KW", c_int),
                ("stride", c_int), ("pad", c_int), ("dil", c_int), ("groups", c_int),
                ("pad_mode", c_int), ("act", c_int), ("act2", c_int), ("act_split", c_int), ("res_mode", c_int),
                ("in0_bs", c_long), ("in1_bs", c_long), ("res_bs", c_long), ("out_bs", c_long), ("mma", c_int), ("status", c_void_p)]


P = c_void_p
_SIGS = {
    "motif_abi_version": (c_int, []),
    "motif_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "motif_set_option": (c_int, [c_char_p, c_int]),
    "motif_get_option": (c_int, [c_char_p, POINTER(c_int)]),
    "motif_splat_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "motif_splat_motif_fwd": (c_int, [P, P, P, P, P, P, c_float, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_splat_motif_acc_fwd": (c_int, [P, P, P, P, P, P, c_float, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_splat_motif_pre_fwd": (c_int, [P, P, P, P, P, P, P, c_float, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_siren_synth_pre_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "motif_reliability_pairs_fwd": (c_int, [P, c_long, c_long, P, P, POINTER(c_int), POINTER(c_float), c_int, c_int, P, P, c_int, c_int, c_int, P]),
    "motif_siren_pack": (c_long, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), c_int, P, P]),
    "motif_frames_u8_to_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "motif_frames_f32_to_u8": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_siren_pack_split": (c_long, [c_int, POINTER(c_void_p), POINTER(c_void_p), P, P]),
    "motif_siren_imnet_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_siren_imnet_add_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_siren_flow_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_siren_synth_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "motif_synth_input_fwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_conv2d_packed_size": (c_long, [POINTER(MotifConvDesc)]),
    "motif_conv2d_pack": (c_int, [POINTER(MotifConvDesc), P, P, P]),
    "motif_conv2d_fwd": (c_int, [POINTER(MotifConvDesc), P, P, P, P, P, P, P]),
    "motif_conv2d_fwd_multi": (c_int, [POINTER(MotifConvDesc), c_int] + [POINTER(c_void_p)] * 6 + [POINTER(c_long)] * 4 + [P]),
    "motif_conv2d_chain_ws_words": (c_long, [POINTER(MotifConvDesc), c_int]),
    "motif_conv2d_chain_fwd": (c_int, [POINTER(MotifConvDesc), c_int, P, P, P, P, c_long, P, P]),
    "motif_dcn_v2_fwd_multi": (c_int, [c_int, POINTER(c_void_p), POINTER(c_long)] + [POINTER(c_void_p)] * 4 + [P, POINTER(c_void_p)]
                               + [c_int] * 11 + [c_long, c_long, c_int, P]),
    "motif_dcn_v2_fwd": (c_int, [P, P, P, P, P, P, P] + [c_int] * 11 + [c_long, c_long, c_int, P]),
    "motif_dcn_v2_fused_fwd_multi": (c_int, [c_int, POINTER(c_void_p), POINTER(c_long)] + [POINTER(c_void_p)] * 5
                                     + [c_int] * 6 + [c_long, c_long, c_int, c_int, P, P]),
    "motif_dcn_split_pack": (c_long, [P, P, c_int, c_int, P]),
    "motif_raft_corr_lookup": (c_int, [P, P, P, c_float, P] + [c_int] * 7 + [c_int, c_int, c_float, P]),
    "motif_raft_corr_lookup_pyramid": (c_int, [P, POINTER(c_void_p), POINTER(c_int), POINTER(c_int), c_int, P, P] + [c_int] * 6 + [c_float, POINTER(c_int), POINTER(c_int), P]),
    "motif_corr81_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_resize_bilinear": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "motif_backwarp": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "motif_pwc_backward_warp": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "motif_reliability_fwd": (c_int, [P, P, c_long, P, P, P, P, c_int, c_int, c_int, P]),
    "motif_instance_norm": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "motif_instance_norm_ws": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "motif_instance_norm_affine_ws": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "motif_instance_norm_moments": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "motif_instance_norm_apply": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "motif_avg_pool2": (c_int, [P, P, c_int, c_int, c_int, P]),
    "motif_nchw_to_nhwc": (c_int, [P, P, c_int, c_int, c_int, P]),
    "motif_gru_update": (c_int, [P, P, P, P, c_long, P]),
    "motif_lstm_gates": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "motif_axpby": (c_int, [P, P, c_float, c_float, P, c_long, P]),
    "motif_axpby_bs": (c_int, [P, P, c_float, c_float, P, c_int, c_long, c_long, P]),
    "motif_flow_roundtrip": (c_int, [P, P, c_int, c_long, c_float, c_float, P]),
    "motif_deconv4x4s2": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
}
ABI_VERSION = 9          # include/motif_hip.h / api.hip: motif_abi_version()
EXPORTS = tuple(_SIGS)


def load(build=False):
    """Return the loaded library; raises RuntimeError if it is absent (no CPU/eager fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if build or (not os.path.exists(SO_PATH) and not os.environ.get("MOTIF_HIP_LIB")):
        try:
            from .csrc import build as _b
            _b.build()
        except Exception as e:                # no hipcc / compile error: still no fallback, say why
            raise RuntimeError("libmotif_hip.so is missing at %s and building it failed (%s: %s) -- needs hipcc "
                               "(--offload-arch=gfx950); there is no fallback path" % (SO_PATH, type(e).__name__, e)) from e
    if not os.path.exists(SO_PATH):
        raise RuntimeError("libmotif_hip.so not found at %s; there is no fallback path" % SO_PATH)
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch with include/motif_hip.h
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d (%s)" % (what, rc, "argument error" if rc < 0 else "hipError_t"))
