// Shared device helpers for libmotif_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/motif_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MOTIF_SCALAR_F32: a kernel compiled WITHOUT the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32).
// Round 6 (met in a fused RAFT bottleneck kernel that was dropped again -- DESIGN_LOG.md; tools/pk_fma_beside_mfma.hip reproduces it
// without the library; profiles/r06_pk_fma_beside_mfma.txt):
//     v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]       with b a VECTOR register pair (the LOW result takes b's HIGH register)
// came out wrong in its low half, in lanes 48..63 of the wave only, while an fp16 / bf16 MFMA kernel ran beside it on another stream -- never
// alone, never beside fp32 MFMAs.  The form is necessary for the failure, not sufficient (other code layouts of the same instruction ran
// clean; what the failing ones share is not known), so it is fenced by form and widely: hipcc picks it by itself (SLP-paired scalar FMAs
// against a broadcast operand), a kernel in which it picked ANY op_sel bit on a vector-register source of a packed fp32 instruction is
// compiled without packed fp32 altogether, and tests/test_isa_hygiene.py fails on any kernel of the BUILT library that holds one.  (op_sel on a
// SCALAR register pair -- conv_direct.hip's weights -- is a per-instruction selection and stays.)  The attribute means nothing to the host pass.
#if !defined(MOTIF_SCALAR_F32) && defined(__HIP_DEVICE_COMPILE__)
#define MOTIF_SCALAR_F32 __attribute__((target("no-packed-fp32-ops")))
#elif !defined(MOTIF_SCALAR_F32)
#define MOTIF_SCALAR_F32
#endif

#define MOTIF_LAUNCH_CHECK()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case MOTIF_ACT_RELU: return v > 0.f ? v : 0.f;
        case MOTIF_ACT_LRELU: return v > 0.f ? v : 0.1f * v;
        case MOTIF_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case MOTIF_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Tuning / test switches (include/motif_hip.h: motif_set_option).  Read from the environment once, never per launch.
enum MotifOpt { MOTIF_OPT_CONV_DBG, MOTIF_OPT_CONV_CK, MOTIF_OPT_CONV_NOSPEC, MOTIF_OPT_CONV_ENGINE, MOTIF_OPT_LDS_PAD,
                MOTIF_OPT_CORR81, MOTIF_OPT_DCN_NOWIN, MOTIF_OPT_DCN_WAVES, MOTIF_OPT_DCN_FRONT_PAD, MOTIF_OPT_DCN_BACK_PAD,
                MOTIF_OPT_SIREN_STAGGER, MOTIF_OPT_CONV_NOVEC, MOTIF_OPT_CONV_NODIRECT, MOTIF_OPT_CONV_WINO_TR, MOTIF_OPT_CONV_WINO_RPRE, MOTIF_OPT_CONV_CHAIN_WGS, MOTIF_OPT_RESIZE_NARROW, MOTIF_OPT_CONV_DIRECT_QUADS, MOTIF_OPT_COUNT };
int motif_opt(int id);                                   // api.hip

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in linear-id order;
// this maps blockIdx.x so that the blocks landing on one XCD get a CONTIGUOUS run of tile ids: vertically adjacent
// tiles then share their halo rows in that XCD's L2 (measured on the recon-trunk launch: FETCH_SIZE 73 -> 27 MB).
// A bijection of [0, gridDim.x) for every grid size (tests: tools/ + CPU check in the commit that introduced it).
__device__ __forceinline__ int xcd_tile_id() {
    const int gx = gridDim.x;
    const int off = (int)(((long)(blockIdx.z * gridDim.y + blockIdx.y) * gx) & 7);
    const int xcd = (blockIdx.x + off) & 7, first = (xcd - off) & 7;
    int base = 0;
    for (int c = 0; c < xcd; ++c) { const int f = (c - off) & 7; base += f < gx ? (gx - f + 7) >> 3 : 0; }
    return base + ((blockIdx.x - first) >> 3);
}
