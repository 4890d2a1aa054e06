// Shared device helpers for libmotif_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/motif_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
