// Shared by the two bf16-matrix-core convolution kernels (conv_split.hip, conv_pp.hip): the 3-way bf16 split of an fp32
// value and the list of (weight part, activation part) products.  See conv_split.hip for the arithmetic.
#pragma once
#include "conv_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
// (weight part, activation part) of each product, smallest terms first
template <int NP> struct SplitProducts;
template <> struct SplitProducts<1> { static constexpr int n = 1; static constexpr int w[1] = {0}; static constexpr int x[1] = {0}; };
template <> struct SplitProducts<2> { static constexpr int n = 3; static constexpr int w[3] = {1, 0, 0}; static constexpr int x[3] = {0, 1, 0}; };
template <> struct SplitProducts<3> {
    static constexpr int n = 6;
    static constexpr int w[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int x[6] = {0, 2, 1, 0, 1, 0};
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 floats -> NP packed-bf16 quads (part 0 = leading bits)
template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[NP]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x0 = v[2 * q], x1 = v[2 * q + 1];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned pk = pk_bf16(x0, x1);
            out[p][q] = pk;
            if (p + 1 < NP) { x0 -= bf_lo(pk); x1 -= bf_hi(pk); }
        }
    }
}
}  // namespace

// mma = 7 (two fp16 parts, conv_wino.hip only): every other kernel runs such a layer in the three-part bf16 form
static inline int split_parts(int mma) { return (mma == 6 || mma == 7) ? 3 : mma == 3 ? 2 : mma == 1 ? 1 : 0; }

// round-3 kernel (conv_split2.hip): same packed weights, same ConvArgs as conv_split.hip
bool motif_conv_split2_eligible(const MotifConvDesc* d, const ConvArgs& a, int P);
int motif_conv_split2_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s);
// round-4 kernel (conv_wino.hip): Winograd F(2,3) along the rows, its own fragment block behind the direct one in the packed blob
long motif_conv_split_packed_floats_direct(const MotifConvDesc* d);
long motif_conv_wino_packed_floats(const MotifConvDesc* d);
int motif_conv_wino_pack(const MotifConvDesc* d, const float* weight, float* packed, hipStream_t s);
bool motif_conv_wino_eligible(const MotifConvDesc* d, const ConvArgs& a, int P);
int motif_conv_wino_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s);
