// The two operand formats of the dense GEMMs on the 16-bit matrix pipe (round 5), and what an operand's producer does with them.
//
//   LPM_OPERAND_BF16X3 (round 1-4, every model): x = xh + xl in bf16 planes (2^-17 residual), THREE products per a . b
//       (ah bh + al bh + ah bl) -- ~5e-6 per GEMM.  Activation images [rows][3K] = [hi | lo | hi], gradient images [hi | hi | lo],
//       weight images with the matching plane orders (split_gemm.hip), fragment tiles with (hi, lo) planes (tile_gemm.h).
//   fp16 planes (round 5, NetVladV1's encoder GEMMs; transformer_utils.py:559-561,583,701-711 and TF autodiff of those layers):
//     LPM_OPERAND_FP16X3, ACTIVATIONS: x = xh + xl in FP16 planes (11 + 11 bits), image rows [hi | lo | hi] like the bf16 form; the
//       FORWARD product runs all three terms against a weight that is split as well ([Wh ; Wh ; Wl]): ~1e-6 per GEMM -- the forward
//       is as exact as before (a forward error would flip ReLU masks: a gradient error of sqrt(forward error)).
//     LPM_OPERAND_FP16X2, GRADIENTS: dy = dyh + dyl in fp16 planes, image rows [hi | lo].  The input gradient is a two-term product,
//       dx = [dyh | dyl] . [Wh ; Wh], against the weight rounded ONCE to fp16 (2.1e-4 relative L2 per GEMM against fp64: the 2^-12
//       rounding of the one-plane operand).  The weight gradient is, by default, the ONE-term product dW = xh^T dyh of the two images'
//       hi planes read in place (ops.DW_TERMS = 1: both operands rounded once, 2.9e-4 per GEMM -- an error that stays in that weight's
//       gradient); LPM_DW_TERMS=2 runs dW = xh^T [dyh | dyl] (the gradient exact, 1.4e-4, twice the matrix-pipe work).  Forward 3 +
//       input gradient 2 + weight gradient 1 = 6 products per GEMM triple against split-bf16's 9.
//     fp16 has five exponent bits: the producer multiplies every value by a power of two (`scale`, chosen by the host from the tensor's
//       max |.| of an EARLIER step -- ops.OperandScales, delayed scaling) and the GEMM's consumer multiplies by 1 / scale (exact).
//       Values beyond the format's range saturate at +-65504 instead of becoming infinite.  The matrix cores keep fp16 subnormals
//       (measured, tools/fp16_probe.py), so below 2^-14 the pair (hi, lo) degrades into fixed point with a 2^-25 quantum: with
//       max |x| scale in [2^10, 2^11) every value down to 2^-24 of the tensor's maximum keeps 11 bits.
// In both formats the producer can record max |x| (before scaling) of what it wrote: one atomic max per wave into *amax.
#pragma once
#include "lpm_common.h"

namespace lpm {

struct OperandFmt {
    int f16;          // 0: bf16 planes;  1: fp16 planes
    int planes;       // 16-bit planes per image row: 3 ([hi | lo | hi] activations, [hi | hi | lo] bf16 gradients) or 2 ([hi | lo])
    float scale;      // multiplied in before the split (power of two; 1 for bf16)
    float* amax;      // device, nullable
};
inline int operand_kind_f16(int kind) { return kind == LPM_OPERAND_BF16X3 ? 0 : 1; }
inline int operand_kind_planes(int kind) { return kind == LPM_OPERAND_FP16X2 ? 2 : 3; }
inline bool operand_kind_ok(int kind) { return kind == LPM_OPERAND_BF16X3 || kind == LPM_OPERAND_FP16X2 || kind == LPM_OPERAND_FP16X3; }
inline OperandFmt operand_fmt(const LpmOperandFormat* f) {
    OperandFmt o{0, 3, 1.f, nullptr};
    if (f) {
        o.f16 = operand_kind_f16(f->kind);
        o.planes = operand_kind_planes(f->kind);
        o.scale = (o.f16 && f->scale > 0.f) ? f->scale : 1.f;
        o.amax = f->amax;
    }
    return o;
}
inline int operand_fmt_check(const LpmOperandFormat* f, const char* what) {
    if (!f) return LPM_OK;
    if (!operand_kind_ok(f->kind)) { set_error("%s: unknown operand format %d", what, f->kind); return LPM_ERR_BADARG; }
    if (!(f->scale > 0.f) || f->scale != f->scale) { set_error("%s: the operand scale must be a positive power of two", what); return LPM_ERR_BADARG; }
    if (((uintptr_t)f->amax & 3) != 0) { set_error("%s: misaligned amax", what); return LPM_ERR_BADARG; }
    return LPM_OK;
}

typedef _Float16 of_h2 __attribute__((ext_vector_type(2)));
typedef __bf16 of_b2 __attribute__((ext_vector_type(2)));
typedef float of_f2 __attribute__((ext_vector_type(2)));
constexpr float kF16Max = 65504.f;

// two values -> packed (hi, lo) words.  bf16: round-to-nearest-even both (v_cvt_pk_bf16_f32), bit for bit tile_gemm.h's tg_split2.
// fp16: v * scale clamped to the format's range, hi = rne(v), lo = rne(v - hi).
__device__ __forceinline__ void of_split2(float a, float b, int f16, float scale, unsigned& hi, unsigned& lo) {
    if (f16) {
        const of_f2 v = {__builtin_amdgcn_fmed3f(a * scale, -kF16Max, kF16Max), __builtin_amdgcn_fmed3f(b * scale, -kF16Max, kF16Max)};
        const of_h2 h = __builtin_convertvector(v, of_h2);
        const of_f2 hf = __builtin_convertvector(h, of_f2);
        const of_h2 l = __builtin_convertvector(v - hf, of_h2);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        const of_f2 v = {a, b};
        const of_b2 h = __builtin_convertvector(v, of_b2);
        const of_f2 hf = __builtin_convertvector(h, of_f2);
        const of_b2 l = __builtin_convertvector(v - hf, of_b2);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    }
}
__device__ __forceinline__ void of_split8(const float* v, int f16, float scale, uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) of_split2(v[2 * i], v[2 * i + 1], f16, scale, h[i], l[i]);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void of_split4(float a, float b, float c, float d, int f16, float scale, uint2& hi, uint2& lo) {
    unsigned h0, l0, h1, l1;
    of_split2(a, b, f16, scale, h0, l0);
    of_split2(c, d, f16, scale, h1, l1);
    hi = make_uint2(h0, h1);
    lo = make_uint2(l0, l1);
}
// fp16 (hi, lo) of a weight (no scale; |w| < 65504)
__device__ __forceinline__ void of_split1_f16(float v, unsigned& h, unsigned& l) {
    const _Float16 hh = (_Float16)__builtin_amdgcn_fmed3f(v, -kF16Max, kF16Max);
    const _Float16 ll = (_Float16)(v - (float)hh);
    h = (unsigned)__builtin_bit_cast(unsigned short, hh);
    l = (unsigned)__builtin_bit_cast(unsigned short, ll);
}
// hi-plane-only (the one-plane operand of a two-term product: rounded once)
__device__ __forceinline__ unsigned of_round2_f16(float a, float b) {
    const of_f2 v = {__builtin_amdgcn_fmed3f(a, -kF16Max, kF16Max), __builtin_amdgcn_fmed3f(b, -kF16Max, kF16Max)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, of_h2));
}
__device__ __forceinline__ float of_f16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float of_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// the value an image holds at (hi, lo), un-scaled
__device__ __forceinline__ float of_value(unsigned short h, unsigned short l, int f16, float inv_scale) {
    return f16 ? (of_f16_to_f32(h) + of_f16_to_f32(l)) * inv_scale : of_bf16_to_f32(h) + of_bf16_to_f32(l);
}

// ---- image rows.  An image row holds K values as `planes` planes of K 16-bit words: three -- activation order [hi | lo | hi], gradient
// order (`grad`) [hi | hi | lo] -- or two, [hi | lo].
__device__ __forceinline__ int64_t of_row_stride(int K, int planes) { return (int64_t)planes * K; }
__device__ __forceinline__ void of_store_row8(unsigned short* row, int K, int c, const uint4& hi, const uint4& lo, int planes, int grad) {
    *reinterpret_cast<uint4*>(row + c) = hi;
    if (planes == 2) {
        *reinterpret_cast<uint4*>(row + K + c) = lo;
    } else {
        *reinterpret_cast<uint4*>(row + K + c) = grad ? hi : lo;
        *reinterpret_cast<uint4*>(row + 2 * (int64_t)K + c) = grad ? lo : hi;
    }
}
__device__ __forceinline__ void of_store_row4(unsigned short* row, int K, int c, const uint2& hi, const uint2& lo, int planes, int grad) {
    *reinterpret_cast<uint2*>(row + c) = hi;
    if (planes == 2) {
        *reinterpret_cast<uint2*>(row + K + c) = lo;
    } else {
        *reinterpret_cast<uint2*>(row + K + c) = grad ? hi : lo;
        *reinterpret_cast<uint2*>(row + 2 * (int64_t)K + c) = grad ? lo : hi;
    }
}
// "the forward activation was > 0" from the hi plane of its image (either format: sign bit 15, zero = all other bits clear)
__device__ __forceinline__ bool of_positive(unsigned h16) { return (h16 & 0x7fffu) != 0u && !(h16 & 0x8000u); }

// ---- max |x|: a thread keeps a running maximum, the wave joins and one lane issues the atomic (non-negative floats order like their bits).
// Atomics on ONE address serialise in L2 at ~6 ns each: 65 536 waves of a split pass cost 400 us that way (measured, round 5: every
// kernel with an unconditional commit grew by 250-450 us).  The slot only ever grows within a step, so a wave first LOOKS (an
// agent-scope load) and stays silent unless it would raise the value.  That still leaves the first residency of a launch -- 2 000 to
// 8 000 waves that all look before anyone has written (+25-45 us, measured) -- so a site's slot is OF_AMAX_SUB sub-slots, one cache line
// apart, chosen by workgroup: the host takes the maximum over them (ops.OperandScales).
// (as an UNSIGNED maximum of the magnitudes' bit patterns: a NaN input orders above Inf and survives into the slot -- v_max_f32 would drop it,
// and the fp16 split turns a NaN into a finite -65504 (v_med3_f32): the host must hear about it.  ops.OperandScales.begin_step raises on a
// harvested maximum that is NaN or Inf -- ADVICE r5)
__device__ __forceinline__ float of_amax8(float m, const float* v) {
    unsigned u = __float_as_uint(m);
#pragma unroll
    for (int e = 0; e < 8; ++e) u = max(u, __float_as_uint(v[e]) & 0x7fffffffu);
    return __uint_as_float(u);
}
__device__ __forceinline__ float of_amax4(float m, float a, float b, float c, float d) {
    const unsigned u = max(max(__float_as_uint(m), __float_as_uint(a) & 0x7fffffffu),
                           max(max(__float_as_uint(b) & 0x7fffffffu, __float_as_uint(c) & 0x7fffffffu), __float_as_uint(d) & 0x7fffffffu));
    return __uint_as_float(u);
}
constexpr int OF_AMAX_SUB = 32;          // sub-slots per site (== LPM_OPERAND_AMAX_SUB)
constexpr int OF_AMAX_STRIDE = 16;       // floats between sub-slots: 64 bytes (== LPM_OPERAND_AMAX_STRIDE)
__device__ __forceinline__ void of_amax_commit(float* amax, float m) {
    if (!amax) return;
    unsigned mu = __float_as_uint(m) & 0x7fffffffu;          // the wave's maximum as bits, so that a NaN is not lost on the way
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mu = max(mu, (unsigned)__shfl_xor((int)mu, o, 64));
    m = __uint_as_float(mu);
    if ((threadIdx.x & 63) == 0 && mu != 0u) {
        unsigned* slot = reinterpret_cast<unsigned*>(amax) + OF_AMAX_STRIDE * ((blockIdx.x + 7 * blockIdx.y + (threadIdx.x >> 6)) & (OF_AMAX_SUB - 1));
        const unsigned cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__float_as_uint(m) > cur) atomicMax(slot, __float_as_uint(m));
    }
}

}  // namespace lpm
