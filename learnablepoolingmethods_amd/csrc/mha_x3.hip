// K4 on the bf16 matrix pipe -- the attention core of mha.hip with split-bf16 operands (v = hi + lo bf16 planes,
// a*b = ah*bh + ah*bl + al*bh accumulated in fp32: ~1e-5 relative error on the logits, inside the 1e-3 parity bar) and
// v_mfma_f32_16x16x32_bf16.  Same structure as the fp32 form: one 256-thread workgroup per (batch, head), TRANSPOSED score
// tiles S^T[key, q] so that a lane owns one query column and the probability tile is already the B operand of the PV
// product -- no LDS round trip for P.  What changes:
//   * K is staged once per workgroup as bf16 hi/lo planes in fragment order (16 bytes per key per 8-column half): one
//     conflict-free ds_read_b128 per key tile.  The 32-deep reduction of the 16x16x32 MFMA holds [Kh | Kl] against
//     [Qh | Qh], so a score tile costs 2 MFMAs (1 for d = 8) instead of 4 fp32 ones at half the rate each.
//   * V is staged TRANSPOSED (Vt[d][key], hi and lo planes) with the keys of every 32-key block permuted to the order in
//     which the score-tile accumulators hold them (lane group g owns keys 4g..4g+3 of both 16-key tiles): the A operand
//     of O^T += V^T P^T is one ds_read_b128 per plane, the B operand is the lane's own 8 probabilities.
//   * fp32 -> (hi, lo) splits use v_cvt_pk_bf16_f32 (2.5 VALU ops per element).
// transformer_utils.py:564-581 (MultiHeadAttention) and :652-659 (logits_bn variant: key_scale / key_shift).
#include "lpm_common.h"
#include "operand_format.h"
#include <atomic>
#include <cstdlib>

namespace lpm {

constexpr float MX_LOG2E = 1.4426950408889634f;
constexpr float MX_LN2 = 0.6931471805599453f;
// Probabilities are formed with v_exp_f32 (2^x) directly: the log2(e) factor rides in the query scale (or, for the
// logits_bn variant, is applied to z once), saving a multiply per score.
typedef __bf16 mx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mx_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned mx_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mx_mfma(mx_u32x4 a, mx_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mx_bf16x8, a), __builtin_bit_cast(mx_bf16x8, b), c, 0, 0, 0);
}
// (a, b) -> packed bf16 hi pair and lo pair (RNE both)
__device__ __forceinline__ void mx_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const mx_f32x2 v = {a, b};
    const mx_bf16x2 h = __builtin_convertvector(v, mx_bf16x2);
    const mx_f32x2 hf = __builtin_convertvector(h, mx_f32x2);
    const mx_bf16x2 l = __builtin_convertvector(v - hf, mx_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void mx_split8(const float* v, mx_u32x4& hi, mx_u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        mx_split2(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h;
        lo[i] = l;
    }
}
// Gradient-image output (img = plane stride in bf16 elements, 0 = plain fp32 output): the 4 values go out as bf16 hi / hi /
// lo planes [hi | hi | lo] at p, p + img, p + 2 img -- the operand image the q/k/v weight- and input-gradient GEMMs read
// (ops._split_rows(grad=True)), so the fp32 gradient and the split pass over it never exist.
// Round 5: both images also exist in the fp16 two-product format (operand_format.h) -- [hi | lo] planes of value * scale, max |value|
// recorded for the host's delayed scale.  MxImg carries the formats of the attention result's image (o*) and of the gradient image (g*).
struct MxImg {
    int of16, oplanes; float oscale, oinv; float* oamax;
    int gf16, gplanes; float gscale; float* gamax;
    // logits_bn's one-pass backward: the key / value sweep's dk still lacks the batch statistics' share (lpm_mha_bn_dk_correct), so it
    // leaves as plain fp32 rows dk_ld floats apart while dv goes into the gradient image; the repair writes dk's part of the image
    int dk_plain; long long dk_ld;
};

__device__ __forceinline__ void mx_store_grad4(float* base, int64_t off, int img, float a, float b, float c, float d, const MxImg& im, float& vmax) {
    if (img == 0) {
        *reinterpret_cast<float4*>(base + off) = make_float4(a, b, c, d);
    } else if (im.gf16) {
        vmax = of_amax4(vmax, a, b, c, d);
        uint2 hi, lo;
        of_split4(a, b, c, d, 1, im.gscale, hi, lo);
        unsigned short* p = reinterpret_cast<unsigned short*>(base) + off;
        *reinterpret_cast<uint2*>(p) = hi;
        if (im.gplanes == 2) {
            *reinterpret_cast<uint2*>(p + img) = lo;
        } else {
            *reinterpret_cast<uint2*>(p + img) = hi;
            *reinterpret_cast<uint2*>(p + 2 * (int64_t)img) = lo;
        }
    } else {
        vmax = of_amax4(vmax, a, b, c, d);
        unsigned h0, l0, h1, l1;
        mx_split2(a, b, h0, l0);
        mx_split2(c, d, h1, l1);
        unsigned short* p = reinterpret_cast<unsigned short*>(base) + off;
        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + img) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + 2 * (int64_t)img) = make_uint2(l0, l1);
    }
}
// the key gradient of one lane: four consecutive columns (from head column c) of global key row `row`
__device__ __forceinline__ void mx_store_dk4(float* dk, int64_t row, int64_t ldd, int c, int img, float a, float b, float cc, float d, const MxImg& im, float& vmax) {
    if (im.dk_plain) *reinterpret_cast<float4*>(dk + row * im.dk_ld + c) = make_float4(a, b, cc, d);
    else mx_store_grad4(dk, row * ldd + c, img, a, b, cc, d, im, vmax);
}
// Activation-image output / input of the attention result (oimg = plane stride in bf16 elements = h*d, 0 = plain fp32): row =
// [hi | lo | hi] planes, the operand image of the output projection GEMM (ops._split_rows): the forward writes it instead of an
// fp32 o and the backward kernels rebuild o = hi + lo (2^-17) for D_q = <dO_q, O_q>.
__device__ __forceinline__ void mx_store_act4(float* base, int64_t row, int64_t ldo, int col, int oimg, float a, float b, float c, float d,
                                              const MxImg& im, float& vmax) {
    if (oimg == 0) {
        *reinterpret_cast<float4*>(base + row * ldo + col) = make_float4(a, b, c, d);
    } else if (im.of16) {
        vmax = of_amax4(vmax, a, b, c, d);
        uint2 hi, lo;
        of_split4(a, b, c, d, 1, im.oscale, hi, lo);
        unsigned short* p = reinterpret_cast<unsigned short*>(base) + row * im.oplanes * (int64_t)oimg + col;
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + oimg) = lo;
        if (im.oplanes == 3) *reinterpret_cast<uint2*>(p + 2 * (int64_t)oimg) = hi;
    } else {
        vmax = of_amax4(vmax, a, b, c, d);
        unsigned h0, l0, h1, l1;
        mx_split2(a, b, h0, l0);
        mx_split2(c, d, h1, l1);
        unsigned short* p = reinterpret_cast<unsigned short*>(base) + row * 3 * (int64_t)oimg + col;
        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + oimg) = make_uint2(l0, l1);
        *reinterpret_cast<uint2*>(p + 2 * (int64_t)oimg) = make_uint2(h0, h1);
    }
}
__device__ __forceinline__ float mx_h2f(unsigned w, int hi16) { return of_f16_to_f32((unsigned short)(hi16 ? (w >> 16) : (w & 0xffffu))); }
__device__ __forceinline__ void mx_load_o8(const float* base, int64_t row, int64_t ldo, int col, int oimg, float4& a, float4& c, const MxImg& im) {
    if (oimg == 0) {
        a = *reinterpret_cast<const float4*>(base + row * ldo + col);
        c = *reinterpret_cast<const float4*>(base + row * ldo + col + 4);
    } else if (im.of16) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + row * im.oplanes * (int64_t)oimg + col;
        const uint4 h = *reinterpret_cast<const uint4*>(p), l = *reinterpret_cast<const uint4*>(p + oimg);
        const float s = im.oinv;
        a = make_float4((mx_h2f(h.x, 0) + mx_h2f(l.x, 0)) * s, (mx_h2f(h.x, 1) + mx_h2f(l.x, 1)) * s, (mx_h2f(h.y, 0) + mx_h2f(l.y, 0)) * s,
                        (mx_h2f(h.y, 1) + mx_h2f(l.y, 1)) * s);
        c = make_float4((mx_h2f(h.z, 0) + mx_h2f(l.z, 0)) * s, (mx_h2f(h.z, 1) + mx_h2f(l.z, 1)) * s, (mx_h2f(h.w, 0) + mx_h2f(l.w, 0)) * s,
                        (mx_h2f(h.w, 1) + mx_h2f(l.w, 1)) * s);
    } else {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + row * 3 * (int64_t)oimg + col;
        const uint4 h = *reinterpret_cast<const uint4*>(p), l = *reinterpret_cast<const uint4*>(p + oimg);
        a = make_float4(__uint_as_float(h.x << 16) + __uint_as_float(l.x << 16), __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u),
                        __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16), __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u));
        c = make_float4(__uint_as_float(h.z << 16) + __uint_as_float(l.z << 16), __uint_as_float(h.z & 0xffff0000u) + __uint_as_float(l.z & 0xffff0000u),
                        __uint_as_float(h.w << 16) + __uint_as_float(l.w << 16), __uint_as_float(h.w & 0xffff0000u) + __uint_as_float(l.w & 0xffff0000u));
    }
}
// ---- round 6: the backward's last products on TWO terms, fp16 planes (the treatment the dense GEMMs' backward got in round 5) --------
// What stays exact (split-bf16 x3): the scores S (they sit under an exp) AND dP = dO V^T -- it meets D_q = <dO_q, O_q> in the difference
// dS = P (dP - D_q), and with a peaked attention that difference is a small remainder of two large numbers: a first version with V
// rounded once (dP as ONE fp16 MFMA) was 8e-4 off in the input gradient of tests/test_gpu_attention_modules.py (kernels x 4, tolerance
// 2e-4) where the three-term form is at 1e-5.  What goes to two fp16 terms are the three products BEHIND dS: dV = dO^T P, dK = Q^T dS,
// dQ = K^T dS.  The operand that is re-read from LDS stays exact as fp16 (hi, lo) -- dO^T * s and Q^T in the dkv kernel, K^T in the dq
// kernel -- and the per-score operand formed in registers is rounded ONCE to fp16 (2^-12 relative, the arithmetic of the dense layers'
// input gradients): P, dS.  Two MFMAs per product instead of three and ONE v_cvt_pk_f16_f32 per pair of scores instead of the (hi, lo)
// split's five VALU operations.  fp16 has five exponent bits and gradients span 1e-9 ... 20: dO is scaled by a power of two s taken
// from max |dO| -- per query in the dq kernel, per (batch, head) in the dkv kernel (its reductions run over queries) -- BEFORE it is
// split for dP, so dP, D_q and dS carry s for free and the stores take it out again (exact both ways).
// LPM_MHA_BWD_TERMS=3 / lpm_mha_bwd_set_terms(3) keeps the three-term bf16 form (A/B).
typedef _Float16 mx_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mx_mfma_h(mx_u32x4 a, mx_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_f16x8, a), __builtin_bit_cast(mx_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ mx_u32x4 mx_round8_h(const float* v) {
    return mx_u32x4{of_round2_f16(v[0], v[1]), of_round2_f16(v[2], v[3]), of_round2_f16(v[4], v[5]), of_round2_f16(v[6], v[7])};
}
__device__ __forceinline__ void mx_split8_h(const float* v, float s, mx_u32x4& hi, mx_u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        of_split2(v[2 * i], v[2 * i + 1], 1, s, h, l);
        hi[i] = h;
        lo[i] = l;
    }
}
// the power of two s with amax * s in [2^4, 2^5) and its inverse (amax == 0 or denormal: 1).  With max |dO| s < 32 a score gradient
// dS = P (dP - D) stays inside fp16's range while sum_d |V_d| < 1024; beyond, of_round2_f16 saturates.
__device__ __forceinline__ void mx_pow2_scale(float amax, float& s, float& inv) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    const int se = e == 0 ? 127 : min(max(258 - e, 1), 253);
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}
// stage rows [0, L) of head hh transposed + permuted as fp16 (hi, lo) planes: T[plane][d][perm(row)]
template <int D>
__device__ __forceinline__ void mx_stage_transposed_h(unsigned char* dst, const float* __restrict__ src, int64_t ld, int b, int L, int LP,
                                                      int hh, int tid, int nt) {
    constexpr int NH = D / 8;
    const int TS = LP * 2 + 16;
    for (int i = tid; i < L * NH; i += nt) {
        const int row = i / NH, hf = i % NH;
        const float* p = src + ((int64_t)b * L + row) * ld + hh * D + 8 * hf;
        const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        mx_u32x4 hi, lo;
        mx_split8_h(v, 1.f, hi, lo);
        const int pos = (row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int dd = 8 * hf + e;
            *reinterpret_cast<unsigned short*>(dst + dd * TS + pos * 2) = (unsigned short)(hi[e >> 1] >> ((e & 1) * 16));
            *reinterpret_cast<unsigned short*>(dst + 16 * TS + dd * TS + pos * 2) = (unsigned short)(lo[e >> 1] >> ((e & 1) * 16));
        }
    }
}

// position of key (or query) `i` inside its 32-block in the permuted order: i = 16 t + 4 g + e  ->  8 g + 4 t + e
__device__ __forceinline__ int mx_perm(int i) { return (i & ~31) | (((i >> 2) & 3) << 3) | (((i >> 4) & 1) << 2) | (i & 3); }

// LDS geometry for a sequence padded to LP (a multiple of 32) positions
//   row planes of an [L, D] operand: (D/8)*2 arrays of LP x 16 bytes, array index = plane * (D/8) + half
//   transposed planes:               2 planes x 16 rows x (LP*2 + 16) bytes
__host__ __device__ constexpr int mx_rowplanes_bytes(int LP, int D) { return (D / 8) * 2 * LP * 16; }
__host__ __device__ constexpr int mx_tstride(int LP) { return LP * 2 + 16; }
__host__ __device__ constexpr int mx_tplanes_bytes(int LP) { return 2 * 16 * mx_tstride(LP); }

// stage rows [0, L) of head hh of a [B, L, ld] tensor as row planes (optionally scaled)
template <int D>
__device__ __forceinline__ void mx_stage_rows(unsigned char* dst, const float* __restrict__ src, int64_t ld, int b, int L, int LP,
                                              int hh, float mul, int tid, int nt = 256) {
    constexpr int NH = D / 8;
    for (int i = tid; i < L * NH; i += nt) {
        const int row = i / NH, hf = i % NH;
        const float* p = src + ((int64_t)b * L + row) * ld + hh * D + 8 * hf;
        const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
        const float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, c.x * mul, c.y * mul, c.z * mul, c.w * mul};
        mx_u32x4 hi, lo;
        mx_split8(v, hi, lo);
        *reinterpret_cast<mx_u32x4*>(dst + ((0 * NH + hf) * LP + row) * 16) = hi;
        *reinterpret_cast<mx_u32x4*>(dst + ((1 * NH + hf) * LP + row) * 16) = lo;
    }
}
// stage the same rows transposed and permuted: T[plane][d][perm(row)]
template <int D>
__device__ __forceinline__ void mx_stage_transposed(unsigned char* dst, const float* __restrict__ src, int64_t ld, int b, int L,
                                                    int LP, int hh, int tid, int nt = 256) {
    constexpr int NH = D / 8;
    const int TS = mx_tstride(LP);
    for (int i = tid; i < L * NH; i += nt) {
        const int row = i / NH, hf = i % NH;
        const float* p = src + ((int64_t)b * L + row) * ld + hh * D + 8 * hf;
        const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        mx_u32x4 hi, lo;
        mx_split8(v, hi, lo);
        const int pos = mx_perm(row);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int dd = 8 * hf + e;
            const unsigned short h16 = (unsigned short)(hi[e >> 1] >> ((e & 1) * 16)), l16 = (unsigned short)(lo[e >> 1] >> ((e & 1) * 16));
            *reinterpret_cast<unsigned short*>(dst + dd * TS + pos * 2) = h16;
            *reinterpret_cast<unsigned short*>(dst + 16 * TS + dd * TS + pos * 2) = l16;
        }
    }
}

// A fragment of a row-plane operand for tile `t` (16 rows): lane group g picks array g (D = 16: Kh0, Kh1, Kl0, Kl1;
// D = 8: Kh0, Kl0, Kh0, Kh0) -- one ds_read_b128.
template <int D>
__device__ __forceinline__ mx_u32x4 mx_row_frag(const unsigned char* planes, int LP, int t, int l15, int g) {
    const int arr = (D == 16) ? g : ((g == 1) ? 1 : 0);
    return *reinterpret_cast<const mx_u32x4*>(planes + (arr * LP + t * 16 + l15) * 16);
}
// B fragments of this lane's own row (8 values already split): first MFMA pairs [h|h] with the planes [Xh|Xl] (D = 16) or
// [h|h|l|0] with [Xh|Xl|Xh|.] (D = 8); the second MFMA (D = 16 only) pairs [l|0] with [Xh|.].
template <int D>
__device__ __forceinline__ void mx_col_frags(const mx_u32x4& hi, const mx_u32x4& lo, int g, mx_u32x4& b1, mx_u32x4& b2) {
    const mx_u32x4 z = {0u, 0u, 0u, 0u};
    if (D == 16) {
        b1 = hi;
        b2 = (g < 2) ? lo : z;
    } else {
        b1 = (g < 2) ? hi : ((g == 2) ? lo : z);
        b2 = z;
    }
}

template <int NKT, bool AFFINE, int D, bool RAGGED>
__global__ __launch_bounds__(256) void mha_fwd_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int64_t ld, int L, int h, float scale,
                                                         const float* __restrict__ key_scale, const float* __restrict__ key_shift,
                                                         float* __restrict__ o, int64_t ldo, float* __restrict__ lse, int oimg, const MxImg im) {
    static_assert(NKT % 2 == 0, "key tiles come in pairs (32-deep PV reduction)");
    float vmax = 0.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LP = NKT * 16;
    constexpr int NQ = (NKT + 3) / 4;
    const int nkt = (L + 15) >> 4;                       // RAGGED = false: L == LP, no padded keys to mask
    const float qmul = AFFINE ? scale : scale * MX_LOG2E;
    unsigned char* Kp = smem;
    unsigned char* Vt = Kp + mx_rowplanes_bytes(LP, D);
    float* ksc = reinterpret_cast<float*>(Vt + mx_tplanes_bytes(LP));
    float* ksh = ksc + LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int TS = mx_tstride(LP);

    if (RAGGED || D < 16) {                 // padded keys and (d = 8) the unused rows of V^T must read as zero
        for (int i = tid; i < (mx_rowplanes_bytes(LP, D) + mx_tplanes_bytes(LP)) / 16; i += 256)
            reinterpret_cast<mx_u32x4*>(smem)[i] = mx_u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    if (AFFINE) {
        for (int i = tid; i < LP; i += 256) {
            ksc[i] = (i < L) ? key_scale[i] * MX_LOG2E : 1.f;      // z in log2 units
            ksh[i] = (i < L) ? key_shift[i] * MX_LOG2E : 0.f;
        }
    }
    mx_stage_rows<D>(Kp, k, ld, b, L, LP, hh, 1.f, tid);
    mx_stage_transposed<D>(Vt, v, ld, b, L, LP, hh, tid);
    __syncthreads();

#pragma unroll 1
    for (int i = 0; i < NQ; ++i) {
        const int qt = wave + 4 * i;
        if (qt >= nkt) break;
        const int qrow = qt * 16 + l15;
        // this lane's query fragment: columns 8*(g&1).. of head hh (D = 16) or all 8 (D = 8), scaled, split
        mx_u32x4 qh = {0u, 0u, 0u, 0u}, ql = {0u, 0u, 0u, 0u};
        if (qrow < L) {
            const float* p = q + ((int64_t)b * L + qrow) * ld + hh * D + ((D == 16) ? 8 * (g & 1) : 0);
            const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
            const float qv[8] = {a.x * qmul, a.y * qmul, a.z * qmul, a.w * qmul, c.x * qmul, c.y * qmul, c.z * qmul, c.w * qmul};
            mx_split8(qv, qh, ql);
        }
        mx_u32x4 b1, b2;
        mx_col_frags<D>(qh, ql, g, b1, b2);
        f32x4 p[NKT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const mx_u32x4 a = mx_row_frag<D>(Kp, LP, kt, l15, g);
            acc = mx_mfma(a, b1, acc);
            if (D == 16) acc = mx_mfma(a, b2, acc);
            if (AFFINE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(acc[r], ksc[kt * 16 + 4 * g + r], ksh[kt * 16 + 4 * g + r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (RAGGED && kt * 16 + 4 * g + r >= L) acc[r] = -INFINITY;
                m = fmaxf(m, acc[r]);
            }
            p[kt] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
        f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKT / 2; ++j) {
            float e8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e8[r] = __builtin_amdgcn_exp2f(p[2 * j][r] - m);
                e8[4 + r] = __builtin_amdgcn_exp2f(p[2 * j + 1][r] - m);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) sum += e8[r];
            mx_u32x4 ph, pl;
            mx_split8(e8, ph, pl);
            const unsigned char* va = Vt + l15 * TS + (32 * j + 8 * g) * 2;
            const mx_u32x4 vh = *reinterpret_cast<const mx_u32x4*>(va), vl = *reinterpret_cast<const mx_u32x4*>(va + 16 * TS);
            oa = mx_mfma(vh, ph, oa);
            ob = mx_mfma(vh, pl, ob);
            oa = mx_mfma(vl, ph, oa);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (qrow < L) {
            if (4 * g < D)
                mx_store_act4(o, (int64_t)b * L + qrow, ldo, hh * D + 4 * g, oimg, (oa[0] + ob[0]) * inv, (oa[1] + ob[1]) * inv,
                              (oa[2] + ob[2]) * inv, (oa[3] + ob[3]) * inv, im, vmax);
            if (g == 0) lse[((int64_t)b * h + hh) * L + qrow] = m * MX_LN2 + __logf(sum);
        }
    }
    if (oimg) of_amax_commit(im.oamax, vmax);
}

// ---- backward ------------------------------------------------------------------------------------------------------
// Two kernels as in the fp32 form (each stages what ITS sweep re-reads; tiles are recomputed from the saved log-sum-exp):
//   mha_bwd_dq_x3_kernel : K and V row planes + K^T planes in LDS; a wave owns query tiles:
//       S^T = K Q^T, dP^T = V dO^T (2 MFMAs each), dS^T in registers, dQ^T += K^T dS^T (3 MFMAs per 32 keys)
//   mha_bwd_dkv_x3_kernel: Q (pre-scaled) and dO row planes + their transposes in LDS; a wave owns key tiles:
//       S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS; also the logits_bn column sums (dz_partial).
constexpr int MX_DQ_NT = 512;      // eight waves per workgroup, as for the dK/dV kernel below
template <int NKT, bool AFFINE, int D, bool RAGGED>
__global__ __launch_bounds__(MX_DQ_NT) void mha_bwd_dq_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, int64_t ld, const float* __restrict__ o,
                                                            const float* __restrict__ dout, int64_t ldo,
                                                            const float* __restrict__ lse, int L, int h, float scale,
                                                            const float* __restrict__ key_scale, const float* __restrict__ key_shift,
                                                            float* __restrict__ dq, int64_t ldd, const float* __restrict__ corr_a,
                                                            const float* __restrict__ corr_b, int img, int oimg, const MxImg im) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float vmax = 0.f;
    constexpr int LP = NKT * 16;
    const int nkt = (L + 15) >> 4;
    const float qmul = AFFINE ? scale : scale * MX_LOG2E;      // scores in log2 units (AFFINE: z is converted instead)
    unsigned char* Kp = smem;
    unsigned char* Vp = Kp + mx_rowplanes_bytes(LP, D);
    unsigned char* Kt = Vp + mx_rowplanes_bytes(LP, D);
    float* ksc = reinterpret_cast<float*>(Kt + mx_tplanes_bytes(LP));
    float* ksh = ksc + LP;
    float* cas = ksh + LP;
    float* cbs = cas + LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int TS = mx_tstride(LP);

    if (RAGGED || D < 16) {
        for (int i = tid; i < (2 * mx_rowplanes_bytes(LP, D) + mx_tplanes_bytes(LP)) / 16; i += MX_DQ_NT)
            reinterpret_cast<mx_u32x4*>(smem)[i] = mx_u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    if (AFFINE) {
        for (int i = tid; i < LP; i += MX_DQ_NT) {
            ksc[i] = (i < L) ? key_scale[i] : 1.f;
            ksh[i] = (i < L) ? key_shift[i] : 0.f;
            cas[i] = (corr_a && i < L) ? corr_a[i] : 0.f;
            cbs[i] = (corr_b && i < L) ? corr_b[i] : 0.f;
        }
    }
    mx_stage_rows<D>(Kp, k, ld, b, L, LP, hh, 1.f, tid, MX_DQ_NT);
    mx_stage_rows<D>(Vp, v, ld, b, L, LP, hh, 1.f, tid, MX_DQ_NT);
    mx_stage_transposed<D>(Kt, k, ld, b, L, LP, hh, tid, MX_DQ_NT);
    __syncthreads();

#pragma unroll 1
    for (int qt = wave; qt < nkt; qt += MX_DQ_NT / 64) {
        const int qrow = qt * 16 + l15;
        const bool qok = qrow < L;
        mx_u32x4 qh = {0u, 0u, 0u, 0u}, ql = qh, gh = qh, gl = qh;
        float dpart = 0.f;
        if (qok) {
            const int c0 = hh * D + ((D == 16) ? 8 * (g & 1) : 0);
            const float* qp = q + ((int64_t)b * L + qrow) * ld + c0;
            const float4 a = *reinterpret_cast<const float4*>(qp), c = *reinterpret_cast<const float4*>(qp + 4);
            const float qv[8] = {a.x * qmul, a.y * qmul, a.z * qmul, a.w * qmul, c.x * qmul, c.y * qmul, c.z * qmul, c.w * qmul};
            mx_split8(qv, qh, ql);
            const int64_t off = ((int64_t)b * L + qrow) * ldo + c0;
            const float4 ga = *reinterpret_cast<const float4*>(dout + off), gc = *reinterpret_cast<const float4*>(dout + off + 4);
            float4 oa, oc;
            mx_load_o8(o, (int64_t)b * L + qrow, ldo, c0, oimg, oa, oc, im);
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
            mx_split8(gv, gh, gl);
            dpart = ga.x * oa.x + ga.y * oa.y + ga.z * oa.z + ga.w * oa.w + gc.x * oc.x + gc.y * oc.y + gc.z * oc.z + gc.w * oc.w;
        }
        if (D == 16) dpart += __shfl_xor(dpart, 16, 64);       // the two 8-column halves live in lane groups g and g^1
        const float dqv = dpart;                               // D_q = <dO_q, O_q>
        const float lq = qok ? lse[((int64_t)b * h + hh) * L + qrow] * MX_LOG2E : INFINITY;
        mx_u32x4 qb1, qb2, gb1, gb2;
        mx_col_frags<D>(qh, ql, g, qb1, qb2);
        mx_col_frags<D>(gh, gl, g, gb1, gb2);
        f32x4 dqa = {0.f, 0.f, 0.f, 0.f}, dqb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int j = 0; j < NKT / 2; ++j) {
            float ds8[8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * j + t;
                const mx_u32x4 ka = mx_row_frag<D>(Kp, LP, kt, l15, g), va = mx_row_frag<D>(Vp, LP, kt, l15, g);
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                st = mx_mfma(ka, qb1, st);
                dp = mx_mfma(va, gb1, dp);
                if (D == 16) {
                    st = mx_mfma(ka, qb2, st);
                    dp = mx_mfma(va, gb2, dp);
                }
                float z[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z[r] = st[r];
                    if (AFFINE) z[r] = fmaf(st[r], ksc[kt * 16 + 4 * g + r], ksh[kt * 16 + 4 * g + r]) * MX_LOG2E;
                }
                if (RAGGED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * 16 + 4 * g + r >= L) z[r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(z[r] - lq);
                    float dsv = pr * (dp[r] - dqv);
                    if (AFFINE) {
                        const int key = kt * 16 + 4 * g + r;
                        dsv = dsv * ksc[key] - cas[key] - st[r] * cbs[key];
                    }
                    ds8[4 * t + r] = dsv;
                }
            }
            mx_u32x4 dh, dl;
            mx_split8(ds8, dh, dl);
            const unsigned char* kc = Kt + l15 * TS + (32 * j + 8 * g) * 2;
            const mx_u32x4 kh = *reinterpret_cast<const mx_u32x4*>(kc), kl = *reinterpret_cast<const mx_u32x4*>(kc + 16 * TS);
            dqa = mx_mfma(kh, dh, dqa);          // dQ^T[dd, q] += K^T[dd, keys] dS^T[keys, q]
            dqb = mx_mfma(kh, dl, dqb);
            dqa = mx_mfma(kl, dh, dqa);
        }
        if (qok && 4 * g < D)
            mx_store_grad4(dq, ((int64_t)b * L + qrow) * ldd + hh * D + 4 * g, img, (dqa[0] + dqb[0]) * scale, (dqa[1] + dqb[1]) * scale,
                           (dqa[2] + dqb[2]) * scale, (dqa[3] + dqb[3]) * scale, im, vmax);
    }
    if (img) of_amax_commit(im.gamax, vmax);
}

// Eight waves per (batch, head): the workgroup's LDS (66 KB at L = 256: two workgroups per CU) is the same for four or eight
// waves, and 110 registers fit four waves per SIMD -- twice the waves to hide the MFMA -> exp2 / split -> MFMA chain behind.
// Past L = 256 (NKT >= 20: 85 KB and more) only ONE workgroup fits a CU: sixteen waves there.
#ifndef LPM_DKV_NT_LONG
#define LPM_DKV_NT_LONG 1024
#endif
__host__ __device__ constexpr int mx_dkv_nt(int nkt) { return nkt >= 20 ? LPM_DKV_NT_LONG : 512; }
// CORR = false: no correction vectors (the one-pass logits_bn backward's first launch: the batch statistics' share of dk is repaired
// afterwards) -- two fewer VALU operations per score in a pass whose vector pipe is ~94 % busy (profiles/pmc_r03_mha_bn)
template <int NKT, bool AFFINE, int D, bool CORR = true>
__global__ __launch_bounds__(mx_dkv_nt(NKT)) void mha_bwd_dkv_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int64_t ld, const float* __restrict__ o,
                                                             const float* __restrict__ dout, int64_t ldo,
                                                             const float* __restrict__ lse, int L, int h, float scale,
                                                             const float* __restrict__ key_scale, const float* __restrict__ key_shift,
                                                             float* __restrict__ dk, float* __restrict__ dv, int64_t ldd,
                                                             const float* __restrict__ corr_a, const float* __restrict__ corr_b,
                                                             float* __restrict__ dz_partial, int img, int oimg, const MxImg im) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float vmax = 0.f;
    constexpr int LP = NKT * 16;
    constexpr int MX_DKV_NT = mx_dkv_nt(NKT);
    constexpr int NH = D / 8;
    const int nkt = (L + 15) >> 4;
    const float qmul = AFFINE ? scale : scale * MX_LOG2E;       // scores in log2 units (AFFINE: z is converted instead)
    unsigned char* Qp = smem;                                   // qmul * Q
    unsigned char* Gp = Qp + mx_rowplanes_bytes(LP, D);         // dO
    unsigned char* Qt = Gp + mx_rowplanes_bytes(LP, D);
    unsigned char* Gt = Qt + mx_tplanes_bytes(LP);
    float* lses = reinterpret_cast<float*>(Gt + mx_tplanes_bytes(LP));
    float* Dq = lses + LP;
    const bool stats_only = (dk == nullptr);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int TS = mx_tstride(LP);

    if (L != LP || D < 16) {
        for (int i = tid; i < (2 * mx_rowplanes_bytes(LP, D) + 2 * mx_tplanes_bytes(LP)) / 16; i += MX_DKV_NT)
            reinterpret_cast<mx_u32x4*>(smem)[i] = mx_u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < LP; i += MX_DKV_NT) {
        lses[i] = (i < L) ? lse[((int64_t)b * h + hh) * L + i] * MX_LOG2E : INFINITY;     // padded queries -> p = 0
        Dq[i] = 0.f;
    }
    __syncthreads();
    // Q (scaled) and dO: row planes and transposed planes from ONE pass over global memory; D_q = <dO_q, O_q>
    for (int i0 = 0; i0 < L * NH; i0 += MX_DKV_NT) {
        const int i = i0 + tid;
        float part = 0.f;
        int row = 0;
        if (i < L * NH) {
            row = i / NH;
            const int hf = i % NH;
            const float* qp = q + ((int64_t)b * L + row) * ld + hh * D + 8 * hf;
            const int64_t off = ((int64_t)b * L + row) * ldo + hh * D + 8 * hf;
            const float4 qa = *reinterpret_cast<const float4*>(qp), qc = *reinterpret_cast<const float4*>(qp + 4);
            const float4 ga = *reinterpret_cast<const float4*>(dout + off), gc = *reinterpret_cast<const float4*>(dout + off + 4);
            float4 oa, oc;
            mx_load_o8(o, (int64_t)b * L + row, ldo, hh * D + 8 * hf, oimg, oa, oc, im);
            part = ga.x * oa.x + ga.y * oa.y + ga.z * oa.z + ga.w * oa.w + gc.x * oc.x + gc.y * oc.y + gc.z * oc.z + gc.w * oc.w;
            const float qv[8] = {qa.x * qmul, qa.y * qmul, qa.z * qmul, qa.w * qmul, qc.x * qmul, qc.y * qmul, qc.z * qmul, qc.w * qmul};
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
            mx_u32x4 qh, ql, gh, gl;
            mx_split8(qv, qh, ql);
            mx_split8(gv, gh, gl);
            *reinterpret_cast<mx_u32x4*>(Qp + ((0 * NH + hf) * LP + row) * 16) = qh;
            *reinterpret_cast<mx_u32x4*>(Qp + ((1 * NH + hf) * LP + row) * 16) = ql;
            *reinterpret_cast<mx_u32x4*>(Gp + ((0 * NH + hf) * LP + row) * 16) = gh;
            *reinterpret_cast<mx_u32x4*>(Gp + ((1 * NH + hf) * LP + row) * 16) = gl;
            const int pos = mx_perm(row);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dd = 8 * hf + e, sh = (e & 1) * 16;
                *reinterpret_cast<unsigned short*>(Qt + dd * TS + pos * 2) = (unsigned short)(qh[e >> 1] >> sh);
                *reinterpret_cast<unsigned short*>(Qt + 16 * TS + dd * TS + pos * 2) = (unsigned short)(ql[e >> 1] >> sh);
                *reinterpret_cast<unsigned short*>(Gt + dd * TS + pos * 2) = (unsigned short)(gh[e >> 1] >> sh);
                *reinterpret_cast<unsigned short*>(Gt + 16 * TS + dd * TS + pos * 2) = (unsigned short)(gl[e >> 1] >> sh);
            }
        }
        if (NH == 2) part += __shfl_xor(part, 1, 64);          // the two halves of a row sit in adjacent threads
        if (i < L * NH && (i % NH) == 0) Dq[row] = part;
    }
    __syncthreads();

#pragma unroll 1
    for (int kt = wave; kt < nkt; kt += MX_DKV_NT / 64) {
        const int krow = kt * 16 + l15;
        const bool kok = krow < L;
        mx_u32x4 kh = {0u, 0u, 0u, 0u}, kl = kh, vh = kh, vl = kh;
        if (kok) {
            const int64_t off = ((int64_t)b * L + krow) * ld + hh * D + ((D == 16) ? 8 * (g & 1) : 0);
            const float4 ka = *reinterpret_cast<const float4*>(k + off), kc = *reinterpret_cast<const float4*>(k + off + 4);
            const float4 va = *reinterpret_cast<const float4*>(v + off), vc = *reinterpret_cast<const float4*>(v + off + 4);
            const float kv[8] = {ka.x, ka.y, ka.z, ka.w, kc.x, kc.y, kc.z, kc.w};
            const float vv[8] = {va.x, va.y, va.z, va.w, vc.x, vc.y, vc.z, vc.w};
            mx_split8(kv, kh, kl);
            mx_split8(vv, vh, vl);
        }
        mx_u32x4 kb1, kb2, vb1, vb2;
        mx_col_frags<D>(kh, kl, g, kb1, kb2);
        mx_col_frags<D>(vh, vl, g, vb1, vb2);
        const float sck = (key_scale && kok) ? key_scale[krow] : 1.f, shk = (key_shift && kok) ? key_shift[krow] : 0.f;
        const float cak = (corr_a && kok) ? corr_a[krow] : 0.f, cbk = (corr_b && kok) ? corr_b[krow] : 0.f;
        f32x4 dka = {0.f, 0.f, 0.f, 0.f}, dkb = dka, dva = dka, dvb = dka;
        float zs = 0.f, zq = 0.f;
#pragma unroll 2
        for (int j = 0; j < NKT / 2; ++j) {
            float p8[8], ds8[8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * j + t;
                const mx_u32x4 qa = mx_row_frag<D>(Qp, LP, qt, l15, g), ga = mx_row_frag<D>(Gp, LP, qt, l15, g);
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                st = mx_mfma(qa, kb1, st);                 // S[q, key] (scale folded into Q)
                dp = mx_mfma(ga, vb1, dp);                 // dP[q, key] = dO V^T
                if (D == 16) {
                    st = mx_mfma(qa, kb2, st);
                    dp = mx_mfma(ga, vb2, dp);
                }
                const float4 lq4v = *reinterpret_cast<const float4*>(lses + qt * 16 + 4 * g);
                const float4 dq4v = *reinterpret_cast<const float4*>(Dq + qt * 16 + 4 * g);
                const float lq4[4] = {lq4v.x, lq4v.y, lq4v.z, lq4v.w}, dq4[4] = {dq4v.x, dq4v.y, dq4v.z, dq4v.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qr = qt * 16 + 4 * g + r;
                    const float sraw = st[r];
                    float z = sraw;
                    if (AFFINE) z = fmaf(sraw, sck, shk) * MX_LOG2E;
                    if (!kok) z = -INFINITY;
                    const float pr = __builtin_amdgcn_exp2f(z - lq4[r]);
                    const float dz = pr * (dp[r] - dq4[r]);
                    float dsv = dz;
                    if (AFFINE) {
                        zs += dz;
                        zq = fmaf(dz, sraw, zq);
                        if constexpr (CORR) dsv = (qr < L) ? dz * sck - cak - sraw * cbk : 0.f;
                        else dsv = (qr < L) ? dz * sck : 0.f;
                    }
                    p8[4 * t + r] = pr;
                    ds8[4 * t + r] = dsv;
                }
            }
            if (!stats_only) {
                mx_u32x4 ph, pl, dh, dl;
                mx_split8(p8, ph, pl);
                mx_split8(ds8, dh, dl);
                const int offt = l15 * TS + (32 * j + 8 * g) * 2;
                const mx_u32x4 gth = *reinterpret_cast<const mx_u32x4*>(Gt + offt), gtl = *reinterpret_cast<const mx_u32x4*>(Gt + 16 * TS + offt);
                const mx_u32x4 qth = *reinterpret_cast<const mx_u32x4*>(Qt + offt), qtl = *reinterpret_cast<const mx_u32x4*>(Qt + 16 * TS + offt);
                dva = mx_mfma(gth, ph, dva);               // dV^T[dd, key] += dO^T[dd, q] P[q, key]
                dvb = mx_mfma(gth, pl, dvb);
                dva = mx_mfma(gtl, ph, dva);
                dka = mx_mfma(qth, dh, dka);               // dK^T[dd, key] += (scale Q)^T[dd, q] dS[q, key]
                dkb = mx_mfma(qth, dl, dkb);
                dka = mx_mfma(qtl, dh, dka);
            }
        }
        if (!stats_only && kok && 4 * g < D) {
            const int64_t off = ((int64_t)b * L + krow) * ldd + hh * D + 4 * g;
            const float kmul = AFFINE ? 1.f : MX_LN2;      // the staged Q carried log2(e)
            mx_store_dk4(dk, (int64_t)b * L + krow, ldd, hh * D + 4 * g, img, (dka[0] + dkb[0]) * kmul, (dka[1] + dkb[1]) * kmul, (dka[2] + dkb[2]) * kmul, (dka[3] + dkb[3]) * kmul, im, vmax);
            mx_store_grad4(dv, off, img, dva[0] + dvb[0], dva[1] + dvb[1], dva[2] + dvb[2], dva[3] + dvb[3], im, vmax);
        }
        if (dz_partial) {
            zs += __shfl_xor(zs, 16, 64); zs += __shfl_xor(zs, 32, 64);
            zq += __shfl_xor(zq, 16, 64); zq += __shfl_xor(zq, 32, 64);
            if (g == 0 && kok) {
                float* out = dz_partial + ((int64_t)b * h + hh) * 2 * L;
                out[krow] = zs;
                out[L + krow] = zq;
            }
        }
    }
    if (img) of_amax_commit(im.gamax, vmax);
}

// ---- backward on two fp16 terms (round 6; the arithmetic is described with the helpers at the top of the file) -----------------------
// Same two sweeps, same LDS geometry, same thread counts as the three-term kernels above.  Valid where dS is LINEAR in dO (the scale
// is taken from max |dO|): the plain attention, and logits_bn's launches without correction vectors (corr_a == NULL: the one-pass
// backward's dkv kernel) -- with corrections the batch statistics' terms enter dS at the size of the GLOBAL gradient whatever this
// query's / head's own dO is, and the launcher keeps the three-term kernels (bf16 has fp32's range).
// Measured (tools/time_mha_bwd.py, rocprofv3, profiles/r06_k4_backward_terms.md; B = 80, h = 64, d = 16), whole backward: L = 256
// 543 -> 512 us (-6 %), L = 300 with logits_bn (dkv only: the dq pass carries corrections) 1037 -> 1016 us (-2 %); a cfg-2 step 6.72 ->
// 6.70 ms.  With dP on one fp16 MFMA as well (V rounded once) it was 557 -> 508 us -- and 8e-4 off where attention is peaked.
// 18-27 % of the MFMAs and 25-38 % of the VALU operations per score buy 2-6 %: the sweeps are bound by neither count.  TPW = 2 (a wave
// runs two tiles side by side through ONE stream of fragment reads: half the LDS traffic per score, twice the independent work per
// wave, but 140 registers = half the waves) measured dkv 290 us against 234, dq 196 against 207 at L = 256 and dq 401 us against 335
// at L = 300 -- what the kernels live on is the number of waves a SIMD can switch between while one waits for its MFMA -> exp2 ->
// convert -> MFMA chain, and that is capped at four by the 66 KB of staged operands per (batch, head).  The same measurement answers
// the merged dq + dkv sweep (DESIGN: ~174 registers, two waves per SIMD): it would run at TPW = 2's occupancy.  NOT the default
// (lpm_mha_bwd_set_terms): 0.3 % of a step does not pay for gradients 2-3e-4 away from the three-term form's (1e-5 from fp64), and
// tests/test_gpu_attention_modules.py holds its module to 2e-4.  TPW = 2 is reachable for L = 256, d = 16 only (LPM_MHA_BWD_TPW=2).
// threads per workgroup: one tile per wave -- as the three-term kernels (dq 512; dkv 512, 1024 from NKT = 20 on); two -- one wave per task
template <int NKT, int TPW, bool DKV> __host__ __device__ constexpr int mx_h_nt() {
    constexpr int w = (NKT + TPW - 1) / TPW;
    return TPW == 1 ? (DKV ? mx_dkv_nt(NKT) : MX_DQ_NT) : (w <= 4 ? 256 : (w >= 16 ? 1024 : 64 * w));
}
template <int NKT, bool AFFINE, int D, bool RAGGED, int TPW>
__global__ __launch_bounds__((mx_h_nt<NKT, TPW, false>())) void mha_bwd_dq_h_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, int64_t ld, const float* __restrict__ o,
                                                            const float* __restrict__ dout, int64_t ldo,
                                                            const float* __restrict__ lse, int L, int h, float scale,
                                                            const float* __restrict__ key_scale, const float* __restrict__ key_shift,
                                                            float* __restrict__ dq, int64_t ldd, const float* __restrict__ corr_a,
                                                            const float* __restrict__ corr_b, int img, int oimg, const MxImg im) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float vmax = 0.f;
    constexpr int LP = NKT * 16;
    constexpr int NT = mx_h_nt<NKT, TPW, false>();
    const int nkt = (L + 15) >> 4;
    const float qmul = AFFINE ? scale : scale * MX_LOG2E;      // scores in log2 units (AFFINE: z is converted instead)
    unsigned char* Kp = smem;                                  // K rows, split-bf16 (hi, lo): the scores stay exact
    unsigned char* Vp = Kp + mx_rowplanes_bytes(LP, D);        // V rows, split-bf16 (hi, lo): dP stays exact (it meets D_q in a difference)
    unsigned char* Kt = Vp + mx_rowplanes_bytes(LP, D);        // K^T, fp16 (hi, lo)
    float* ksc = reinterpret_cast<float*>(Kt + mx_tplanes_bytes(LP));
    float* ksh = ksc + LP;
    float* cas = ksh + LP;
    float* cbs = cas + LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int TS = mx_tstride(LP);

    if (RAGGED || D < 16) {
        for (int i = tid; i < (2 * mx_rowplanes_bytes(LP, D) + mx_tplanes_bytes(LP)) / 16; i += NT)
            reinterpret_cast<mx_u32x4*>(smem)[i] = mx_u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    if (AFFINE) {
        for (int i = tid; i < LP; i += NT) {
            ksc[i] = (i < L) ? key_scale[i] : 1.f;
            ksh[i] = (i < L) ? key_shift[i] : 0.f;
            cas[i] = (corr_a && i < L) ? corr_a[i] : 0.f;
            cbs[i] = (corr_b && i < L) ? corr_b[i] : 0.f;
        }
    }
    mx_stage_rows<D>(Kp, k, ld, b, L, LP, hh, 1.f, tid, NT);
    mx_stage_rows<D>(Vp, v, ld, b, L, LP, hh, 1.f, tid, NT);
    mx_stage_transposed_h<D>(Kt, k, ld, b, L, LP, hh, tid, NT);
    __syncthreads();

    const mx_u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int t0 = wave * TPW; t0 < nkt; t0 += (NT / 64) * TPW) {
        mx_u32x4 qb1[TPW], qb2[TPW], gb1[TPW], gb2[TPW];
        float dqv[TPW], lq[TPW], sq[TPW], sqinv[TPW];
        f32x4 dqa[TPW], dqb[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int qrow = (t0 + u) * 16 + l15;
            const bool qok = qrow < L;
            mx_u32x4 qh = z4, ql = z4, gh = z4, gl = z4;
            float dpart = 0.f, gmax = 0.f;
            float gv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (qok) {
                const int c0 = hh * D + ((D == 16) ? 8 * (g & 1) : 0);
                const float* qp = q + ((int64_t)b * L + qrow) * ld + c0;
                const float4 a = *reinterpret_cast<const float4*>(qp), c = *reinterpret_cast<const float4*>(qp + 4);
                const float qv[8] = {a.x * qmul, a.y * qmul, a.z * qmul, a.w * qmul, c.x * qmul, c.y * qmul, c.z * qmul, c.w * qmul};
                mx_split8(qv, qh, ql);
                const int64_t off = ((int64_t)b * L + qrow) * ldo + c0;
                const float4 ga = *reinterpret_cast<const float4*>(dout + off), gc = *reinterpret_cast<const float4*>(dout + off + 4);
                float4 oa, oc;
                mx_load_o8(o, (int64_t)b * L + qrow, ldo, c0, oimg, oa, oc, im);
                gv[0] = ga.x; gv[1] = ga.y; gv[2] = ga.z; gv[3] = ga.w; gv[4] = gc.x; gv[5] = gc.y; gv[6] = gc.z; gv[7] = gc.w;
                gmax = of_amax8(0.f, gv);
                dpart = ga.x * oa.x + ga.y * oa.y + ga.z * oa.z + ga.w * oa.w + gc.x * oc.x + gc.y * oc.y + gc.z * oc.z + gc.w * oc.w;
            }
            if (D == 16) {                                         // the two 8-column halves live in lane groups g and g^1
                dpart += __shfl_xor(dpart, 16, 64);
                gmax = fmaxf(gmax, __shfl_xor(gmax, 16, 64));
            }
            // this query's power-of-two scale s_q (max |dO_q| s_q in [16, 32)).  A lane owns one query COLUMN of every product below, so
            // dP, D_q, dS and dQ of the query all carry s_q and the store takes it out again.
            mx_pow2_scale(gmax, sq[u], sqinv[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] *= sq[u];            // (a power of two: exact; bf16 has fp32's range)
            mx_split8(gv, gh, gl);
            dqv[u] = dpart * sq[u];                                // D_q = <dO_q, O_q>
            lq[u] = qok ? lse[((int64_t)b * h + hh) * L + qrow] * MX_LOG2E : INFINITY;
            mx_col_frags<D>(qh, ql, g, qb1[u], qb2[u]);
            mx_col_frags<D>(gh, gl, g, gb1[u], gb2[u]);
            dqa[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            dqb[u] = dqa[u];
        }
#pragma unroll 2
        for (int j = 0; j < NKT / 2; ++j) {
            float ds8[TPW][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * j + t;
                const mx_u32x4 ka = mx_row_frag<D>(Kp, LP, kt, l15, g);
                const mx_u32x4 va = mx_row_frag<D>(Vp, LP, kt, l15, g);
                float sc4[4], sh4[4], ca4[4], cb4[4];
                if (AFFINE) {
                    const float4 a = *reinterpret_cast<const float4*>(ksc + kt * 16 + 4 * g), c = *reinterpret_cast<const float4*>(ksh + kt * 16 + 4 * g);
                    const float4 e = *reinterpret_cast<const float4*>(cas + kt * 16 + 4 * g), f = *reinterpret_cast<const float4*>(cbs + kt * 16 + 4 * g);
                    sc4[0] = a.x; sc4[1] = a.y; sc4[2] = a.z; sc4[3] = a.w; sh4[0] = c.x; sh4[1] = c.y; sh4[2] = c.z; sh4[3] = c.w;
                    ca4[0] = e.x; ca4[1] = e.y; ca4[2] = e.z; ca4[3] = e.w; cb4[0] = f.x; cb4[1] = f.y; cb4[2] = f.z; cb4[3] = f.w;
                }
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    st = mx_mfma(ka, qb1[u], st);
                    dp = mx_mfma(va, gb1[u], dp);                  // dP^T s_q = V (dO s_q)^T: three terms, as the scores
                    if (D == 16) {
                        st = mx_mfma(ka, qb2[u], st);
                        dp = mx_mfma(va, gb2[u], dp);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z = st[r];
                        if (AFFINE) z = fmaf(st[r], sc4[r], sh4[r]) * MX_LOG2E;
                        if (RAGGED && kt * 16 + 4 * g + r >= L) z = -INFINITY;
                        const float pr = __builtin_amdgcn_exp2f(z - lq[u]);
                        float dsv = pr * (dp[r] - dqv[u]);
                        if (AFFINE) dsv = fmaf(dsv, sc4[r], -sq[u] * fmaf(st[r], cb4[r], ca4[r]));
                        ds8[u][4 * t + r] = dsv;
                    }
                }
            }
            const unsigned char* kc = Kt + l15 * TS + (32 * j + 8 * g) * 2;
            const mx_u32x4 kh = *reinterpret_cast<const mx_u32x4*>(kc), kl = *reinterpret_cast<const mx_u32x4*>(kc + 16 * TS);
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const mx_u32x4 dh = mx_round8_h(ds8[u]);           // dS rounded once; K^T exact: two products, two accumulators
                dqa[u] = mx_mfma_h(kh, dh, dqa[u]);                // dQ^T[dd, q] += K^T[dd, keys] dS^T[keys, q]
                dqb[u] = mx_mfma_h(kl, dh, dqb[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int qrow = (t0 + u) * 16 + l15;
            const float omul = scale * sqinv[u];
            if (qrow < L && 4 * g < D)
                mx_store_grad4(dq, ((int64_t)b * L + qrow) * ldd + hh * D + 4 * g, img, (dqa[u][0] + dqb[u][0]) * omul, (dqa[u][1] + dqb[u][1]) * omul,
                               (dqa[u][2] + dqb[u][2]) * omul, (dqa[u][3] + dqb[u][3]) * omul, im, vmax);
        }
    }
    if (img) of_amax_commit(im.gamax, vmax);
}

template <int NKT, bool AFFINE, int D, bool CORR, int TPW>
__global__ __launch_bounds__((mx_h_nt<NKT, TPW, true>())) void mha_bwd_dkv_h_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int64_t ld, const float* __restrict__ o,
                                                             const float* __restrict__ dout, int64_t ldo,
                                                             const float* __restrict__ lse, int L, int h, float scale,
                                                             const float* __restrict__ key_scale, const float* __restrict__ key_shift,
                                                             float* __restrict__ dk, float* __restrict__ dv, int64_t ldd,
                                                             const float* __restrict__ corr_a, const float* __restrict__ corr_b,
                                                             float* __restrict__ dz_partial, int img, int oimg, const MxImg im) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float vmax = 0.f;
    constexpr int LP = NKT * 16;
    constexpr int NT = mx_h_nt<NKT, TPW, true>();
    constexpr int NH = D / 8;
    const int nkt = (L + 15) >> 4;
    const float qmul = AFFINE ? scale : scale * MX_LOG2E;       // scores in log2 units (AFFINE: z is converted instead)
    unsigned char* Qp = smem;                                   // qmul * Q rows, split-bf16 (hi, lo): the scores stay exact
    unsigned char* Gp = Qp + mx_rowplanes_bytes(LP, D);         // dO s rows, split-bf16 (hi, lo): dP stays exact (it meets D_q in a difference)
    unsigned char* Qt = Gp + mx_rowplanes_bytes(LP, D);         // (qmul * Q)^T, fp16 (hi, lo)
    unsigned char* Gt = Qt + mx_tplanes_bytes(LP);              // (dO s)^T, fp16 (hi, lo)
    float* lses = reinterpret_cast<float*>(Gt + mx_tplanes_bytes(LP));
    float* Dq = lses + LP;
    unsigned* gmaxw = reinterpret_cast<unsigned*>(Dq + LP);
    const bool stats_only = (dk == nullptr);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int TS = mx_tstride(LP);

    if (L != LP || D < 16) {
        for (int i = tid; i < (2 * mx_rowplanes_bytes(LP, D) + 2 * mx_tplanes_bytes(LP)) / 16; i += NT)
            reinterpret_cast<mx_u32x4*>(smem)[i] = mx_u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < LP; i += NT) {
        lses[i] = (i < L) ? lse[((int64_t)b * h + hh) * L + i] * MX_LOG2E : INFINITY;     // padded queries -> p = 0
        Dq[i] = 0.f;
    }
    if (tid == 0) *gmaxw = 0u;
    __syncthreads();
    // ONE power-of-two scale s for the (batch, head)'s dO (the reductions below run over queries, so it cannot be per query): max |dO| s
    // in [16, 32); dP, D_q, dS, the statistics and dK / dV all carry s, the stores take it out.  The maximum comes from a pass of its own
    // over the head's 16 KB of dO (L2-resident for the staging pass behind it): waves join, one LDS atomic per wave.
    float gs, gsinv;
    {
        float m = 0.f;
        for (int i = tid; i < L * NH; i += NT) {
            const int64_t off = ((int64_t)b * L + i / NH) * ldo + hh * D + 8 * (i % NH);
            const float4 ga = *reinterpret_cast<const float4*>(dout + off), gc = *reinterpret_cast<const float4*>(dout + off + 4);
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
            m = of_amax8(m, gv);
        }
        m = wave_max(m);
        if (lane == 0) atomicMax(gmaxw, __float_as_uint(m));
        __syncthreads();
        mx_pow2_scale(__uint_as_float(*gmaxw), gs, gsinv);
    }
    // Q (scaled) and dO: row planes and transposed planes from ONE pass over global memory; D_q = <dO_q, O_q>
    for (int i0 = 0; i0 < L * NH; i0 += NT) {
        const int i = i0 + tid;
        float part = 0.f;
        int row = 0;
        if (i < L * NH) {
            row = i / NH;
            const int hf = i % NH;
            const float* qp = q + ((int64_t)b * L + row) * ld + hh * D + 8 * hf;
            const int64_t off = ((int64_t)b * L + row) * ldo + hh * D + 8 * hf;
            const float4 qa = *reinterpret_cast<const float4*>(qp), qc = *reinterpret_cast<const float4*>(qp + 4);
            const float4 ga = *reinterpret_cast<const float4*>(dout + off), gc = *reinterpret_cast<const float4*>(dout + off + 4);
            float4 oa, oc;
            mx_load_o8(o, (int64_t)b * L + row, ldo, hh * D + 8 * hf, oimg, oa, oc, im);
            part = ga.x * oa.x + ga.y * oa.y + ga.z * oa.z + ga.w * oa.w + gc.x * oc.x + gc.y * oc.y + gc.z * oc.z + gc.w * oc.w;
            const float qv[8] = {qa.x * qmul, qa.y * qmul, qa.z * qmul, qa.w * qmul, qc.x * qmul, qc.y * qmul, qc.z * qmul, qc.w * qmul};
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
            const float gsv[8] = {gv[0] * gs, gv[1] * gs, gv[2] * gs, gv[3] * gs, gv[4] * gs, gv[5] * gs, gv[6] * gs, gv[7] * gs};
            mx_u32x4 qh, ql, gbh, gbl, gh, gl, qfh, qfl;
            mx_split8(qv, qh, ql);
            mx_split8(gsv, gbh, gbl);                              // rows (dP): split-bf16
            mx_split8_h(gv, gs, gh, gl);                           // transposed (dV against the rounded P): fp16 (hi, lo)
            mx_split8_h(qv, 1.f, qfh, qfl);
            *reinterpret_cast<mx_u32x4*>(Qp + ((0 * NH + hf) * LP + row) * 16) = qh;
            *reinterpret_cast<mx_u32x4*>(Qp + ((1 * NH + hf) * LP + row) * 16) = ql;
            *reinterpret_cast<mx_u32x4*>(Gp + ((0 * NH + hf) * LP + row) * 16) = gbh;
            *reinterpret_cast<mx_u32x4*>(Gp + ((1 * NH + hf) * LP + row) * 16) = gbl;
            const int pos = mx_perm(row);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dd = 8 * hf + e, sh = (e & 1) * 16;
                *reinterpret_cast<unsigned short*>(Qt + dd * TS + pos * 2) = (unsigned short)(qfh[e >> 1] >> sh);
                *reinterpret_cast<unsigned short*>(Qt + 16 * TS + dd * TS + pos * 2) = (unsigned short)(qfl[e >> 1] >> sh);
                *reinterpret_cast<unsigned short*>(Gt + dd * TS + pos * 2) = (unsigned short)(gh[e >> 1] >> sh);
                *reinterpret_cast<unsigned short*>(Gt + 16 * TS + dd * TS + pos * 2) = (unsigned short)(gl[e >> 1] >> sh);
            }
        }
        if (NH == 2) part += __shfl_xor(part, 1, 64);          // the two halves of a row sit in adjacent threads
        if (i < L * NH && (i % NH) == 0) Dq[row] = part * gs;
    }
    __syncthreads();

    const mx_u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int t0 = wave * TPW; t0 < nkt; t0 += (NT / 64) * TPW) {
        mx_u32x4 kb1[TPW], kb2[TPW], vb1[TPW], vb2[TPW];
        float sck[TPW], shk[TPW], cak[TPW], cbk[TPW], zs[TPW], zq[TPW];
        bool kok[TPW];
        f32x4 dka[TPW], dkb[TPW], dva[TPW], dvb[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int krow = (t0 + u) * 16 + l15;
            kok[u] = krow < L;
            mx_u32x4 kh = z4, kl = z4, vh = z4, vl = z4;
            if (kok[u]) {
                const int64_t off = ((int64_t)b * L + krow) * ld + hh * D + ((D == 16) ? 8 * (g & 1) : 0);
                const float4 ka = *reinterpret_cast<const float4*>(k + off), kc = *reinterpret_cast<const float4*>(k + off + 4);
                const float4 va = *reinterpret_cast<const float4*>(v + off), vc = *reinterpret_cast<const float4*>(v + off + 4);
                const float kv[8] = {ka.x, ka.y, ka.z, ka.w, kc.x, kc.y, kc.z, kc.w};
                const float vv[8] = {va.x, va.y, va.z, va.w, vc.x, vc.y, vc.z, vc.w};
                mx_split8(kv, kh, kl);
                mx_split8(vv, vh, vl);
            }
            mx_col_frags<D>(kh, kl, g, kb1[u], kb2[u]);
            mx_col_frags<D>(vh, vl, g, vb1[u], vb2[u]);
            sck[u] = (key_scale && kok[u]) ? key_scale[krow] : 1.f;
            shk[u] = (key_shift && kok[u]) ? key_shift[krow] : 0.f;
            cak[u] = ((corr_a && kok[u]) ? corr_a[krow] : 0.f) * gs;
            cbk[u] = ((corr_b && kok[u]) ? corr_b[krow] : 0.f) * gs;
            zs[u] = zq[u] = 0.f;
            dka[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            dkb[u] = dva[u] = dvb[u] = dka[u];
        }
        constexpr int UNR = (AFFINE && TPW == 2) ? 1 : 2;       // (the logits_bn form with two key tiles: 180-215 registers unrolled by two)
#pragma unroll UNR
        for (int j = 0; j < NKT / 2; ++j) {
            float p8[TPW][8], ds8[TPW][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * j + t;
                const mx_u32x4 qa = mx_row_frag<D>(Qp, LP, qt, l15, g), ga = mx_row_frag<D>(Gp, LP, qt, l15, g);
                const float4 lq4v = *reinterpret_cast<const float4*>(lses + qt * 16 + 4 * g);
                const float4 dq4v = *reinterpret_cast<const float4*>(Dq + qt * 16 + 4 * g);
                const float lq4[4] = {lq4v.x, lq4v.y, lq4v.z, lq4v.w}, dq4[4] = {dq4v.x, dq4v.y, dq4v.z, dq4v.w};
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    st = mx_mfma(qa, kb1[u], st);              // S[q, key] (scale folded into Q)
                    dp = mx_mfma(ga, vb1[u], dp);              // dP[q, key] s = (dO s) V^T: three terms, as the scores
                    if (D == 16) {
                        st = mx_mfma(qa, kb2[u], st);
                        dp = mx_mfma(ga, vb2[u], dp);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sraw = st[r];
                        float z = sraw;
                        if (AFFINE) z = fmaf(sraw, sck[u], shk[u]) * MX_LOG2E;
                        if (!kok[u]) z = -INFINITY;
                        const float pr = __builtin_amdgcn_exp2f(z - lq4[r]);
                        const float dz = pr * (dp[r] - dq4[r]);
                        float dsv = dz;
                        if (AFFINE) {
                            zs[u] += dz;
                            zq[u] = fmaf(dz, sraw, zq[u]);
                            const bool qin = qt * 16 + 4 * g + r < L;
                            if constexpr (CORR) dsv = qin ? dz * sck[u] - cak[u] - sraw * cbk[u] : 0.f;
                            else dsv = qin ? dz * sck[u] : 0.f;
                        }
                        p8[u][4 * t + r] = pr;
                        ds8[u][4 * t + r] = dsv;
                    }
                }
            }
            if (!stats_only) {
                const int offt = l15 * TS + (32 * j + 8 * g) * 2;
                const mx_u32x4 gth = *reinterpret_cast<const mx_u32x4*>(Gt + offt), gtl = *reinterpret_cast<const mx_u32x4*>(Gt + 16 * TS + offt);
                const mx_u32x4 qth = *reinterpret_cast<const mx_u32x4*>(Qt + offt), qtl = *reinterpret_cast<const mx_u32x4*>(Qt + 16 * TS + offt);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    // P in [0, 1] needs no clamp; dS is clamped to fp16's range (mx_round8_h)
                    const mx_u32x4 ph = {__builtin_bit_cast(unsigned, __builtin_convertvector(of_f2{p8[u][0], p8[u][1]}, of_h2)),
                                         __builtin_bit_cast(unsigned, __builtin_convertvector(of_f2{p8[u][2], p8[u][3]}, of_h2)),
                                         __builtin_bit_cast(unsigned, __builtin_convertvector(of_f2{p8[u][4], p8[u][5]}, of_h2)),
                                         __builtin_bit_cast(unsigned, __builtin_convertvector(of_f2{p8[u][6], p8[u][7]}, of_h2))};
                    const mx_u32x4 dh = mx_round8_h(ds8[u]);
                    dva[u] = mx_mfma_h(gth, ph, dva[u]);       // dV^T[dd, key] += (dO s)^T[dd, q] P[q, key]
                    dvb[u] = mx_mfma_h(gtl, ph, dvb[u]);
                    dka[u] = mx_mfma_h(qth, dh, dka[u]);       // dK^T[dd, key] += (scale Q)^T[dd, q] dS[q, key]
                    dkb[u] = mx_mfma_h(qtl, dh, dkb[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int krow = (t0 + u) * 16 + l15;
            if (!stats_only && kok[u] && 4 * g < D) {
                const int64_t off = ((int64_t)b * L + krow) * ldd + hh * D + 4 * g;
                const float kmul = (AFFINE ? 1.f : MX_LN2) * gsinv;      // the staged Q carried log2(e)
                mx_store_dk4(dk, (int64_t)b * L + krow, ldd, hh * D + 4 * g, img, (dka[u][0] + dkb[u][0]) * kmul, (dka[u][1] + dkb[u][1]) * kmul,
                             (dka[u][2] + dkb[u][2]) * kmul, (dka[u][3] + dkb[u][3]) * kmul, im, vmax);
                mx_store_grad4(dv, off, img, (dva[u][0] + dvb[u][0]) * gsinv, (dva[u][1] + dvb[u][1]) * gsinv, (dva[u][2] + dvb[u][2]) * gsinv,
                               (dva[u][3] + dvb[u][3]) * gsinv, im, vmax);
            }
            if (dz_partial) {
                float a = zs[u], c = zq[u];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
                if (g == 0 && kok[u]) {
                    float* out = dz_partial + ((int64_t)b * h + hh) * 2 * L;
                    out[krow] = a * gsinv;
                    out[L + krow] = c * gsinv;
                }
            }
        }
    }
    if (img) of_amax_commit(im.gamax, vmax);
}

// logits_bn backward in ONE pass over the scores (round 3).  The batch norm's backward subtracts from ds two terms that need sums over
// every (batch, head, query) first: ds[q, j] = sck[j] dz[q, j] - ca[j] - s[q, j] cb[j] (lpm_mha_bn_corrections).  The statistics pass that
// produced those sums recomputed S, P and dP for nothing else.  But both terms are AFFINE in s = (scale q) . k, so their share of
//   dK[j] = sum_q ds[q, j] (scale q)   is   - ca[j] Sq - cb[j] Qm k[j],      Sq = sum_q (scale q),  Qm = sum_q (scale q)(scale q)^T
// -- a d-vector and a d x d matrix per (batch, head).  So: the dkv kernel runs ONCE with no corrections and emits the statistics on the
// way (dz_partial), lpm_mha_bn_corrections turns them into ca / cb, the dq kernel (which walks the scores again anyway) applies them
// in place as before, and this kernel repairs dK: one workgroup per (batch, head), q staged in LDS, 4 threads per key row.
// The d x d moment matrix Qm = sum_q (scale q)(scale q)^T and the d-vector Sq = sum_q scale q of one (batch, head): q staged in LDS
// as fp32 [L][D], Qm as 4 x 4 register blocks -- thread = (row group rg of 256 / (D4 * D4), block row ab, block column cb), a row costs
// two 16-byte LDS reads for 16 products (one thread per entry walking every row: 600 scalar LDS reads per thread; the LDS pipe of a CU
// with 20 of these workgroups queued was the whole kernel) -- and the row groups meet through LDS in a fixed order.
// LDS: qs [L][D] | Qm [D][D] | Sq [D] | red [RG][D * D + D];  256 threads; Qm / Sq valid for everyone on return.
template <int D>
__device__ __forceinline__ void mx_q_moments(const float* __restrict__ q, int64_t ld, int b, int L, int hh, float scale, float* qs, float* Qm,
                                             float* Sq, float* red, int tid) {
    constexpr int D4 = D / 4;
    for (int i = tid; i < L * D4; i += 256) {
        const int row = i / D4, c4 = i % D4;
        float4 v = *reinterpret_cast<const float4*>(q + ((int64_t)b * L + row) * ld + hh * D + 4 * c4);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4*>(qs + row * D + 4 * c4) = v;
    }
    __syncthreads();
    constexpr int NB = D4 * D4, RG = 256 / NB;
    {
        const int rg = tid / NB, ab = (tid % NB) / D4, cb = tid % D4;
        float acc[4][4] = {};
        float sq[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = rg; i < L; i += RG) {
            const float4 x = *reinterpret_cast<const float4*>(qs + i * D + 4 * ab), y = *reinterpret_cast<const float4*>(qs + i * D + 4 * cb);
            const float xa[4] = {x.x, x.y, x.z, x.w}, ya[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sq[u] += xa[u];
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[u][w] = fmaf(xa[u], ya[w], acc[u][w]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<float4*>(red + rg * D * D + (4 * ab + u) * D + 4 * cb) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
            if (cb == 0) red[RG * D * D + rg * D + 4 * ab + u] = sq[u];
        }
    }
    __syncthreads();
    if (tid < D * D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) t += red[r * D * D + tid];
        Qm[tid] = t;
        if (tid < D) {
            float u = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) u += red[RG * D * D + r * D + tid];
            Sq[tid] = u;
        }
    }
    __syncthreads();
}
__host__ __device__ constexpr int mx_moments_floats(int L, int d) { return L * d + d * d + d + (256 / ((d / 4) * (d / 4))) * (d * d + d); }

template <int D>
__global__ __launch_bounds__(256) void mha_bn_dk_fix_kernel(const float* __restrict__ q, const float* __restrict__ k, int64_t ld, int L, int h,
                                                            float scale, const float* __restrict__ corr_a, const float* __restrict__ corr_b,
                                                            float* __restrict__ dk, int64_t ldd, const float* __restrict__ moments,
                                                            float* __restrict__ dk_img, int64_t ldi, int img, const MxImg im) {
    // dk_img != NULL: the repaired rows leave as dk's part of the [dq | dk | dv] gradient image (row stride ldi, plane stride img, both in
    // 16-bit elements; mx_store_grad4) instead of in place -- dk itself is then the key / value sweep's plain fp32 scratch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float vmax = 0.f;
    const int tid = threadIdx.x;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    constexpr int D4 = D / 4;
    float *Qm, *Sq;
    if (moments) {                                              // the forward's moments (of the unscaled q): Qm scale^2, Sq scale
        Qm = reinterpret_cast<float*>(smem);
        Sq = Qm + D * D;
        for (int i = tid; i < D * D + D; i += 256) Qm[i] = moments[(int64_t)lid * (D * D + D) + i] * (i < D * D ? scale * scale : scale);
        __syncthreads();
    } else {
        float* qs = reinterpret_cast<float*>(smem);
        Qm = qs + (size_t)L * D;
        Sq = Qm + D * D;
        mx_q_moments<D>(q, ld, b, L, hh, scale, qs, Qm, Sq, Sq + D, tid);
    }
    // dk rows: a thread keeps ITS four rows of Qm (c4 = tid % D4 for every row it visits: 256 is a multiple of D4) in registers
    const int c4 = tid % D4;
    float qm[4][D], sq4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        sq4[u] = Sq[4 * c4 + u];
#pragma unroll
        for (int e = 0; e < D4; ++e) {
            const float4 t = *reinterpret_cast<const float4*>(Qm + (4 * c4 + u) * D + 4 * e);
            qm[u][4 * e] = t.x; qm[u][4 * e + 1] = t.y; qm[u][4 * e + 2] = t.z; qm[u][4 * e + 3] = t.w;
        }
    }
    for (int i = tid; i < L * D4; i += 256) {
        const int row = i / D4;
        const float* kp = k + ((int64_t)b * L + row) * ld + hh * D;
        float kv[D];
#pragma unroll
        for (int e = 0; e < D4; ++e) {
            const float4 t = *reinterpret_cast<const float4*>(kp + 4 * e);
            kv[4 * e] = t.x; kv[4 * e + 1] = t.y; kv[4 * e + 2] = t.z; kv[4 * e + 3] = t.w;
        }
        const float ca = corr_a[row], cb = corr_b[row];
        float* dp = dk + ((int64_t)b * L + row) * ldd + hh * D + 4 * c4;
        float4 g = *reinterpret_cast<float4*>(dp);
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < D; ++e) t = fmaf(qm[u][e], kv[e], t);
            r[u] = fmaf(cb, t, ca * sq4[u]);
        }
        g.x -= r[0]; g.y -= r[1]; g.z -= r[2]; g.w -= r[3];
        if (dk_img) mx_store_grad4(dk_img, ((int64_t)b * L + row) * ldi + hh * D + 4 * c4, img, g.x, g.y, g.z, g.w, im, vmax);
        else *reinterpret_cast<float4*>(dp) = g;
    }
    if (dk_img) of_amax_commit(im.gamax, vmax);
}

static inline size_t mx_bwd_dq_lds(int LP, int D) { return (size_t)2 * mx_rowplanes_bytes(LP, D) + mx_tplanes_bytes(LP) + 4 * LP * 4; }
static inline size_t mx_bwd_dkv_lds(int LP, int D) { return (size_t)2 * mx_rowplanes_bytes(LP, D) + 2 * mx_tplanes_bytes(LP) + 2 * LP * 4 + 16; }
static inline size_t mx_fwd_lds(int LP, int D) { return (size_t)mx_rowplanes_bytes(LP, D) + mx_tplanes_bytes(LP) + 2 * LP * 4; }

template <typename KernT>
static int mx_reserve(KernT kern, size_t bytes, const char* what) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %zu bytes of LDS", what, bytes);
        return LPM_ERR_LAUNCH;
    }
    return LPM_OK;
}

}  // namespace lpm

#define LPM_MX_CHECK(name)                                                                                        \
    LPM_REQUIRE(B > 0 && L > 0 && h > 0 && (d == 8 || d == 16) && L <= 512, LPM_ERR_UNSUPPORTED_SHAPE,              \
                name ": need d in {8,16} and L <= 512 (L=%d d=%d)", L, d);                                          \
    LPM_REQUIRE(ld >= (int64_t)h * d && ld % 4 == 0, LPM_ERR_BADARG, name ": bad leading dimension")

// Process-wide arithmetic of the backward kernels: 3 = split-bf16 three-term products throughout (the default), 2 = the products behind dS
// on two fp16 terms (round 6: built, measured, NOT the default -- 6 % of the backward at L = 256, 2 % at L = 300 with logits_bn, 0.3 % of a
// cfg-2 step, for gradients 2-3e-4 from the three-term form's: see the kernels' header).  LPM_MHA_BWD_TERMS=2 opts in;
// lpm_mha_bwd_set_terms switches at run time (tests, A/B) and returns the old value.
static std::atomic<int> g_mha_bwd_terms{[] {
    const char* e = getenv("LPM_MHA_BWD_TERMS");
    return (e && e[0] == '2') ? 2 : 3;
}()};
extern "C" int lpm_mha_bwd_set_terms(int terms) {
    if (terms != 2 && terms != 3) return g_mha_bwd_terms.load();
    return g_mha_bwd_terms.exchange(terms);
}

static lpm::MxImg mx_img(const LpmOperandFormat* o_fmt, const LpmOperandFormat* g_fmt) {
    const lpm::OperandFmt fo = lpm::operand_fmt(o_fmt), fg = lpm::operand_fmt(g_fmt);
    return lpm::MxImg{fo.f16, fo.planes, fo.scale, 1.f / fo.scale, fo.amax, fg.f16, fg.planes, fg.scale, fg.amax, 0, 0};
}
static int mx_fwd_launch(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d, float scale,
                         const float* key_scale, const float* key_shift, float* o, int64_t ldo, float* lse, int oimg,
                         lpm_stream_t stream, const char* what, const LpmOperandFormat* o_fmt = nullptr) {
    using namespace lpm;
    const MxImg im = mx_img(o_fmt, nullptr);
    hipStream_t s = (hipStream_t)stream;
    const int nkt = (L + 15) / 16;
    dim3 grid(B * h);
#define LPM_MX_FWD3(N, AFF, DD, RG)                                                                              \
    do {                                                                                                         \
        auto kern = mha_fwd_x3_kernel<N, AFF, DD, RG>;                                                           \
        const size_t lds = mx_fwd_lds(N * 16, DD);                                                               \
        if (int rc = mx_reserve(kern, lds, what)) return rc;                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q, k, v, ld, L, h, scale, key_scale, key_shift, o, ldo, lse, oimg, im); \
    } while (0)
#define LPM_MX_FWD1(N, AFF, RG)        \
    do {                               \
        if (d == 16) LPM_MX_FWD3(N, AFF, 16, RG); \
        else LPM_MX_FWD3(N, AFF, 8, RG); \
    } while (0)
#define LPM_MX_FWD(N)                  \
    do {                               \
        if (key_scale) LPM_MX_FWD1(N, true, true); \
        else if (L == N * 16) LPM_MX_FWD1(N, false, false); \
        else LPM_MX_FWD1(N, false, true); \
    } while (0)
    if (nkt <= 4) LPM_MX_FWD(4);
    else if (nkt <= 8) LPM_MX_FWD(8);
    else if (nkt <= 16) LPM_MX_FWD(16);
    else if (nkt <= 20) LPM_MX_FWD(20);
    else LPM_MX_FWD(32);
#undef LPM_MX_FWD
#undef LPM_MX_FWD1
#undef LPM_MX_FWD3
    return check_launch(what);
}

extern "C" int lpm_mha_fwd_x3(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d, float scale,
                              const float* key_scale, const float* key_shift, float* o, int64_t ldo, float* lse,
                              lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o && lse, LPM_ERR_BADARG, "lpm_mha_fwd_x3: null pointer");
    LPM_MX_CHECK("lpm_mha_fwd_x3");
    LPM_REQUIRE((key_scale == nullptr) == (key_shift == nullptr), LPM_ERR_BADARG, "lpm_mha_fwd_x3: key_scale/key_shift go together");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0, LPM_ERR_BADARG, "lpm_mha_fwd_x3: bad ldo");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, LPM_ERR_BADARG,
                "lpm_mha_fwd_x3: pointers must be 16-byte aligned");
    return mx_fwd_launch(q, k, v, ld, B, L, h, d, scale, key_scale, key_shift, o, ldo, lse, 0, stream, "lpm_mha_fwd_x3");
}

extern "C" int lpm_mha_fwd_x3_image(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d,
                                    float scale, void* o3, float* lse, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o3 && lse, LPM_ERR_BADARG, "lpm_mha_fwd_x3_image: null pointer");
    LPM_MX_CHECK("lpm_mha_fwd_x3_image");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o3) & 15) == 0, LPM_ERR_BADARG,
                "lpm_mha_fwd_x3_image: pointers must be 16-byte aligned");
    return mx_fwd_launch(q, k, v, ld, B, L, h, d, scale, nullptr, nullptr, (float*)o3, 0, lse, h * d, stream, "lpm_mha_fwd_x3_image");
}
extern "C" int lpm_mha_fwd_x3_image_fmt(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d,
                                        float scale, void* o3, float* lse, const LpmOperandFormat* o_fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o3 && lse, LPM_ERR_BADARG, "lpm_mha_fwd_x3_image: null pointer");
    LPM_MX_CHECK("lpm_mha_fwd_x3_image");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o3) & 15) == 0, LPM_ERR_BADARG,
                "lpm_mha_fwd_x3_image: pointers must be 16-byte aligned");
    if (const int rc = operand_fmt_check(o_fmt, "lpm_mha_fwd_x3_image")) return rc;
    return mx_fwd_launch(q, k, v, ld, B, L, h, d, scale, nullptr, nullptr, (float*)o3, 0, lse, h * d, stream, "lpm_mha_fwd_x3_image", o_fmt);
}

static int mx_bwd_launch(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout, int64_t ldo,
                         const float* lse, int B, int L, int h, int d, float scale, const float* key_scale, const float* key_shift,
                         float* dq, float* dk, float* dv, int64_t ldd, const float* corr_a, const float* corr_b, float* dz_partial,
                         int img, int oimg, lpm_stream_t stream, const char* what, const LpmOperandFormat* o_fmt = nullptr,
                         const LpmOperandFormat* g_fmt = nullptr, int64_t dk_plain_ld = 0) {
    using namespace lpm;
    MxImg im = mx_img(o_fmt, g_fmt);
    if (dk_plain_ld) {                                      // dk: plain fp32 rows (dk_plain_ld floats apart) beside an image-form dv
        im.dk_plain = 1;
        im.dk_ld = dk_plain_ld;
    }
    hipStream_t s = (hipStream_t)stream;
    const int nkt = (L + 15) / 16;
    dim3 grid(B * h);
    // two fp16 terms: only where dS is linear in dO (no correction vectors: see the kernels' header)
    const bool h16 = g_mha_bwd_terms.load(std::memory_order_relaxed) == 2 && !corr_a;
    static const int tpw = [] { const char* e = getenv("LPM_MHA_BWD_TPW"); return (e && e[0] == '2') ? 2 : 1; }();   // tiles per wave (A/B)
#define LPM_MX_BWDH_Q(N, AFF, DD, RG, TP)                                                                              \
    do {                                                                                                               \
        auto kq = mha_bwd_dq_h_kernel<N, AFF, DD, RG, TP>;                                                             \
        const size_t lq = mx_bwd_dq_lds(N * 16, DD);                                                                   \
        if (int rc = mx_reserve(kq, lq, what)) return rc;                                                              \
        hipLaunchKernelGGL(kq, grid, dim3(mx_h_nt<N, TP, false>()), lq, s, q, k, v, ld, o, dout, ldo, lse, L, h, scale, key_scale, key_shift, \
                           dq, ldd, corr_a, corr_b, img, oimg, im);                                                    \
    } while (0)
#define LPM_MX_BWDH_KV(N, AFF, DD, TP)                                                                                 \
    do {                                                                                                               \
        auto kk = mha_bwd_dkv_h_kernel<N, AFF, DD, false, TP>;                                                         \
        const size_t lk = mx_bwd_dkv_lds(N * 16, DD);                                                                  \
        if (int rc = mx_reserve(kk, lk, what)) return rc;                                                              \
        hipLaunchKernelGGL(kk, grid, dim3(mx_h_nt<N, TP, true>()), lk, s, q, k, v, ld, o, dout, ldo, lse, L, h, scale, key_scale, key_shift, \
                           dk, dv, ldd, corr_a, corr_b, dz_partial, img, oimg, im);                                    \
    } while (0)
    if (h16 && tpw == 2 && L == 256 && d == 16 && !key_scale) {         // the measured two-tiles-per-wave form (not the default)
        if (dq) LPM_MX_BWDH_Q(16, false, 16, false, 2);
        if (dk || dz_partial) LPM_MX_BWDH_KV(16, false, 16, 2);
        return check_launch(what);
    }
#define LPM_MX_BWD3(N, AFF, DD, RG)                                                                                    \
    do {                                                                                                               \
        if (h16) {                                                                                                     \
            if (dq) LPM_MX_BWDH_Q(N, AFF, DD, RG, 1);                                                                  \
            if (dk || dz_partial) LPM_MX_BWDH_KV(N, AFF, DD, 1);                                                       \
            break;                                                                                                     \
        }                                                                                                              \
        if (dq) {                                                                                                      \
            auto kq = mha_bwd_dq_x3_kernel<N, AFF, DD, RG>;                                                            \
            const size_t lq = mx_bwd_dq_lds(N * 16, DD);                                                               \
            if (int rc = mx_reserve(kq, lq, what)) return rc;                                                          \
            hipLaunchKernelGGL(kq, grid, dim3(MX_DQ_NT), lq, s, q, k, v, ld, o, dout, ldo, lse, L, h, scale, key_scale, key_shift, dq, \
                               ldd, corr_a, corr_b, img, oimg, im);                                                    \
        }                                                                                                              \
        if (dk || dz_partial) {                                                                                        \
            auto kk = (AFF && !corr_a) ? mha_bwd_dkv_x3_kernel<N, AFF, DD, false> : mha_bwd_dkv_x3_kernel<N, AFF, DD, true>;          \
            const size_t lk = mx_bwd_dkv_lds(N * 16, DD);                                                              \
            if (int rc = mx_reserve(kk, lk, what)) return rc;                                                          \
            hipLaunchKernelGGL(kk, grid, dim3(mx_dkv_nt(N)), lk, s, q, k, v, ld, o, dout, ldo, lse, L, h, scale, key_scale, key_shift, dk, \
                               dv, ldd, corr_a, corr_b, dz_partial, img, oimg, im);                                    \
        }                                                                                                              \
    } while (0)
#define LPM_MX_BWD1(N, AFF, RG)        \
    do {                               \
        if (d == 16) LPM_MX_BWD3(N, AFF, 16, RG); \
        else LPM_MX_BWD3(N, AFF, 8, RG); \
    } while (0)
#define LPM_MX_BWD(N)                  \
    do {                               \
        if (key_scale) LPM_MX_BWD1(N, true, true); \
        else if (L == N * 16) LPM_MX_BWD1(N, false, false); \
        else LPM_MX_BWD1(N, false, true); \
    } while (0)
    if (nkt <= 4) LPM_MX_BWD(4);
    else if (nkt <= 8) LPM_MX_BWD(8);
    else if (nkt <= 16) LPM_MX_BWD(16);
    else if (nkt <= 20) LPM_MX_BWD(20);
    else LPM_MX_BWD(32);
#undef LPM_MX_BWD
#undef LPM_MX_BWD1
#undef LPM_MX_BWD3
#undef LPM_MX_BWDH_Q
#undef LPM_MX_BWDH_KV
    return check_launch(what);
}

extern "C" int lpm_mha_bwd_x3(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout,
                              int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, const float* key_scale,
                              const float* key_shift, float* dq, float* dk, float* dv, int64_t ldd, const float* corr_a,
                              const float* corr_b, float* dz_partial, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o && dout && lse, LPM_ERR_BADARG, "lpm_mha_bwd_x3: null pointer");
    LPM_REQUIRE((dk != nullptr) == (dv != nullptr) && (dq || dk || dz_partial), LPM_ERR_BADARG,
                "lpm_mha_bwd_x3: dk and dv go together; give dq, (dk, dv) or both -- or none of them with dz_partial (statistics only)");
    LPM_REQUIRE((corr_a == nullptr) == (corr_b == nullptr), LPM_ERR_BADARG, "lpm_mha_bwd_x3: corr_a/corr_b go together");
    LPM_MX_CHECK("lpm_mha_bwd_x3");
    LPM_REQUIRE((key_scale == nullptr) == (key_shift == nullptr), LPM_ERR_BADARG, "lpm_mha_bwd_x3: key_scale/key_shift go together");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0 && ldd >= (int64_t)h * d && ldd % 4 == 0, LPM_ERR_BADARG, "lpm_mha_bwd_x3: bad ldo/ldd");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                LPM_ERR_BADARG, "lpm_mha_bwd_x3: pointers must be 16-byte aligned");
    return mx_bwd_launch(q, k, v, ld, o, dout, ldo, lse, B, L, h, d, scale, key_scale, key_shift, dq, dk, dv, ldd, corr_a, corr_b,
                         dz_partial, 0, 0, stream, "lpm_mha_bwd_x3");
}

static int mx_dk_correct_launch(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float scale, const float* corr_a,
                                const float* corr_b, float* dk, int64_t ldd, const float* moments, float* dk_img, int64_t ldi, int img,
                                const LpmOperandFormat* g_fmt, lpm_stream_t stream, const char* what) {
    using namespace lpm;
    if (!((q || moments) && k && corr_a && corr_b && dk)) { set_error("%s: null pointer", what); return LPM_ERR_BADARG; }
    if (!(B > 0 && L > 0 && h > 0 && (d == 8 || d == 16) && ld >= (int64_t)h * d && ld % 4 == 0 && ldd >= (int64_t)h * d && ldd % 4 == 0)) {
        set_error("%s: need d in {8, 16} and row strides >= h * d, multiples of 4 (d=%d)", what, d);
        return LPM_ERR_UNSUPPORTED_SHAPE;
    }
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)dk | (uintptr_t)dk_img) & 15) != 0) { set_error("%s: pointers must be 16-byte aligned", what); return LPM_ERR_BADARG; }
    const size_t lds = (moments ? (size_t)(d * d + d) : (size_t)mx_moments_floats(L, d)) * sizeof(float);
    if (lds > 160 * 1024) { set_error("%s: L = %d does not fit LDS", what, L); return LPM_ERR_UNSUPPORTED_SHAPE; }
    const MxImg im = mx_img(nullptr, g_fmt);
    hipStream_t s = (hipStream_t)stream;
    if (d == 16) {
        auto kern = mha_bn_dk_fix_kernel<16>;
        if (int rc = mx_reserve(kern, lds, what)) return rc;
        hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), lds, s, q, k, ld, L, h, scale, corr_a, corr_b, dk, ldd, moments, dk_img, ldi, img, im);
    } else {
        auto kern = mha_bn_dk_fix_kernel<8>;
        if (int rc = mx_reserve(kern, lds, what)) return rc;
        hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), lds, s, q, k, ld, L, h, scale, corr_a, corr_b, dk, ldd, moments, dk_img, ldi, img, im);
    }
    return check_launch(what);
}
extern "C" int lpm_mha_bn_dk_correct(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float scale, const float* corr_a,
                                     const float* corr_b, float* dk, int64_t ldd, const float* moments, lpm_stream_t stream) {
    return mx_dk_correct_launch(q, k, ld, B, L, h, d, scale, corr_a, corr_b, dk, ldd, moments, nullptr, 0, 0, nullptr, stream, "lpm_mha_bn_dk_correct");
}

// logits_bn's one-pass backward with the q / k / v gradients leaving as the [dq | dk | dv] GRADIENT IMAGE the q/k/v layer's GEMMs read
// (round 6: cfg-3 paid a 295 MB -> 295 MB lpm_split_rows pass per step for it).  Three launches write one image, each its own columns:
//   lpm_mha_bwd_x3_bn_image_fmt(dk_plain != NULL, corr NULL, dz_partial): the key / value sweep -- dv into the image, dk WITHOUT the batch
//       statistics' share into dk_plain [B*L, h*d] fp32, the statistics into dz_partial;
//   lpm_mha_bwd_x3_bn_image_fmt(dk_plain NULL, corr_a, corr_b): the query sweep -- dq into the image;
//   lpm_mha_bn_dk_correct_image: dk_plain minus the share -> dk's columns of the image.
// g_fmt: the image's operand format (NULL: split-bf16 [hi | hi | lo] planes; fp16 two-product: [hi | lo]); o, dout plain fp32.
extern "C" int lpm_mha_bwd_x3_bn_image_fmt(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout,
                                           int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, const float* key_scale,
                                           const float* key_shift, float* dk_plain, const float* corr_a, const float* corr_b, float* dz_partial,
                                           void* dqkv_img, const LpmOperandFormat* g_fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o && dout && lse && dqkv_img && key_scale && key_shift, LPM_ERR_BADARG, "lpm_mha_bwd_x3_bn_image: null pointer");
    LPM_REQUIRE((corr_a == nullptr) == (corr_b == nullptr), LPM_ERR_BADARG, "lpm_mha_bwd_x3_bn_image: corr_a/corr_b go together");
    LPM_REQUIRE((dk_plain != nullptr) != (corr_a != nullptr), LPM_ERR_BADARG,
                "lpm_mha_bwd_x3_bn_image: either the key / value sweep (dk_plain, no corrections) or the query sweep (corrections, no dk_plain)");
    LPM_MX_CHECK("lpm_mha_bwd_x3_bn_image");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0, LPM_ERR_BADARG, "lpm_mha_bwd_x3_bn_image: bad ldo");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dqkv_img | (uintptr_t)dk_plain) & 15) == 0,
                LPM_ERR_BADARG, "lpm_mha_bwd_x3_bn_image: pointers must be 16-byte aligned");
    if (const int rc = operand_fmt_check(g_fmt, "lpm_mha_bwd_x3_bn_image")) return rc;
    const int N = h * d;
    const int gpl = g_fmt ? operand_kind_planes(g_fmt->kind) : 3;
    unsigned short* base = (unsigned short*)dqkv_img;
    if (dk_plain)
        return mx_bwd_launch(q, k, v, ld, o, dout, ldo, lse, B, L, h, d, scale, key_scale, key_shift, nullptr, dk_plain, (float*)(base + 2 * N),
                             (int64_t)(3 * gpl) * N, nullptr, nullptr, dz_partial, 3 * N, 0, stream, "lpm_mha_bwd_x3_bn_image", nullptr, g_fmt, N);
    return mx_bwd_launch(q, k, v, ld, o, dout, ldo, lse, B, L, h, d, scale, key_scale, key_shift, (float*)base, nullptr, nullptr,
                         (int64_t)(3 * gpl) * N, corr_a, corr_b, nullptr, 3 * N, 0, stream, "lpm_mha_bwd_x3_bn_image", nullptr, g_fmt);
}
extern "C" int lpm_mha_bn_dk_correct_image(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float scale, const float* corr_a,
                                           const float* corr_b, const float* dk_plain, const float* moments, void* dqkv_img,
                                           const LpmOperandFormat* g_fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dqkv_img, LPM_ERR_BADARG, "lpm_mha_bn_dk_correct_image: null pointer");
    if (const int rc = operand_fmt_check(g_fmt, "lpm_mha_bn_dk_correct_image")) return rc;
    const int N = h * d;
    const int gpl = g_fmt ? operand_kind_planes(g_fmt->kind) : 3;
    return mx_dk_correct_launch(q, k, ld, B, L, h, d, scale, corr_a, corr_b, const_cast<float*>(dk_plain), N, moments,
                                (float*)((unsigned short*)dqkv_img + N), (int64_t)(3 * gpl) * N, 3 * N, g_fmt, stream, "lpm_mha_bn_dk_correct_image");
}

extern "C" int lpm_mha_bwd_x3_image(const float* q, const float* k, const float* v, int64_t ld, const void* o, int o_is_image,
                                    const float* dout, int64_t ldo, const float* lse, int B, int L, int h, int d, float scale,
                                    void* dqkv3, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o && dout && lse && dqkv3, LPM_ERR_BADARG, "lpm_mha_bwd_x3_image: null pointer");
    LPM_MX_CHECK("lpm_mha_bwd_x3_image");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0, LPM_ERR_BADARG, "lpm_mha_bwd_x3_image: bad ldo");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dqkv3) & 15) == 0, LPM_ERR_BADARG,
                "lpm_mha_bwd_x3_image: pointers must be 16-byte aligned");
    const int N = h * d;                                  // image row: [hi(3N) | hi(3N) | lo(3N)] over the columns [dq | dk | dv]
    unsigned short* base = (unsigned short*)dqkv3;
    return mx_bwd_launch(q, k, v, ld, (const float*)o, dout, ldo, lse, B, L, h, d, scale, nullptr, nullptr, (float*)base, (float*)(base + N),
                         (float*)(base + 2 * N), (int64_t)9 * N, nullptr, nullptr, nullptr, 3 * N, o_is_image ? N : 0, stream,
                         "lpm_mha_bwd_x3_image");
}
// ... with the attention result's image (always an image here) and the [dq | dk | dv] gradient image in either operand format
extern "C" int lpm_mha_bwd_x3_image_fmt(const float* q, const float* k, const float* v, int64_t ld, const void* o_img,
                                        const LpmOperandFormat* o_fmt, const float* dout, int64_t ldo, const float* lse, int B, int L, int h,
                                        int d, float scale, void* dqkv_img, const LpmOperandFormat* g_fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o_img && dout && lse && dqkv_img, LPM_ERR_BADARG, "lpm_mha_bwd_x3_image: null pointer");
    LPM_MX_CHECK("lpm_mha_bwd_x3_image");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0, LPM_ERR_BADARG, "lpm_mha_bwd_x3_image: bad ldo");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o_img | (uintptr_t)dout | (uintptr_t)dqkv_img) & 15) == 0, LPM_ERR_BADARG,
                "lpm_mha_bwd_x3_image: pointers must be 16-byte aligned");
    if (const int rc = operand_fmt_check(o_fmt, "lpm_mha_bwd_x3_image")) return rc;
    if (const int rc = operand_fmt_check(g_fmt, "lpm_mha_bwd_x3_image")) return rc;
    const int N = h * d;
    const int gpl = g_fmt ? operand_kind_planes(g_fmt->kind) : 3;      // two planes: row = [hi(3N) | lo(3N)]
    unsigned short* base = (unsigned short*)dqkv_img;
    return mx_bwd_launch(q, k, v, ld, (const float*)o_img, dout, ldo, lse, B, L, h, d, scale, nullptr, nullptr, (float*)base, (float*)(base + N),
                         (float*)(base + 2 * N), (int64_t)(3 * gpl) * N, nullptr, nullptr, nullptr, 3 * N, N, stream,
                         "lpm_mha_bwd_x3_image", o_fmt, g_fmt);
}
