// Residual add + tf.contrib.layers.layer_norm (transformer_utils.py:405-411, 451-454, 712-713).
// TF1 defaults: ONE mean/variance per example over all non-batch axes (L*F = 262144 elements for the video
// encoder), gamma/beta over the last axis, variance epsilon 1e-12.  A per-row LayerNorm kernel gets 80 rows of
// 1 MB each -- 80 workgroups on 256 CUs; here each example is cut into row chunks (grid = B x NB) with a
// partial-sum hand-off between two light HBM-bound passes, and the residual add is fused into pass 1.
//   fwd : z = act(a + bias) (+ r);  y = (z - mean) * rstd * gamma + beta        saves z, stats[b] = (mean, rstd)
//   bwd : g = dy*gamma; dz = rstd * (g - mean_e(g) - zhat * mean_e(g*zhat));  dgamma = sum dy*zhat; dbeta = sum dy
// The producing dense layer's bias add and ReLU (tf.layers.dense(..., use_bias=True[, activation=relu]) at
// transformer_utils.py:583 and :708-711) ride in pass 1 of the forward; in the backward da = dz * [a + bias > 0] and
// dbias = column sums of da come out of the apply pass -- two elementwise passes and a reduction per layer less.
#include "lpm_common.h"
#include "operand_format.h"
#include <atomic>

namespace lpm {

constexpr int LN_NB = 16;   // chunks per example
constexpr int LN_RS = 16;               // slices of the backward column reductions (ln_colreduce_kernel)
constexpr int LN_CR_COUNTERS = 32;      // its per-call arrival counters: (F / 64 <= 16 column blocks) x 2 sets

typedef __bf16 ln_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ln_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ln_bf16_pair(float a, float b) {     // (a, b) -> packed bf16, round to nearest even
    const ln_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ln_bf16x2));
}
__device__ __forceinline__ float ln_bf16_up(unsigned h) { return __uint_as_float(h << 16); }

__device__ __forceinline__ float block_sum(float v, float* sh) {   // 256 threads; result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// pass 1: z = a + r (written when r != NULL), partial[b][chunk] = (sum z, sum z^2)
// MASK (NetVladV2's encoder: tf.layers.dropout between the dense layer and the layer norm, transformer_utils.py:450): the keep mask
// [B, L, F] (one byte per element) and its scale 1 / keep_probability apply to act(a + bias), BEFORE the residual is added.
template <bool MASK>
__global__ __launch_bounds__(256) void ln_fwd_stats_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                           const float* __restrict__ bias, int relu, int F,
                                                           int64_t n_per, float* __restrict__ z,
                                                           float* __restrict__ partial, const float* __restrict__ r_scale,
                                                           const unsigned char* __restrict__ mask, float mscale) {
    __shared__ float sh[4];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int64_t n4 = n_per / 4;
    const float4* ap = reinterpret_cast<const float4*>(a + (int64_t)b * n_per);
    const float4* rp = r ? reinterpret_cast<const float4*>(r + (int64_t)b * n_per) : nullptr;
    float4* zp = reinterpret_cast<float4*>(z + (int64_t)b * n_per);
    float s = 0.f, q = 0.f;
    const int F4 = F / 4;
    // chunk ch takes the 256-float4 pieces ch, ch + LN_NB, ...: at any moment the LN_NB workgroups of an example read
    // LN_NB consecutive 4 KB pieces (contiguous chunks put every workgroup of the grid at the same offset of its own
    // 64 KB-aligned range at the same time: the same few HBM channels for everyone, ~3 TB/s)
    for (int64_t i = (int64_t)ch * 256 + threadIdx.x; i < n4; i += (int64_t)LN_NB * 256) {
        float4 v = ap[i];
        if (bias) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + (int)(i % F4) * 4);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            if (relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
        }
        if (MASK) {
            const uchar4 mk = reinterpret_cast<const uchar4*>(mask + (int64_t)b * n_per)[i];
            v.x = mk.x ? v.x * mscale : 0.f; v.y = mk.y ? v.y * mscale : 0.f; v.z = mk.z ? v.z * mscale : 0.f; v.w = mk.w ? v.w * mscale : 0.f;
        }
        if (rp) {
            float4 w = rp[i];
            if (r_scale) {                     // the residual is a lazily normalised descriptor: one factor per (example, row)
                const float rs = r_scale[(int64_t)b * (n_per / F) + i / F4];
                w.x *= rs; w.y *= rs; w.z *= rs; w.w *= rs;
            }
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (rp || bias || MASK) zp[i] = v;
        s += (v.x + v.y) + (v.z + v.w);
        q = fmaf(v.x, v.x, q); q = fmaf(v.y, v.y, q); q = fmaf(v.z, v.z, q); q = fmaf(v.w, v.w, q);
    }
    s = block_sum(s, sh);
    q = block_sum(q, sh);
    if (threadIdx.x == 0) {
        partial[((int64_t)b * LN_NB + ch) * 2] = s;
        partial[((int64_t)b * LN_NB + ch) * 2 + 1] = q;
    }
}

// pass 2: y = (z - mean) * rstd * gamma + beta.  r2 != NULL (layer-norm pair): what leaves is y + r2 -- the NEXT layer norm's
// residual sum -- together with its per-chunk (sum, sum of squares) in partial_next, so the next layer norm needs no statistics
// pass of its own.
__global__ __launch_bounds__(256) void ln_fwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ partial,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int64_t n_per, int F,
                                                           float eps, float* __restrict__ y, int64_t y_batch,
                                                           float* __restrict__ stats, const float* __restrict__ r2,
                                                           float* __restrict__ partial_next, unsigned short* __restrict__ y3,
                                                           const OperandFmt fmt) {
    // y3 != NULL: y ALSO leaves as the split-bf16 activation image [rows][3 F] = [hi | lo | hi] of the dense layer that reads it next
    // (split_gemm.hip's format): the separate split pass over y (read 4 B, write 6 B per element) is not run
    __shared__ float sh[4];
    const int b = blockIdx.x, ch = blockIdx.y;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < LN_NB; ++i) {
        s += (double)partial[((int64_t)b * LN_NB + i) * 2];
        q += (double)partial[((int64_t)b * LN_NB + i) * 2 + 1];
    }
    const double mu = s / (double)n_per;
    double var = q / (double)n_per - mu * mu;
    if (var < 0.0) var = 0.0;
    const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (ch == 0 && threadIdx.x == 0) {
        stats[2 * b] = mean;
        stats[2 * b + 1] = rstd;
    }
    const int64_t n4 = n_per / 4;
    const int F4 = F / 4;
    const float4* zp = reinterpret_cast<const float4*>(z + (int64_t)b * n_per);
    const float4* rp = r2 ? reinterpret_cast<const float4*>(r2 + (int64_t)b * n_per) : nullptr;
    float4* yp = reinterpret_cast<float4*>(y + (int64_t)b * y_batch);
    float ns = 0.f, nq = 0.f, vmax = 0.f;
    for (int64_t i = (int64_t)ch * 256 + threadIdx.x; i < n4; i += (int64_t)LN_NB * 256) {      // interleaved pieces, as in pass 1
        const int c = (int)(i % F4) * 4;
        const float4 v = zp[i];
        const float4 g = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = fmaf((v.x - mean) * rstd, g.x, bt.x);
        o.y = fmaf((v.y - mean) * rstd, g.y, bt.y);
        o.z = fmaf((v.z - mean) * rstd, g.z, bt.z);
        o.w = fmaf((v.w - mean) * rstd, g.w, bt.w);
        if (rp) {
            const float4 w = rp[i];
            o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w;
            ns += (o.x + o.y) + (o.z + o.w);
            nq = fmaf(o.x, o.x, nq); nq = fmaf(o.y, o.y, nq); nq = fmaf(o.z, o.z, nq); nq = fmaf(o.w, o.w, nq);
        }
        yp[i] = o;
        if (y3 && fmt.f16) {       // the fp16 two-product format (operand_format.h): [hi | lo] planes of y * scale
            vmax = of_amax4(vmax, o.x, o.y, o.z, o.w);
            uint2 hi, lo;
            of_split4(o.x, o.y, o.z, o.w, 1, fmt.scale, hi, lo);
            of_store_row4(y3 + ((int64_t)b * (n_per / F) + i / F4) * fmt.planes * F, F, c, hi, lo, fmt.planes, 0);
        } else if (y3) {
            vmax = of_amax4(vmax, o.x, o.y, o.z, o.w);
            const unsigned hx = ln_bf16_pair(o.x, o.y), hz = ln_bf16_pair(o.z, o.w);
            const unsigned lx = ln_bf16_pair(o.x - ln_bf16_up(hx & 0xffffu), o.y - ln_bf16_up(hx >> 16));
            const unsigned lz = ln_bf16_pair(o.z - ln_bf16_up(hz & 0xffffu), o.w - ln_bf16_up(hz >> 16));
            unsigned short* p = y3 + ((int64_t)b * (n_per / F) + i / F4) * 3 * F + c;
            *reinterpret_cast<uint2*>(p) = make_uint2(hx, hz);
            *reinterpret_cast<uint2*>(p + F) = make_uint2(lx, lz);
            *reinterpret_cast<uint2*>(p + 2 * F) = make_uint2(hx, hz);
        }
    }
    if (rp) {
        ns = block_sum(ns, sh);
        nq = block_sum(nq, sh);
        if (threadIdx.x == 0) {
            partial_next[((int64_t)b * LN_NB + ch) * 2] = ns;
            partial_next[((int64_t)b * LN_NB + ch) * 2 + 1] = nq;
        }
    }
    if (y3) of_amax_commit(fmt.amax, vmax);
}

// backward pass 1: per (example, chunk): sums of g and g*zhat (g = dy*gamma) and the column partials of
// dy*zhat / dy for dgamma / dbeta.  Threads are laid out (row group, float4 column) with F4 = F/4 dividing 256,
// and chunks are sets of whole rows, so a thread always sees the same 4 columns.
__global__ __launch_bounds__(256) void ln_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, int L, int F,
                                                           float* __restrict__ partial, float* __restrict__ colpart,
                                                           int64_t dy_batch, unsigned* __restrict__ counters) {
    __shared__ float sh[4];
    __shared__ float4 cs[2][256];
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    // this call's arrival counters of ln_colreduce_kernel (two launches further down the same stream) start from zero
    if (b == 0 && ch == 0 && tid < LN_CR_COUNTERS) counters[tid] = 0u;
    const int F4 = F / 4, RG = 256 / F4;
    const int c4 = tid % F4, rg = tid / F4;
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + 4 * c4);
    float s1 = 0.f, s2 = 0.f;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = ch * RG + rg; l < L; l += LN_NB * RG) {      // row groups ch, ch + LN_NB, ...: see ln_fwd_stats_kernel
        const int64_t off = ((int64_t)b * L + l) * F + 4 * c4;
        const float4 d = *reinterpret_cast<const float4*>(dy + (int64_t)b * dy_batch + (int64_t)l * F + 4 * c4);
        const float4 v = *reinterpret_cast<const float4*>(z + off);
        const float hx = (v.x - mean) * rstd, hy = (v.y - mean) * rstd, hz = (v.z - mean) * rstd, hw = (v.w - mean) * rstd;
        const float gx = d.x * g4.x, gy = d.y * g4.y, gz = d.z * g4.z, gw = d.w * g4.w;
        s1 += (gx + gy) + (gz + gw);
        s2 = fmaf(gx, hx, s2); s2 = fmaf(gy, hy, s2); s2 = fmaf(gz, hz, s2); s2 = fmaf(gw, hw, s2);
        dg.x = fmaf(d.x, hx, dg.x); dg.y = fmaf(d.y, hy, dg.y); dg.z = fmaf(d.z, hz, dg.z); dg.w = fmaf(d.w, hw, dg.w);
        db.x += d.x; db.y += d.y; db.z += d.z; db.w += d.w;
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (tid == 0) {
        partial[((int64_t)b * LN_NB + ch) * 2] = s1;
        partial[((int64_t)b * LN_NB + ch) * 2 + 1] = s2;
    }
    cs[0][tid] = dg;
    cs[1][tid] = db;
    __syncthreads();
    if (rg == 0) {
        for (int i = 1; i < RG; ++i) {
            const float4 a = cs[0][i * F4 + c4], c = cs[1][i * F4 + c4];
            dg.x += a.x; dg.y += a.y; dg.z += a.z; dg.w += a.w;
            db.x += c.x; db.y += c.y; db.z += c.z; db.w += c.w;
        }
        float* cp = colpart + ((int64_t)b * LN_NB + ch) * 2 * F;
        *reinterpret_cast<float4*>(cp + 4 * c4) = dg;
        *reinterpret_cast<float4*>(cp + F + 4 * c4) = db;
    }
}

// backward pass 2: dz = rstd * (g - S1/N - zhat * S2/N).  With a fused bias (act_a != NULL): da = dz * [act_a + bias > 0]
// when relu (written to `da`), da = dz otherwise, and the per-(example, chunk) column partials of da -> biaspart for dbias.
// dr_extra != NULL: the residual's gradient leaves as dz + dr_extra (another consumer's gradient of the same tensor, e.g. the
// dy of this very layer norm when its residual also feeds the next one) -- the add rides on the store; da is then written
// separately even without a ReLU.
// Threads are laid out (row group, float4 column) with F4 dividing 256 and chunks of whole rows, as in pass 1.
// MASK: da = dz * keep mask * scale (after the ReLU mask, if any): the gradient of the dense layer's raw output through the dropout.
template <bool MASK>
__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ partial, int L, int F,
                                                           float* __restrict__ dz, const float* __restrict__ act_a,
                                                           const float* __restrict__ bias, int relu, float* __restrict__ da,
                                                           float* __restrict__ biaspart, const float* __restrict__ dr_extra,
                                                           int64_t dy_batch, unsigned short* __restrict__ da_img,
                                                           const unsigned char* __restrict__ mask, float mscale, const OperandFmt fmt) {
    __shared__ float4 cs[256];
    float vmax = 0.f;
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const int64_t n_per = (int64_t)L * F;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < LN_NB; ++i) {
        s1 += (double)partial[((int64_t)b * LN_NB + i) * 2];
        s2 += (double)partial[((int64_t)b * LN_NB + i) * 2 + 1];
    }
    const float m1 = (float)(s1 / (double)n_per), m2 = (float)(s2 / (double)n_per);
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const int F4 = F / 4, RG = 256 / F4;
    const int c4 = tid % F4, rg = tid / F4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c4);
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bb = *reinterpret_cast<const float4*>(bias + 4 * c4);
    for (int l = ch * RG + rg; l < L; l += LN_NB * RG) {      // row groups ch, ch + LN_NB, ...: see ln_fwd_stats_kernel
        const int64_t off = ((int64_t)b * L + l) * F + 4 * c4;
        const float4 d = *reinterpret_cast<const float4*>(dy + (int64_t)b * dy_batch + (int64_t)l * F + 4 * c4);
        const float4 v = *reinterpret_cast<const float4*>(z + off);
        float4 o;
        o.x = rstd * (d.x * g.x - m1 - (v.x - mean) * rstd * m2);
        o.y = rstd * (d.y * g.y - m1 - (v.y - mean) * rstd * m2);
        o.z = rstd * (d.z * g.z - m1 - (v.z - mean) * rstd * m2);
        o.w = rstd * (d.w * g.w - m1 - (v.w - mean) * rstd * m2);
        if (dr_extra) {
            const float4 e = *reinterpret_cast<const float4*>(dr_extra + off);
            *reinterpret_cast<float4*>(dz + off) = make_float4(o.x + e.x, o.y + e.y, o.z + e.z, o.w + e.w);
        } else {
            *reinterpret_cast<float4*>(dz + off) = o;
        }
        if (relu) {
            const float4 av = *reinterpret_cast<const float4*>(act_a + off);
            o.x = (av.x + bb.x > 0.f) ? o.x : 0.f;
            o.y = (av.y + bb.y > 0.f) ? o.y : 0.f;
            o.z = (av.z + bb.z > 0.f) ? o.z : 0.f;
            o.w = (av.w + bb.w > 0.f) ? o.w : 0.f;
        }
        if (MASK) {
            const uchar4 mk = *reinterpret_cast<const uchar4*>(mask + off);
            o.x = mk.x ? o.x * mscale : 0.f; o.y = mk.y ? o.y * mscale : 0.f; o.z = mk.z ? o.z * mscale : 0.f; o.w = mk.w ? o.w * mscale : 0.f;
        }
        if (da && (relu || dr_extra || MASK)) *reinterpret_cast<float4*>(da + off) = o;
        if (da_img && fmt.f16) {
            vmax = of_amax4(vmax, o.x, o.y, o.z, o.w);
            uint2 hi, lo;
            of_split4(o.x, o.y, o.z, o.w, 1, fmt.scale, hi, lo);
            of_store_row4(da_img + ((int64_t)b * L + l) * fmt.planes * F, F, 4 * c4, hi, lo, fmt.planes, 1);
        } else if (da_img) {      // da only feeds GEMMs: it leaves as their split-bf16 gradient image, row = [hi | hi | lo] planes of F
            vmax = of_amax4(vmax, o.x, o.y, o.z, o.w);
            unsigned short* p = da_img + ((int64_t)b * L + l) * 3 * F + 4 * c4;
            const uint2 hi = make_uint2(ln_bf16_pair(o.x, o.y), ln_bf16_pair(o.z, o.w));
            const uint2 lo = make_uint2(ln_bf16_pair(o.x - ln_bf16_up(hi.x & 0xffffu), o.y - ln_bf16_up(hi.x >> 16)),
                                        ln_bf16_pair(o.z - ln_bf16_up(hi.y & 0xffffu), o.w - ln_bf16_up(hi.y >> 16)));
            *reinterpret_cast<uint2*>(p) = hi;
            *reinterpret_cast<uint2*>(p + F) = hi;
            *reinterpret_cast<uint2*>(p + 2 * F) = lo;
        }
        if (bias) { acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    }
    if (bias) {
        cs[tid] = acc;
        __syncthreads();
        if (rg == 0) {
            for (int i = 1; i < RG; ++i) {
                const float4 a = cs[i * F4 + c4];
                acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
            }
            *reinterpret_cast<float4*>(biaspart + ((int64_t)b * LN_NB + ch) * F + 4 * c4) = acc;
        }
    }
    if (da_img) of_amax_commit(fmt.amax, vmax);
}

// Column reductions of the per-(example, chunk) partials (1280 rows x 2 x F at cfg-2: 10 MB + 5 MB with a bias) in ONE launch
// behind the apply pass (they are not on its dependency chain).  grid (ceil(F / 64), LN_RS, 1 or 2): set z = 0 sums the two arrays
// of colpart [nblk][2][F] -> dgamma, dbeta; set z = 1 (fused bias) sums biaspart [nblk][1][F] -> dbias.  Slice y sums rows y,
// y + LN_RS, ... into tmp[z][y][.][F] (write-through stores), counts itself in on the (column block, set) counter of this launch's
// slot, and the workgroup that arrives last adds the LN_RS slices in fp64 IN SLICE ORDER (L2-bypassing loads): which workgroup is
// last does not change a bit of the result.  (Before: two stages x two launches of ~5 us per reduction, four launches per
// layer-norm backward with a bias; a single stage of 16-column workgroups read its 64-byte row pieces at 0.9 TB/s: 17 us.)
// Counters: per CALL, in the caller's workspace behind tmp ((F / 64 <= 16 column blocks) x 2 sets), zeroed by the call's first kernel
// (ln_bwd_stats_kernel, stream order) -- no persistent device state: an aborted launch or any number of launches in flight on any
// streams cannot leave a counter shared or non-zero.  The arrival is an acq_rel agent-scope fetch_add: the release orders this
// workgroup's slice stores before its ticket, the acquire orders the last arriver's slice loads behind it.

__global__ __launch_bounds__(256) void ln_colreduce_kernel(const float* __restrict__ colpart, const float* __restrict__ biaspart, int nblk,
                                                           int F, float* __restrict__ tmp, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dbias,
                                                           unsigned* __restrict__ counters) {
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) unsigned gu32;
    __shared__ float sh[4][2][64];
    __shared__ int last;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, y = blockIdx.y;
    const bool second = blockIdx.z == 1;
    const int narr = second ? 1 : 2;
    const float* part = second ? biaspart : colpart;
    float* tz = tmp + (int64_t)blockIdx.z * LN_RS * 2 * F;
    float acc0 = 0.f, acc1 = 0.f;
    if (c < F) {
        for (int b = y + LN_RS * rg; b < nblk; b += LN_RS * 4 * 8) {      // eight rows per round: independent loads, fixed order of additions
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = b + LN_RS * 4 * u;
                const float* p = part + (int64_t)min(bb, nblk - 1) * narr * F + c;
                v0[u] = (bb < nblk) ? p[0] : 0.f;
                v1[u] = (bb < nblk && !second) ? p[F] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 += v0[u];
                acc1 += v1[u];
            }
        }
    }
    sh[rg][0][cl] = acc0;
    sh[rg][1][cl] = acc1;
    __syncthreads();
    if (rg == 0 && c < F) {
        for (int a = 0; a < narr; ++a)
            __hip_atomic_store((gfloat*)(tz + ((int64_t)y * 2 + a) * F + c), (sh[0][a][cl] + sh[1][a][cl]) + (sh[2][a][cl] + sh[3][a][cl]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the storing wave drains its write-through stores
    __syncthreads();
    gu32* cnt = (gu32*)(counters + blockIdx.z * 16 + blockIdx.x);
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(LN_RS - 1);
    __syncthreads();
    if (!last) return;
    if (rg == 0 && c < F) {
        float* outs[2] = {second ? dbias : dgamma, dbeta};
        for (int a = 0; a < narr; ++a) {
            float t[LN_RS];
#pragma unroll
            for (int i = 0; i < LN_RS; ++i)
                t[i] = __hip_atomic_load((gfloat*)(tz + ((int64_t)i * 2 + a) * F + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < LN_RS; ++i) s += (double)t[i];
            outs[a][c] = (float)s;
        }
    }
}

}  // namespace lpm

extern "C" size_t lpm_layer_norm_workspace_bytes(int B, int F) {
    return ((size_t)B * lpm::LN_NB * 2 + (size_t)B * lpm::LN_NB * 3 * F + (size_t)lpm::LN_RS * 4 * F + lpm::LN_CR_COUNTERS) * sizeof(float);
}

#define LPM_LN_CHECK(name)                                                                                              \
    LPM_REQUIRE(B > 0 && L > 0 && (F == 128 || F == 256 || F == 512 || F == 1024), LPM_ERR_UNSUPPORTED_SHAPE,             \
                name ": need F in {128,256,512,1024} (F=%d)", F);                                                        \
    LPM_REQUIRE(workspace && workspace_bytes >= lpm_layer_norm_workspace_bytes(B, F), LPM_ERR_WORKSPACE, name ": workspace too small")

static int layer_norm_act_fwd_impl(const float* a, const float* bias, int relu, const float* r, const float* r_scale, const float* gamma,
                                   const float* beta, int B, int L, int F, float eps, float* y, int64_t y_batch_stride, float* z,
                                   float* stats, void* workspace, size_t workspace_bytes, lpm_stream_t stream, void* y3 = nullptr,
                                   const unsigned char* mask = nullptr, float mask_scale = 1.f, const LpmOperandFormat* fmt = nullptr) {
    using namespace lpm;
    if (const int rc = operand_fmt_check(fmt, "lpm_layer_norm_act_fwd")) return rc;
    LPM_REQUIRE(a && gamma && beta && y && stats && (z || (!r && !bias && !mask)), LPM_ERR_BADARG,
                "lpm_layer_norm_act_fwd: null pointer (z is required with a residual, a bias or a mask)");
    LPM_REQUIRE(!mask || ((uintptr_t)mask & 3) == 0, LPM_ERR_BADARG, "lpm_layer_norm_act_fwd: the keep mask must be 4-byte aligned");
    LPM_REQUIRE(bias || !relu, LPM_ERR_BADARG, "lpm_layer_norm_act_fwd: relu needs the bias it follows");
    LPM_REQUIRE(!r_scale || r, LPM_ERR_BADARG, "lpm_layer_norm_act_fwd: a residual row scale needs the residual");
    LPM_LN_CHECK("lpm_layer_norm_act_fwd");
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const int64_t n_per = (int64_t)L * F;
    const int64_t yb = y_batch_stride ? y_batch_stride : n_per;
    LPM_REQUIRE(yb >= n_per && yb % 4 == 0 && ((uintptr_t)y & 15) == 0, LPM_ERR_BADARG,
                "lpm_layer_norm_act_fwd: y_batch_stride must be >= L*F and a multiple of 4, y 16-byte aligned");
    dim3 grid(B, LN_NB);
    if (mask)
        hipLaunchKernelGGL(ln_fwd_stats_kernel<true>, grid, dim3(256), 0, s, a, r, bias, relu, F, n_per, z, partial, r_scale, mask, mask_scale);
    else
        hipLaunchKernelGGL(ln_fwd_stats_kernel<false>, grid, dim3(256), 0, s, a, r, bias, relu, F, n_per, z, partial, r_scale,
                           (const unsigned char*)nullptr, 1.f);
    hipLaunchKernelGGL(ln_fwd_apply_kernel, grid, dim3(256), 0, s, (r || bias || mask) ? z : a, partial, gamma, beta, n_per, F, eps, y, yb, stats,
                       (const float*)nullptr, (float*)nullptr, (unsigned short*)y3, operand_fmt(fmt));
    return check_launch("lpm_layer_norm_act_fwd");
}
// ... with tf.layers.dropout between the dense layer and the layer norm (NetVladV2's TransformerEncoderMod, transformer_utils.py:450-454):
// y = layer_norm(act(a + bias) * keep * mask_scale + r); mask [B, L, F] one byte per element (non-zero = kept), y3 optional
extern "C" int lpm_layer_norm_act_mask_image_fwd(const float* a, const float* bias, int relu, const unsigned char* mask, float mask_scale,
                                                 const float* r, const float* gamma, const float* beta, int B, int L, int F, float eps,
                                                 float* y, int64_t y_batch_stride, void* y3, float* z, float* stats, void* workspace,
                                                 size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(mask, LPM_ERR_BADARG, "lpm_layer_norm_act_mask_image_fwd: null keep mask");
    LPM_REQUIRE(!y3 || ((uintptr_t)y3 & 7) == 0, LPM_ERR_BADARG, "lpm_layer_norm_act_mask_image_fwd: y3 not 8-byte aligned");
    return layer_norm_act_fwd_impl(a, bias, relu, r, nullptr, gamma, beta, B, L, F, eps, y, y_batch_stride, z, stats, workspace,
                                   workspace_bytes, stream, y3, mask, mask_scale);
}
extern "C" int lpm_layer_norm_act_mask_image_fwd_fmt(const float* a, const float* bias, int relu, const unsigned char* mask, float mask_scale,
                                                     const float* r, const float* gamma, const float* beta, int B, int L, int F, float eps,
                                                     float* y, int64_t y_batch_stride, void* y3, float* z, float* stats, void* workspace,
                                                     size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(mask, LPM_ERR_BADARG, "lpm_layer_norm_act_mask_image_fwd: null keep mask");
    LPM_REQUIRE(!y3 || ((uintptr_t)y3 & 7) == 0, LPM_ERR_BADARG, "lpm_layer_norm_act_mask_image_fwd: y3 not 8-byte aligned");
    return layer_norm_act_fwd_impl(a, bias, relu, r, nullptr, gamma, beta, B, L, F, eps, y, y_batch_stride, z, stats, workspace,
                                   workspace_bytes, stream, y3, mask, mask_scale, fmt);
}
extern "C" int lpm_layer_norm_act_fwd(const float* a, const float* bias, int relu, const float* r, const float* gamma,
                                      const float* beta, int B, int L, int F, float eps, float* y, int64_t y_batch_stride,
                                      float* z, float* stats, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    return layer_norm_act_fwd_impl(a, bias, relu, r, nullptr, gamma, beta, B, L, F, eps, y, y_batch_stride, z, stats, workspace,
                                   workspace_bytes, stream);
}
// ... with the residual given as raw rows times one factor per (example, row): r_scale [B * L] (the pooled descriptor in its lazily
// normalised form, lpm_vlad_row_scales)
extern "C" int lpm_layer_norm_act_image_fwd(const float* a, const float* bias, int relu, const float* r, const float* r_scale,
                                            const float* gamma, const float* beta, int B, int L, int F, float eps, float* y,
                                            int64_t y_batch_stride, void* y3, float* z, float* stats, void* workspace,
                                            size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(y3 && ((uintptr_t)y3 & 7) == 0, LPM_ERR_BADARG, "lpm_layer_norm_act_image_fwd: y3 missing or not 8-byte aligned");
    return layer_norm_act_fwd_impl(a, bias, relu, r, r_scale, gamma, beta, B, L, F, eps, y, y_batch_stride, z, stats, workspace,
                                   workspace_bytes, stream, y3);
}
extern "C" int lpm_layer_norm_act_image_fwd_fmt(const float* a, const float* bias, int relu, const float* r, const float* r_scale,
                                                const float* gamma, const float* beta, int B, int L, int F, float eps, float* y,
                                                int64_t y_batch_stride, void* y3, float* z, float* stats, void* workspace,
                                                size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(y3 && ((uintptr_t)y3 & 7) == 0, LPM_ERR_BADARG, "lpm_layer_norm_act_image_fwd: y3 missing or not 8-byte aligned");
    return layer_norm_act_fwd_impl(a, bias, relu, r, r_scale, gamma, beta, B, L, F, eps, y, y_batch_stride, z, stats, workspace,
                                   workspace_bytes, stream, y3, nullptr, 1.f, fmt);
}
extern "C" int lpm_layer_norm_act_fwd_rs(const float* a, const float* bias, int relu, const float* r, const float* r_scale,
                                         const float* gamma, const float* beta, int B, int L, int F, float eps, float* y,
                                         int64_t y_batch_stride, float* z, float* stats, void* workspace, size_t workspace_bytes,
                                         lpm_stream_t stream) {
    return layer_norm_act_fwd_impl(a, bias, relu, r, r_scale, gamma, beta, B, L, F, eps, y, y_batch_stride, z, stats, workspace,
                                   workspace_bytes, stream);
}

// layer_norm(layer_norm(act(a + bias) + r; gamma1, beta1) + r; gamma2, beta2): the two layer norms at the end of the V1 encoder
// (transformer_utils.py:712-713 then :409-411), which add the SAME residual.  Three passes instead of four: the first layer
// norm's apply pass adds r again and reduces the second one's statistics on the way out (its output n is never stored).
// Saves for the backward: z1, stats1 (first layer norm), z2, stats2 (second); y = the result, batch stride as above.
extern "C" int lpm_layer_norm_pair_fwd(const float* a, const float* bias, int relu, const float* r, const float* gamma1,
                                       const float* beta1, const float* gamma2, const float* beta2, int B, int L, int F, float eps,
                                       float* y, int64_t y_batch_stride, float* z1, float* stats1, float* z2, float* stats2,
                                       void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(a && r && gamma1 && beta1 && gamma2 && beta2 && y && z1 && stats1 && z2 && stats2, LPM_ERR_BADARG,
                "lpm_layer_norm_pair_fwd: null pointer");
    LPM_REQUIRE(bias || !relu, LPM_ERR_BADARG, "lpm_layer_norm_pair_fwd: relu needs the bias it follows");
    LPM_LN_CHECK("lpm_layer_norm_pair_fwd");
    hipStream_t s = (hipStream_t)stream;
    float* partial1 = (float*)workspace;
    float* partial2 = partial1 + (size_t)B * LN_NB * 2;       // (the backward's column-partial region: free in the forward)
    const int64_t n_per = (int64_t)L * F;
    const int64_t yb = y_batch_stride ? y_batch_stride : n_per;
    LPM_REQUIRE(yb >= n_per && yb % 4 == 0 && ((uintptr_t)y & 15) == 0, LPM_ERR_BADARG,
                "lpm_layer_norm_pair_fwd: y_batch_stride must be >= L*F and a multiple of 4, y 16-byte aligned");
    dim3 grid(B, LN_NB);
    hipLaunchKernelGGL(ln_fwd_stats_kernel<false>, grid, dim3(256), 0, s, a, r, bias, relu, F, n_per, z1, partial1, (const float*)nullptr,
                       (const unsigned char*)nullptr, 1.f);
    hipLaunchKernelGGL(ln_fwd_apply_kernel, grid, dim3(256), 0, s, (const float*)z1, (const float*)partial1, gamma1, beta1, n_per, F, eps,
                       z2, n_per, stats1, r, partial2, (unsigned short*)nullptr, OperandFmt{0, 3, 1.f, nullptr});
    hipLaunchKernelGGL(ln_fwd_apply_kernel, grid, dim3(256), 0, s, (const float*)z2, (const float*)partial2, gamma2, beta2, n_per, F, eps,
                       y, yb, stats2, (const float*)nullptr, (float*)nullptr, (unsigned short*)nullptr, OperandFmt{0, 3, 1.f, nullptr});
    return check_launch("lpm_layer_norm_pair_fwd");
}

extern "C" int lpm_layer_norm_fwd(const float* a, const float* r, const float* gamma, const float* beta, int B, int L, int F,
                                  float eps, float* y, float* z, float* stats, void* workspace, size_t workspace_bytes,
                                  lpm_stream_t stream) {
    return lpm_layer_norm_act_fwd(a, nullptr, 0, r, gamma, beta, B, L, F, eps, y, 0, z, stats, workspace, workspace_bytes, stream);
}

static int layer_norm_act_bwd_impl(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats,
                                      const float* gamma, const float* a,
                                      const float* bias, int relu, int B, int L, int F, float* dz, float* da, float* dgamma,
                                      float* dbeta, float* dbias, const float* dr_extra, void* da_image, void* workspace,
                                      size_t workspace_bytes, lpm_stream_t stream, const unsigned char* mask, float mask_scale,
                                      const LpmOperandFormat* fmt = nullptr) {
    using namespace lpm;
    LPM_REQUIRE(dy && z && stats && gamma && dz && dgamma && dbeta, LPM_ERR_BADARG, "lpm_layer_norm_act_bwd: null pointer");
    if (const int rc = operand_fmt_check(fmt, "lpm_layer_norm_act_bwd")) return rc;
    LPM_REQUIRE(!mask || ((da || da_image) && ((uintptr_t)mask & 3) == 0), LPM_ERR_BADARG,
                "lpm_layer_norm_act_bwd: a keep mask needs da / da_image and 4-byte alignment");
    LPM_REQUIRE(!dr_extra || da || da_image, LPM_ERR_BADARG, "lpm_layer_norm_act_bwd: dr_extra needs a separate da (or da_image)");
    LPM_REQUIRE(!relu || (bias && a && (da || da_image)), LPM_ERR_BADARG, "lpm_layer_norm_act_bwd: relu needs a, bias and da / da_image");
    LPM_REQUIRE(!bias || dbias, LPM_ERR_BADARG, "lpm_layer_norm_act_bwd: a fused bias needs dbias");
    LPM_LN_CHECK("lpm_layer_norm_act_bwd");
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* colpart = partial + (size_t)B * LN_NB * 2;
    float* biaspart = colpart + (size_t)B * LN_NB * 2 * F;
    float* tmp = biaspart + (size_t)B * LN_NB * F;
    dim3 grid(B, LN_NB);
    const int nblk = B * LN_NB;
    const int64_t dyb = dy_batch_stride ? dy_batch_stride : (int64_t)L * F;
    LPM_REQUIRE(dyb >= (int64_t)L * F && dyb % 4 == 0 && ((uintptr_t)dy & 15) == 0, LPM_ERR_BADARG,
                "lpm_layer_norm_act_bwd: dy_batch_stride must be >= L*F and a multiple of 4, dy 16-byte aligned");
    unsigned* counters = (unsigned*)(tmp + (size_t)LN_RS * 4 * F);
    hipLaunchKernelGGL(ln_bwd_stats_kernel, grid, dim3(256), 0, s, dy, z, stats, gamma, L, F, partial, colpart, dyb, counters);
    if (mask)
        hipLaunchKernelGGL(ln_bwd_apply_kernel<true>, grid, dim3(256), 0, s, dy, z, stats, gamma, partial, L, F, dz, a, bias, relu, da, biaspart,
                           dr_extra, dyb, (unsigned short*)da_image, mask, mask_scale, operand_fmt(fmt));
    else
        hipLaunchKernelGGL(ln_bwd_apply_kernel<false>, grid, dim3(256), 0, s, dy, z, stats, gamma, partial, L, F, dz, a, bias, relu, da, biaspart,
                           dr_extra, dyb, (unsigned short*)da_image, (const unsigned char*)nullptr, 1.f, operand_fmt(fmt));
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((F + 63) / 64, LN_RS, bias ? 2 : 1), dim3(256), 0, s, colpart, biaspart, nblk, F, tmp, dgamma,
                       dbeta, dbias, counters);
    return check_launch("lpm_layer_norm_act_bwd");
}
extern "C" int lpm_layer_norm_act_bwd(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats,
                                      const float* gamma, const float* a,
                                      const float* bias, int relu, int B, int L, int F, float* dz, float* da, float* dgamma,
                                      float* dbeta, float* dbias, const float* dr_extra, void* da_image, void* workspace,
                                      size_t workspace_bytes, lpm_stream_t stream) {
    return layer_norm_act_bwd_impl(dy, dy_batch_stride, z, stats, gamma, a, bias, relu, B, L, F, dz, da, dgamma, dbeta, dbias, dr_extra,
                                   da_image, workspace, workspace_bytes, stream, nullptr, 1.f);
}
extern "C" int lpm_layer_norm_act_bwd_fmt(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats,
                                          const float* gamma, const float* a,
                                          const float* bias, int relu, int B, int L, int F, float* dz, float* da, float* dgamma,
                                          float* dbeta, float* dbias, const float* dr_extra, void* da_image, void* workspace,
                                          size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    return layer_norm_act_bwd_impl(dy, dy_batch_stride, z, stats, gamma, a, bias, relu, B, L, F, dz, da, dgamma, dbeta, dbias, dr_extra,
                                   da_image, workspace, workspace_bytes, stream, nullptr, 1.f, fmt);
}
// backward of lpm_layer_norm_act_mask_image_fwd: da (or da_image) = dz * [ReLU mask] * keep * mask_scale, dbias = its column sums
extern "C" int lpm_layer_norm_act_mask_bwd(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats,
                                           const float* gamma, const float* a, const float* bias, int relu, const unsigned char* mask,
                                           float mask_scale, int B, int L, int F, float* dz, float* da, float* dgamma, float* dbeta,
                                           float* dbias, const float* dr_extra, void* da_image, void* workspace, size_t workspace_bytes,
                                           lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(mask, LPM_ERR_BADARG, "lpm_layer_norm_act_mask_bwd: null keep mask");
    return layer_norm_act_bwd_impl(dy, dy_batch_stride, z, stats, gamma, a, bias, relu, B, L, F, dz, da, dgamma, dbeta, dbias, dr_extra,
                                   da_image, workspace, workspace_bytes, stream, mask, mask_scale);
}

// ... with da_image in either operand format (round 6: the V2 encoder's attention half as one node hands the masked gradient straight to
// attention_bn + output_transform's backward as its GEMMs' operand image)
extern "C" int lpm_layer_norm_act_mask_bwd_fmt(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats,
                                               const float* gamma, const float* a, const float* bias, int relu, const unsigned char* mask,
                                               float mask_scale, int B, int L, int F, float* dz, float* da, float* dgamma, float* dbeta,
                                               float* dbias, const float* dr_extra, void* da_image, void* workspace, size_t workspace_bytes,
                                               const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(mask, LPM_ERR_BADARG, "lpm_layer_norm_act_mask_bwd: null keep mask");
    return layer_norm_act_bwd_impl(dy, dy_batch_stride, z, stats, gamma, a, bias, relu, B, L, F, dz, da, dgamma, dbeta, dbias, dr_extra,
                                   da_image, workspace, workspace_bytes, stream, mask, mask_scale, fmt);
}

extern "C" int lpm_layer_norm_bwd(const float* dy, const float* z, const float* stats, const float* gamma, int B, int L, int F,
                                  float* dz, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                  lpm_stream_t stream) {
    return lpm_layer_norm_act_bwd(dy, 0, z, stats, gamma, nullptr, nullptr, 0, B, L, F, dz, nullptr, dgamma, dbeta, nullptr, nullptr, nullptr,
                                  workspace, workspace_bytes, stream);
}
