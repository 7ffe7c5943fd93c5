// Training-mode batch-norm plumbing around the hot kernels (slim.batch_norm semantics,
// SURVEY.md App. B): statistics fold (frame_level_models.py:2266,2784) and the backward
// through the batch statistics of the assignment logits (App. F.4).
#include "lpm_common.h"
#include "operand_format.h"

namespace lpm {

// partial [nblk, 2, C] -> mean/var/scale/shift (+ moving averages); 1024 threads per 16 columns (partial_colsums16)
template <int CW>
__global__ __launch_bounds__(1024) void bn_fold_kernel(const float* __restrict__ partial, int nblk, int C,
                                                       double inv_rows, double unbias,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float decay,
                                                       float* mean, float* var, float* scale, float* shift,
                                                       float* moving_mean, float* moving_var) {
    double s, q;
    int c;
    partial_colsums<CW>(partial, nblk, 2 * (int64_t)C, C, C, s, q, c);
    if (threadIdx.x < CW && c < C) {
        const double mu = s * inv_rows;
        double vr = q * inv_rows - mu * mu;
        if (vr < 0.0) vr = 0.0;
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = g * (float)(1.0 / sqrt(vr + (double)eps));
        if (mean) mean[c] = (float)mu;
        if (var) var[c] = (float)vr;
        scale[c] = sc;
        shift[c] = b - (float)mu * sc;
        if (moving_mean) {
            moving_mean[c] = moving_mean[c] * decay + (float)mu * (1.f - decay);
            moving_var[c] = moving_var[c] * decay + (float)(vr * unbias) * (1.f - decay);
        }
    }
}

// Pre-activation of the channel-last batch norms (lpm_bn_rows_act_fwd / lpm_bn_act_bwd): the normalised tensor is act(x + bias) with
// x the raw output of the dense layer in front (tf.layers.dense(use_bias=True, activation=relu) -> slim.batch_norm,
// transformer_utils.py:741-760).  It is formed where it is read -- statistics, apply, both backward passes -- and never stored.
__device__ __forceinline__ float4 bn_preact(float4 v, const float* __restrict__ pb, int c, int relu) {
    if (pb) {
        const float4 b = *reinterpret_cast<const float4*>(pb + c);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    return v;
}

// four consecutive elements of the normalised tensor: float4 number i4 of an fp32 matrix, or the same four of a bf16 matrix (8 bytes; the
// bf16-storage configuration keeps its logits that way: round 6 -- a .float() copy of the [38 400, 512] logits cost 62 us per step)
__device__ __forceinline__ float4 bn_load4(const float* __restrict__ base, int64_t i4, int bf16) {
    if (!bf16) return reinterpret_cast<const float4*>(base)[i4];
    const uint2 w = reinterpret_cast<const uint2*>(base)[i4];
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
}

// pass 1 of the BN backward: per-block column partials of dlt and dlt*Lhat.
constexpr int BNB_ROWS = 64;
// column chunks (gridDim.y) of the two statistics passes: one per 256 float4 column groups, at most 8; the per-block partial sums of a
// column are the same numbers whichever workgroup forms them
static inline unsigned bn_col_chunks(int C) {
    static const int cap = [] { const char* e = getenv("LPM_BN_COL_CHUNKS"); return e ? atoi(e) : 8; }();      // 1: one chunk, the round-5 launch (A/B)
    const int c = (C / 4 + 255) / 256;
    return (unsigned)(c < 1 ? 1 : (c > cap ? (cap < 1 ? 1 : cap) : c));
}
// Threads are (row group, float4 column): 256 / (K/4) row groups when K/4 divides 256 (K = 256: four rows of 1 KB per
// round), else one row group striding the columns; four rounds' loads are issued together.  (One thread per column walking
// 64 rows with one 4-byte load in flight ran at 1.9 TB/s.)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dlt,
                                                             const float* __restrict__ logits,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ var, float eps, int M,
                                                             int K, float* __restrict__ partial, const float* __restrict__ pb, int relu,
                                                             int lbf16) {
    __shared__ float4 red[2][256];
    const int r0 = blockIdx.x * BNB_ROWS;
    const int r1 = min(M, r0 + BNB_ROWS);
    const int tid = threadIdx.x, K4 = K >> 2;
    const int RG = (K4 < 256 && 256 % K4 == 0) ? 256 / K4 : 1;
    const int rg = RG > 1 ? tid / K4 : 0;
    // (gridDim.y column chunks: a [24000, 4096] tensor is 375 row blocks -- 1.5 rounds of the chip at one 4-wave workgroup per CU, each
    // thread walking four column groups one after the other with 32 KB in flight per CU: 3.9 TB/s; as 375 x 4 workgroups 6 per CU)
    for (int c4 = RG > 1 ? tid % K4 : tid + 256 * (int)blockIdx.y; c4 < K4; c4 += (RG > 1 ? K4 : 256 * (int)gridDim.y)) {      // RG > 1: exactly one pass, every thread in it
        const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * c4);
        const float4 vr = *reinterpret_cast<const float4*>(var + 4 * c4);
        const float4 rs = make_float4(rsqrtf(vr.x + eps), rsqrtf(vr.y + eps), rsqrtf(vr.z + eps), rsqrtf(vr.w + eps));
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
        auto add = [&](const float4 d, float4 l) {
            l = bn_preact(l, pb, 4 * c4, relu);
            s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
            q.x += d.x * ((l.x - mu.x) * rs.x); q.y += d.y * ((l.y - mu.y) * rs.y);
            q.z += d.z * ((l.z - mu.z) * rs.z); q.w += d.w * ((l.w - mu.w) * rs.w);
        };
        const float4* dp = reinterpret_cast<const float4*>(dlt) + c4;
        int r = r0 + rg;
        for (; r + 3 * RG < r1; r += 4 * RG) {
            float4 d[4], l[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = dp[(int64_t)(r + u * RG) * K4];
                l[u] = bn_load4(logits, (int64_t)(r + u * RG) * K4 + c4, lbf16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(d[u], l[u]);
        }
        for (; r < r1; r += RG) add(dp[(int64_t)r * K4], bn_load4(logits, (int64_t)r * K4 + c4, lbf16));
        if (RG > 1) {
            red[0][tid] = s;
            red[1][tid] = q;
            __syncthreads();
            if (rg == 0)
                for (int i = 1; i < RG; ++i) {
                    const float4 a = red[0][i * K4 + c4], b = red[1][i * K4 + c4];
                    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                    q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
                }
        }
        if (rg == 0) {
            float* p = partial + (int64_t)blockIdx.x * 2 * K;
            *reinterpret_cast<float4*>(p + 4 * c4) = s;
            *reinterpret_cast<float4*>(p + K + 4 * c4) = q;
        }
    }
}

// pass 2: reduce partials -> dbeta (sum dlt), dgamma (sum dlt*Lhat)
__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int K,
                                                             float* dgamma, float* dbeta) {
    double s, q;
    int c;
    partial_colsums16(partial, nblk, 2 * (int64_t)K, K, K, s, q, c);
    if (threadIdx.x < 16 && c < K) {
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
    }
}

// first stage for a tall partial array (rows x K, e.g. 512 x 4096 = 8 MB): grid (ceil(K / 64), BN_CS_SLICES), 256 threads = 4 row groups x
// 64 columns (256-byte row pieces); slice y sums rows y, y + BN_CS_SLICES, ... -> tmp[y][K]   (16-column workgroups over the tall array
// read 64-byte pieces: 108 us for 8 MB)
constexpr int BN_CS_SLICES = 16;
__global__ __launch_bounds__(256) void bn_colsum_stage1_kernel(const float* __restrict__ part, int rows, int K, float* __restrict__ tmp) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, y = blockIdx.y;
    float acc = 0.f;
    if (c < K) {
        for (int r = y + BN_CS_SLICES * rg; r < rows; r += BN_CS_SLICES * 4 * 4) {      // four rows per round: independent loads, fixed order
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + BN_CS_SLICES * 4 * u;
                v[u] = rr < rows ? part[(int64_t)rr * K + c] : 0.f;
            }
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < K) tmp[(int64_t)y * K + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

// column sums of one array [rows][K] (the bias gradient partials of the pre-activation form)
__global__ __launch_bounds__(1024) void bn_bwd_colsum_kernel(const float* __restrict__ part, int rows, int K, float* __restrict__ out) {
    double s, q;
    int c;
    partial_colsums16(part, rows, (int64_t)K, 0, K, s, q, c);
    if (threadIdx.x < 16 && c < K) out[c] = (float)s;
}

// logits_bn backward bookkeeping of the NetVladV2 attention (ops._MHACoreBN.backward): partial [nblk][2][L] holds per (batch, head) the
// column sums of dz and dz * s over the queries.  -> dbeta = sum dz, dgamma = sum dz * s_hat = rstd (sum dz s - mean sum dz), and the
// two correction vectors the main pass subtracts (training: the gradient through the batch statistics)
//   corr_b = kscale * (dgamma / n) * rstd,   corr_a = kscale * (dbeta / n - mean * rstd * dgamma / n)        (fp64, one launch)
template <int CW>
__global__ __launch_bounds__(1024) void mha_bn_corrections_kernel(const float* __restrict__ partial, int nblk, int L,
                                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                                  const float* __restrict__ kscale, float eps, double inv_n,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ corr_a, float* __restrict__ corr_b) {
    double s, q;
    int c;
    partial_colsums<CW>(partial, nblk, 2 * (int64_t)L, L, L, s, q, c);
    if (threadIdx.x < CW && c < L) {
        const double rstd = 1.0 / sqrt((double)var[c] + (double)eps), mu = (double)mean[c];
        const double sdz_hat = rstd * (q - mu * s);
        dbeta[c] = (float)s;
        dgamma[c] = (float)sdz_hat;
        if (corr_a) {
            const double c1 = s * inv_n, c2 = sdz_hat * inv_n, ks = (double)kscale[c];
            corr_b[c] = (float)(ks * c2 * rstd);
            corr_a[c] = (float)(ks * (c1 - mu * rstd * c2));
        }
    }
}

// A float4 as the split-bf16 operand image of the encoder GEMMs (split_gemm.hip): row-major [M][3 C] bf16, planes [hi | lo | hi]
// (activations, order 0) or [hi | hi | lo] (gradients, order 1); hi = bf16(v), lo = bf16(v - hi), round-to-nearest-even both.
__device__ __forceinline__ unsigned bn_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// Round 5: fmt.f16 -- the fp16 operand formats (operand_format.h): planes of v * scale, three for an activation / two for a gradient;
// vmax collects max |v| for the host's delayed scale.
__device__ __forceinline__ void bn_store_image(unsigned short* __restrict__ img, int64_t row, int c, int C, int order, float4 v, const OperandFmt& fmt,
                                               float& vmax) {
    vmax = of_amax4(vmax, v.x, v.y, v.z, v.w);
    if (fmt.f16) {
        uint2 hi, lo;
        of_split4(v.x, v.y, v.z, v.w, 1, fmt.scale, hi, lo);
        of_store_row4(img + row * fmt.planes * (int64_t)C, C, c, hi, lo, fmt.planes, order);
        return;
    }
    const unsigned hx = bn_rne(v.x), hy = bn_rne(v.y), hz = bn_rne(v.z), hw = bn_rne(v.w);
    const unsigned lx = bn_rne(v.x - __uint_as_float(hx << 16)), ly = bn_rne(v.y - __uint_as_float(hy << 16));
    const unsigned lz = bn_rne(v.z - __uint_as_float(hz << 16)), lw = bn_rne(v.w - __uint_as_float(hw << 16));
    const uint2 hi = make_uint2(hx | (hy << 16), hz | (hw << 16)), lo = make_uint2(lx | (ly << 16), lz | (lw << 16));
    unsigned short* p = img + row * 3 * (int64_t)C + c;
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + C) = order ? hi : lo;
    *reinterpret_cast<uint2*>(p + 2 * (int64_t)C) = order ? lo : hi;
}

// pass 3: dl = gamma*rstd*(dlt - mean_r(dlt) - Lhat*mean_r(dlt*Lhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dlt,
                                                           const float* __restrict__ logits,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ var,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float eps, int M, int K,
                                                           float* __restrict__ dl, const float* __restrict__ pb, int relu,
                                                           float* __restrict__ dbpart, unsigned short* __restrict__ dl3, const OperandFmt fmt,
                                                           int lbf16) {
    float vmax = 0.f;
    // dl3 != null: the result leaves as the GRADIENT image [M][3K] = [hi | hi | lo] of the dense layer in front instead of as fp32
    // pb / relu: the normalised tensor was act(logits + pb); dl is then the gradient of the RAW logits (masked where the ReLU was off)
    // and dbpart [stride / (K/4)][K] receives this thread's column sums of it (the bias gradient; needs the fixed-column arrangement)
    // dl = A d + Bq (l - mean) + Cq per column, A = gamma rstd, Bq = -gamma rstd^2 dgamma / M, Cq = -gamma rstd dbeta / M.  When the
    // grid stride is a multiple of the row length a thread keeps its four columns and the coefficients are formed once (the
    // launcher arranges that); otherwise once per element group.  (l - mean) is formed first: columns with |mean| >> sigma.
    const int64_t total4 = (int64_t)M * K / 4;
    const float invM = 1.f / (float)M;
    const int K4 = K / 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const bool fixed = (stride % K4) == 0;
    float4 A, Bq, Cq, mu;
    auto coeffs = [&](int c) {
        const float4 vr = *reinterpret_cast<const float4*>(var + c);
        mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 dg = *reinterpret_cast<const float4*>(dgamma + c), db = *reinterpret_cast<const float4*>(dbeta + c);
        const float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float rx = rsqrtf(vr.x + eps), ry = rsqrtf(vr.y + eps), rz = rsqrtf(vr.z + eps), rw = rsqrtf(vr.w + eps);
        A = make_float4(g.x * rx, g.y * ry, g.z * rz, g.w * rw);
        Bq = make_float4(-A.x * rx * (dg.x * invM), -A.y * ry * (dg.y * invM), -A.z * rz * (dg.z * invM), -A.w * rw * (dg.w * invM));
        Cq = make_float4(-A.x * (db.x * invM), -A.y * (db.y * invM), -A.z * (db.z * invM), -A.w * (db.w * invM));
    };
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (fixed && i0 < total4) coeffs((int)(i0 % K4) * 4);
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto one = [&](int64_t i, const float4 d, float4 l) {
        if (!fixed) coeffs((int)(i % K4) * 4);
        l = bn_preact(l, pb, (int)(i % K4) * 4, relu);
        float4 o;
        o.x = fmaf(A.x, d.x, fmaf(Bq.x, l.x - mu.x, Cq.x));
        o.y = fmaf(A.y, d.y, fmaf(Bq.y, l.y - mu.y, Cq.y));
        o.z = fmaf(A.z, d.z, fmaf(Bq.z, l.z - mu.z, Cq.z));
        o.w = fmaf(A.w, d.w, fmaf(Bq.w, l.w - mu.w, Cq.w));
        if (pb && relu) {                  // l = relu(.): zero exactly where the unit was off
            o.x = l.x > 0.f ? o.x : 0.f; o.y = l.y > 0.f ? o.y : 0.f; o.z = l.z > 0.f ? o.z : 0.f; o.w = l.w > 0.f ? o.w : 0.f;
        }
        bsum.x += o.x; bsum.y += o.y; bsum.z += o.z; bsum.w += o.w;
        if (dl3) bn_store_image(dl3, i / K4, (int)(i % K4) * 4, K, 1, o, fmt, vmax);
        else reinterpret_cast<float4*>(dl)[i] = o;
    };
    const float4* dp = reinterpret_cast<const float4*>(dlt);
    int64_t i = i0;
    for (; i + 3 * stride < total4; i += 4 * stride) {          // four grid strides' loads together
        float4 d[4], l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d[u] = dp[i + u * stride];
            l[u] = bn_load4(logits, i + u * stride, lbf16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u * stride, d[u], l[u]);
    }
    for (; i < total4; i += stride) one(i, dp[i], bn_load4(logits, i, lbf16));
    if (dbpart && i0 < stride)            // (fixed: the thread kept columns 4 (i0 % K4) ..; threads beyond the data wrote nothing: zero sums)
        *reinterpret_cast<float4*>(dbpart + (i0 / K4) * K + (i0 % K4) * 4) = bsum;
    if (dl3) of_amax_commit(fmt.amax, vmax);
}

// ---- channel-last batch norm of a [M, C] matrix (the V2 encoder's [B, L, C] tensors seen as rows) -----------------------------
// statistics: per 64-row block column (sum, sum of squares) -> partial [nblk][2][C]  (then bn_fold_kernel)
__global__ __launch_bounds__(256) void bn_rows_stats_kernel(const float* __restrict__ x, int M, int C, float* __restrict__ partial,
                                                            const float* __restrict__ pb, int relu) {
    __shared__ float4 red[2][256];
    const int r0 = blockIdx.x * BNB_ROWS, r1 = min(M, r0 + BNB_ROWS);
    const int tid = threadIdx.x, C4 = C / 4;
    const int RG = (C4 < 256 && 256 % C4 == 0) ? 256 / C4 : 1;      // thread layout as in bn_bwd_partial_kernel
    const int rg = RG > 1 ? tid / C4 : 0;
    for (int c4 = RG > 1 ? tid % C4 : tid + 256 * (int)blockIdx.y; c4 < C4; c4 += (RG > 1 ? C4 : 256 * (int)gridDim.y)) {     // (column chunks: see bn_bwd_partial_kernel)
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
        auto add = [&](float4 v) {
            v = bn_preact(v, pb, 4 * c4, relu);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
        };
        const float4* xp = reinterpret_cast<const float4*>(x) + c4;
        int r = r0 + rg;
        for (; r + 7 * RG < r1; r += 8 * RG) {              // eight rows' loads together (round 4: four left a 375-workgroup grid at
            float4 v[8];                                    // 3.6 TB/s on the [24000, 4096] tensor of cfg-3), rows consumed in order
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)(r + u * RG) * C4];
#pragma unroll
            for (int u = 0; u < 8; ++u) add(v[u]);
        }
        for (; r + 3 * RG < r1; r += 4 * RG) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = xp[(int64_t)(r + u * RG) * C4];
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u]);
        }
        for (; r < r1; r += RG) add(xp[(int64_t)r * C4]);
        if (RG > 1) {
            red[0][tid] = s;
            red[1][tid] = q;
            __syncthreads();
            if (rg == 0)
                for (int i = 1; i < RG; ++i) {
                    const float4 a = red[0][i * C4 + c4], b = red[1][i * C4 + c4];
                    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                    q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
                }
        }
        if (rg == 0) {
            float* p = partial + (int64_t)blockIdx.x * 2 * C + 4 * c4;
            *reinterpret_cast<float4*>(p) = s;
            *reinterpret_cast<float4*>(p + C) = q;
        }
    }
}
// y = x * scale[c] + shift[c]
__global__ __launch_bounds__(256) void bn_rows_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int64_t total4, int C4,
                                                            float* __restrict__ y, const float* __restrict__ pb, int relu,
                                                            unsigned short* __restrict__ y3, const OperandFmt fmt) {
    float vmax = 0.f;
    // y3 != null: the result leaves as the ACTIVATION image [M][3C] = [hi | lo | hi] of the next dense layer instead of as fp32
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const float4 v = bn_preact(reinterpret_cast<const float4*>(x)[i], pb, c, relu);
        const float4 a = *reinterpret_cast<const float4*>(scale + c), b = *reinterpret_cast<const float4*>(shift + c);
        const float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
        if (y3) bn_store_image(y3, i / C4, c, C4 * 4, 0, o, fmt, vmax);
        else reinterpret_cast<float4*>(y)[i] = o;
    }
    if (y3) of_amax_commit(fmt.amax, vmax);
}

}  // namespace lpm

extern "C" size_t lpm_bn_rows_workspace_bytes(int M, int C) {
    const int nblk = (M + lpm::BNB_ROWS - 1) / lpm::BNB_ROWS;
    return ((size_t)nblk * 2 * C + 2 * (size_t)C) * sizeof(float);
}

static int bn_rows_fwd_impl(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma, const float* beta,
                            float eps, float decay, int biased_moving_variance, float* y, float* mean, float* var, float* moving_mean,
                            float* moving_var, void* workspace, size_t workspace_bytes, lpm_stream_t stream, void* y3 = nullptr,
                            const LpmOperandFormat* fmt = nullptr);
extern "C" int lpm_bn_rows_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float decay,
                               int biased_moving_variance, float* y, float* mean, float* var, float* moving_mean,
                               float* moving_var, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    return bn_rows_fwd_impl(x, nullptr, 0, M, C, gamma, beta, eps, decay, biased_moving_variance, y, mean, var, moving_mean, moving_var,
                            workspace, workspace_bytes, stream);
}
// y = batch_norm(act(x + pre_bias)): the bias add (+ ReLU) of the dense layer in front rides in the statistics and apply passes
extern "C" int lpm_bn_rows_act_fwd(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma, const float* beta,
                                   float eps, float decay, int biased_moving_variance, float* y, float* mean, float* var,
                                   float* moving_mean, float* moving_var, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(pre_bias && (((uintptr_t)pre_bias) & 15) == 0, LPM_ERR_BADARG, "lpm_bn_rows_act_fwd: needs a 16-byte aligned bias");
    return bn_rows_fwd_impl(x, pre_bias, pre_relu, M, C, gamma, beta, eps, decay, biased_moving_variance, y, mean, var, moving_mean, moving_var,
                            workspace, workspace_bytes, stream);
}
// out3 [M, 3C] bf16: batch_norm(act(x + pre_bias)) written ONLY as the split-bf16 activation image [hi | lo | hi] the dense layer behind
// it reads (FeedForwardNetworkMod, transformer_utils.py:741-756: dense -> relu -> batch_norm -> dense): no fp32 copy, no split pass
extern "C" int lpm_bn_rows_act_image_fwd(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma,
                                         const float* beta, float eps, float decay, int biased_moving_variance, void* out3, float* mean,
                                         float* var, float* moving_mean, float* moving_var, void* workspace, size_t workspace_bytes,
                                         lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(pre_bias && out3 && ((((uintptr_t)pre_bias) | (uintptr_t)out3) & 15) == 0 && C % 8 == 0, LPM_ERR_BADARG,
                "lpm_bn_rows_act_image_fwd: needs a 16-byte aligned bias and image and C %% 8 == 0");
    return bn_rows_fwd_impl(x, pre_bias, pre_relu, M, C, gamma, beta, eps, decay, biased_moving_variance, nullptr, mean, var, moving_mean,
                            moving_var, workspace, workspace_bytes, stream, out3);
}
extern "C" int lpm_bn_rows_act_image_fwd_fmt(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma,
                                             const float* beta, float eps, float decay, int biased_moving_variance, void* out_img, float* mean,
                                             float* var, float* moving_mean, float* moving_var, void* workspace, size_t workspace_bytes,
                                             const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(pre_bias && out_img && ((((uintptr_t)pre_bias) | (uintptr_t)out_img) & 15) == 0 && C % 8 == 0, LPM_ERR_BADARG,
                "lpm_bn_rows_act_image_fwd: needs a 16-byte aligned bias and image and C %% 8 == 0");
    if (const int rc = operand_fmt_check(fmt, "lpm_bn_rows_act_image_fwd")) return rc;
    return bn_rows_fwd_impl(x, pre_bias, pre_relu, M, C, gamma, beta, eps, decay, biased_moving_variance, nullptr, mean, var, moving_mean,
                            moving_var, workspace, workspace_bytes, stream, out_img, fmt);
}
static int bn_rows_fwd_impl(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma, const float* beta,
                            float eps, float decay, int biased_moving_variance, float* y, float* mean, float* var, float* moving_mean,
                            float* moving_var, void* workspace, size_t workspace_bytes, lpm_stream_t stream, void* y3, const LpmOperandFormat* fmt) {
    using namespace lpm;
    LPM_REQUIRE(x && (y || y3) && mean && var && workspace, LPM_ERR_BADARG, "lpm_bn_rows_fwd: null pointer");
    LPM_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_bn_rows_fwd: need C %% 4 == 0 and 16-byte aligned pointers (M=%d C=%d)", M, C);
    LPM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), LPM_ERR_BADARG, "lpm_bn_rows_fwd: moving statistics go together");
    LPM_REQUIRE(workspace_bytes >= lpm_bn_rows_workspace_bytes(M, C), LPM_ERR_WORKSPACE, "lpm_bn_rows_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (M + BNB_ROWS - 1) / BNB_ROWS;
    float* partial = (float*)workspace;
    float* scale = partial + (size_t)nblk * 2 * C;
    float* shift = scale + C;
    hipLaunchKernelGGL(bn_rows_stats_kernel, dim3(nblk, bn_col_chunks(C)), dim3(256), 0, s, x, M, C, partial, pre_bias, pre_relu);
    const double unbias = (biased_moving_variance || M <= 1) ? 1.0 : (double)M / (double)(M - 1);
    hipLaunchKernelGGL(bn_fold_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, s, partial, nblk, C, 1.0 / (double)M, unbias, gamma, beta, eps,
                       decay, mean, var, scale, shift, moving_mean, moving_var);
    const int64_t total4 = (int64_t)M * C / 4;
    const int64_t want = (total4 + 255) / 256;
    hipLaunchKernelGGL(bn_rows_apply_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, x, scale, shift, total4, C / 4, y, pre_bias,
                       pre_relu, (unsigned short*)y3, operand_fmt(fmt));
    return check_launch("lpm_bn_rows_fwd");
}

namespace lpm {
// ---- batch norm of a SMALL [M, C] matrix (M <= 256 rows: the clip-level tail of the model, frame_level_models.py:2321-2368) in ONE launch
// each way.  torch's fused batch norm is three launches forward and two backward, the activation behind it one or two more; at 80-128
// rows every one of them is a ~5 us dependent launch that does nothing.  One workgroup per 32 columns, 256 threads = 32 columns x 8 row
// groups, the rows of a thread in registers (two-pass variance), statistics through LDS in a fixed order.
//   act 0: y = bn(x)    act 1: y = relu6(bn(x))  (tf.nn.relu6, :2337)    act 2: y = mul * sigmoid(bn(x))  (context gating, :2367-2368)
// Training mode only: batch statistics, moving averages updated in place with the UNBIASED variance (TF's fused rank-2 batch norm).
constexpr int BS_RMAX = 32;                // rows per thread: M <= 8 * 32
__device__ __forceinline__ float bs_colsum(float v, float (*sh)[32], int cl, int rg) {      // sum over the 8 row groups of a column; all threads
    __syncthreads();
    sh[rg][cl] = v;
    __syncthreads();
    return ((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl])) + ((sh[4][cl] + sh[5][cl]) + (sh[6][cl] + sh[7][cl]));
}
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, int M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float decay, int act,
                                                           const float* __restrict__ mul, float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, float* moving_mean, float* moving_var) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    const bool ok = c < C;
    float v[BS_RMAX];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BS_RMAX; ++i) {
        const int r = rg + 8 * i;
        v[i] = (ok && r < M) ? x[(int64_t)r * C + c] : 0.f;
        s += v[i];
    }
    const float mu = bs_colsum(s, sh, cl, rg) / (float)M;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < BS_RMAX; ++i) {
        const int r = rg + 8 * i;
        const float d = (ok && r < M) ? v[i] - mu : 0.f;
        q = fmaf(d, d, q);
    }
    const float var = bs_colsum(q, sh, cl, rg) / (float)M;
    const float rstd = rsqrtf(var + eps);
    if (!ok) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
#pragma unroll
    for (int i = 0; i < BS_RMAX; ++i) {
        const int r = rg + 8 * i;
        if (r < M) {
            float z = fmaf((v[i] - mu) * rstd, g, b);
            if (act == 1) z = fminf(fmaxf(z, 0.f), 6.f);
            else if (act == 2) z = mul[(int64_t)r * C + c] / (1.f + __expf(-z));
            y[(int64_t)r * C + c] = z;
        }
    }
    if (rg == 0) {
        mean_out[c] = mu;
        rstd_out[c] = rstd;
        if (moving_mean) {
            const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
            moving_mean[c] = moving_mean[c] * decay + mu * (1.f - decay);
            moving_var[c] = moving_var[c] * decay + unb * (1.f - decay);
        }
    }
}
// dy -> dx (through the activation and the batch statistics), dgamma, dbeta, and (act 2) dmul = dy * sigmoid(bn(x))
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                           const float* __restrict__ mul, float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dmul) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    const bool ok = c < C;
    const float mu = ok ? mean[c] : 0.f, rs = ok ? rstd[c] : 0.f, g = (ok && gamma) ? gamma[c] : 1.f, b = (ok && beta) ? beta[c] : 0.f;
    float xh[BS_RMAX], dz[BS_RMAX];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < BS_RMAX; ++i) {
        const int r = rg + 8 * i;
        xh[i] = 0.f; dz[i] = 0.f;
        if (ok && r < M) {
            const int64_t o = (int64_t)r * C + c;
            xh[i] = (x[o] - mu) * rs;
            const float z = fmaf(xh[i], g, b), d = dy[o];
            if (act == 1) dz[i] = (z > 0.f && z < 6.f) ? d : 0.f;
            else if (act == 2) {
                const float sg = 1.f / (1.f + __expf(-z)), m = mul[o];
                dmul[o] = d * sg;
                dz[i] = d * m * sg * (1.f - sg);
            } else dz[i] = d;
            s += dz[i];
            q = fmaf(dz[i], xh[i], q);
        }
    }
    const float sdz = bs_colsum(s, sh, cl, rg), sdzx = bs_colsum(q, sh, cl, rg);
    if (!ok) return;
    const float a = g * rs, m1 = sdz / (float)M, m2 = sdzx / (float)M;
#pragma unroll
    for (int i = 0; i < BS_RMAX; ++i) {
        const int r = rg + 8 * i;
        if (r < M) dx[(int64_t)r * C + c] = a * (dz[i] - m1 - xh[i] * m2);
    }
    if (rg == 0) {
        dgamma[c] = sdzx;
        dbeta[c] = sdz;
    }
}
}  // namespace lpm

extern "C" int lpm_bn_small_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float decay, int act,
                                const float* mul, float* y, float* mean, float* rstd, float* moving_mean, float* moving_var,
                                lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && y && mean && rstd && (act != 2 || mul), LPM_ERR_BADARG, "lpm_bn_small_fwd: null pointer");
    LPM_REQUIRE(M > 0 && M <= 8 * BS_RMAX && C > 0 && act >= 0 && act <= 2, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_bn_small_fwd: need 1 <= M <= %d rows and act in {0, 1, 2} (M=%d)", 8 * BS_RMAX, M);
    LPM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), LPM_ERR_BADARG, "lpm_bn_small_fwd: moving_mean and moving_var go together");
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, M, C, gamma, beta, eps, decay, act, mul,
                       y, mean, rstd, moving_mean, moving_var);
    return check_launch("lpm_bn_small_fwd");
}

extern "C" int lpm_bn_small_bwd(const float* dy, const float* x, int M, int C, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, int act, const float* mul, float* dx, float* dgamma, float* dbeta, float* dmul,
                                lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dy && x && mean && rstd && dx && dgamma && dbeta && (act != 2 || (mul && dmul)), LPM_ERR_BADARG, "lpm_bn_small_bwd: null pointer");
    LPM_REQUIRE(M > 0 && M <= 8 * BS_RMAX && C > 0 && act >= 0 && act <= 2, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_bn_small_bwd: need 1 <= M <= %d rows and act in {0, 1, 2} (M=%d)", 8 * BS_RMAX, M);
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, dy, x, M, C, gamma, beta, mean, rstd, act,
                       mul, dx, dgamma, dbeta, dmul);
    return check_launch("lpm_bn_small_bwd");
}

extern "C" int lpm_bn_fold(const float* partial, int nblk, int C, int64_t rows, const float* gamma, const float* beta,
                           float eps, float decay, float* mean, float* var, float* scale, float* shift,
                           float* moving_mean, float* moving_var, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(partial && scale && shift, LPM_ERR_BADARG, "lpm_bn_fold: null pointer");
    LPM_REQUIRE(nblk > 0 && C > 0 && rows > 0, LPM_ERR_BADARG, "lpm_bn_fold: bad sizes");
    LPM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), LPM_ERR_BADARG,
                "lpm_bn_fold: moving_mean and moving_var must be given together");
    const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
    if (partial_colsums_cw(nblk) == 4)
        hipLaunchKernelGGL(bn_fold_kernel<4>, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C,
                           1.0 / (double)rows, unbias, gamma, beta, eps, decay, mean, var, scale, shift, moving_mean, moving_var);
    else
        hipLaunchKernelGGL(bn_fold_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C,
                           1.0 / (double)rows, unbias, gamma, beta, eps, decay, mean, var, scale, shift, moving_mean, moving_var);
    return check_launch("lpm_bn_fold");
}

extern "C" size_t lpm_bn_bwd_workspace_bytes(int M, int K) {
    const int nblk = (M + lpm::BNB_ROWS - 1) / lpm::BNB_ROWS;
    return (size_t)nblk * 2 * K * sizeof(float);
}

namespace lpm {
static int bn_bwd_grid(int M, int K) {
    const int64_t total4 = (int64_t)M * K / 4;
    const int64_t want = (total4 + 255) / 256;
    int grid = (int)(want < 2048 ? want : 2048);
    // a grid stride that is a multiple of the row length K/4 lets a thread keep its columns (coefficients formed once)
    const int K4 = K / 4;
    int step = K4;                                   // smallest workgroup count g with (g * 256) % K4 == 0: K4 / gcd(K4, 256)
    for (int a = K4, b = 256; b;) { const int t = a % b; a = b; b = t; step = K4 / a; }
    if (step <= grid) grid = grid / step * step;
    return grid;
}
}  // namespace lpm
static int bn_bwd_impl(const float* dlt, const float* logits, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                       const float* gamma, float eps, int M, int K, float* dl, float* dgamma, float* dbeta, float* dbias, void* workspace,
                       size_t workspace_bytes, lpm_stream_t stream, void* dl3 = nullptr, const LpmOperandFormat* fmt = nullptr,
                       int logits_bf16 = 0);
extern "C" int lpm_bn_bwd(const float* dlt, const float* logits, const float* mean, const float* var,
                          const float* gamma, float eps, int M, int K, float* dl, float* dgamma, float* dbeta,
                          void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    return bn_bwd_impl(dlt, logits, nullptr, 0, mean, var, gamma, eps, M, K, dl, dgamma, dbeta, nullptr, workspace, workspace_bytes, stream);
}
// ... of a tensor stored as bf16 (the bf16-storage configuration's logits): read in place, no fp32 copy (round 6)
extern "C" int lpm_bn_bwd_x16(const float* dlt, const void* logits_bf16, const float* mean, const float* var, const float* gamma, float eps, int M,
                              int K, float* dl, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    return bn_bwd_impl(dlt, (const float*)logits_bf16, nullptr, 0, mean, var, gamma, eps, M, K, dl, dgamma, dbeta, nullptr, workspace,
                       workspace_bytes, stream, nullptr, nullptr, 1);
}
// backward of lpm_bn_rows_act_fwd: x = the RAW dense output it normalised as act(x + pre_bias); dl = the gradient of x (the ReLU mask
// applied), dbias = its column sums.  0 from lpm_bn_act_bwd_supported: the thread layout cannot keep columns fixed for this shape.
extern "C" int lpm_bn_act_bwd_supported(int M, int K) {
    if (M <= 0 || K <= 0 || K % 4) return 0;
    return ((int64_t)lpm::bn_bwd_grid(M, K) * 256) % (K / 4) == 0 ? 1 : 0;
}
extern "C" size_t lpm_bn_act_bwd_workspace_bytes(int M, int K) {
    const int nblk = (M + lpm::BNB_ROWS - 1) / lpm::BNB_ROWS;
    const size_t rows = (size_t)lpm::bn_bwd_grid(M, K) * 256 / (size_t)(K / 4);
    return ((size_t)nblk * 2 * K + rows * K + (size_t)lpm::BN_CS_SLICES * K) * sizeof(float);
}
extern "C" int lpm_bn_act_bwd(const float* dlt, const float* x, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                              const float* gamma, float eps, int M, int K, float* dl, float* dgamma, float* dbeta, float* dbias,
                              void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(pre_bias && dbias && (((uintptr_t)pre_bias) & 15) == 0, LPM_ERR_BADARG, "lpm_bn_act_bwd: needs a 16-byte aligned bias and dbias");
    LPM_REQUIRE(lpm_bn_act_bwd_supported(M, K), LPM_ERR_UNSUPPORTED_SHAPE, "lpm_bn_act_bwd: shape not supported (M=%d K=%d)", M, K);
    LPM_REQUIRE(workspace_bytes >= lpm_bn_act_bwd_workspace_bytes(M, K), LPM_ERR_WORKSPACE, "lpm_bn_act_bwd: workspace too small");
    return bn_bwd_impl(dlt, x, pre_bias, pre_relu, mean, var, gamma, eps, M, K, dl, dgamma, dbeta, dbias, workspace, workspace_bytes, stream);
}
// ... with the gradient of x written ONLY as the split-bf16 gradient image dl3 [M, 3K] = [hi | hi | lo] the input-gradient and
// weight-gradient GEMMs of the dense layer in front read
extern "C" int lpm_bn_act_bwd_image(const float* dlt, const float* x, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                                    const float* gamma, float eps, int M, int K, void* dl3, float* dgamma, float* dbeta, float* dbias,
                                    void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(pre_bias && dbias && dl3 && ((((uintptr_t)pre_bias) | (uintptr_t)dl3) & 15) == 0 && K % 8 == 0, LPM_ERR_BADARG,
                "lpm_bn_act_bwd_image: needs a 16-byte aligned bias, image and dbias and K %% 8 == 0");
    LPM_REQUIRE(lpm_bn_act_bwd_supported(M, K), LPM_ERR_UNSUPPORTED_SHAPE, "lpm_bn_act_bwd_image: shape not supported (M=%d K=%d)", M, K);
    LPM_REQUIRE(workspace_bytes >= lpm_bn_act_bwd_workspace_bytes(M, K), LPM_ERR_WORKSPACE, "lpm_bn_act_bwd_image: workspace too small");
    return bn_bwd_impl(dlt, x, pre_bias, pre_relu, mean, var, gamma, eps, M, K, nullptr, dgamma, dbeta, dbias, workspace, workspace_bytes, stream, dl3);
}
extern "C" int lpm_bn_act_bwd_image_fmt(const float* dlt, const float* x, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                                        const float* gamma, float eps, int M, int K, void* dl_img, float* dgamma, float* dbeta, float* dbias,
                                        void* workspace, size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(pre_bias && dbias && dl_img && ((((uintptr_t)pre_bias) | (uintptr_t)dl_img) & 15) == 0 && K % 8 == 0, LPM_ERR_BADARG,
                "lpm_bn_act_bwd_image: needs a 16-byte aligned bias, image and dbias and K %% 8 == 0");
    LPM_REQUIRE(lpm_bn_act_bwd_supported(M, K), LPM_ERR_UNSUPPORTED_SHAPE, "lpm_bn_act_bwd_image: shape not supported (M=%d K=%d)", M, K);
    LPM_REQUIRE(workspace_bytes >= lpm_bn_act_bwd_workspace_bytes(M, K), LPM_ERR_WORKSPACE, "lpm_bn_act_bwd_image: workspace too small");
    if (const int rc = operand_fmt_check(fmt, "lpm_bn_act_bwd_image")) return rc;
    return bn_bwd_impl(dlt, x, pre_bias, pre_relu, mean, var, gamma, eps, M, K, nullptr, dgamma, dbeta, dbias, workspace, workspace_bytes, stream,
                       dl_img, fmt);
}
static int bn_bwd_impl(const float* dlt, const float* logits, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                       const float* gamma, float eps, int M, int K, float* dl, float* dgamma, float* dbeta, float* dbias, void* workspace,
                       size_t workspace_bytes, lpm_stream_t stream, void* dl3, const LpmOperandFormat* fmt, int logits_bf16) {
    using namespace lpm;
    LPM_REQUIRE(dlt && logits && mean && var && (dl || dl3) && dgamma && dbeta && workspace, LPM_ERR_BADARG,
                "lpm_bn_bwd: null pointer");
    LPM_REQUIRE(M > 0 && K > 0 && K % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_bn_bwd: need K %% 4 == 0 (M=%d K=%d)", M, K);
    LPM_REQUIRE((((uintptr_t)dlt | (uintptr_t)logits | (uintptr_t)mean | (uintptr_t)var | (uintptr_t)dl | (uintptr_t)workspace | (uintptr_t)gamma | (uintptr_t)dgamma | (uintptr_t)dbeta) & 15) == 0,
                LPM_ERR_BADARG, "lpm_bn_bwd: pointers must be 16-byte aligned");
    LPM_REQUIRE(workspace_bytes >= lpm_bn_bwd_workspace_bytes(M, K), LPM_ERR_WORKSPACE, "lpm_bn_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (M + BNB_ROWS - 1) / BNB_ROWS;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk, bn_col_chunks(K)), dim3(256), 0, s, dlt, logits, mean, var, eps, M, K, partial, pre_bias, pre_relu,
                       logits_bf16);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((K + 15) / 16), dim3(1024), 0, s, partial, nblk, K, dgamma, dbeta);
    const int grid = bn_bwd_grid(M, K);
    float* dbpart = dbias ? partial + (size_t)nblk * 2 * K : nullptr;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, s, dlt, logits, mean, var, gamma, dgamma, dbeta,
                       eps, M, K, dl, pre_bias, pre_relu, dbpart, (unsigned short*)dl3, operand_fmt(fmt), logits_bf16);
    if (dbias) {
        const int rows = (int)((int64_t)grid * 256 / (K / 4));
        if (rows > 4 * BN_CS_SLICES) {
            float* tmp = dbpart + (size_t)rows * K;
            hipLaunchKernelGGL(bn_colsum_stage1_kernel, dim3((K + 63) / 64, BN_CS_SLICES), dim3(256), 0, s, (const float*)dbpart, rows, K, tmp);
            hipLaunchKernelGGL(bn_bwd_colsum_kernel, dim3((K + 15) / 16), dim3(1024), 0, s, (const float*)tmp, BN_CS_SLICES, K, dbias);
        } else {
            hipLaunchKernelGGL(bn_bwd_colsum_kernel, dim3((K + 15) / 16), dim3(1024), 0, s, (const float*)dbpart, rows, K, dbias);
        }
    }
    return check_launch("lpm_bn_bwd");
}

/* partial [nblk][2][L] (lpm_mha_bwd's statistics pass) -> dgamma, dbeta of logits_bn and, when corr_a / corr_b are given (training), the
 * two correction vectors of the main backward pass; n = the number of logits per key position (B * h * L) */
extern "C" int lpm_mha_bn_corrections(const float* partial, int nblk, int L, const float* mean, const float* var, const float* kscale,
                                      float eps, int64_t n, float* dgamma, float* dbeta, float* corr_a, float* corr_b, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(partial && mean && var && dgamma && dbeta && ((corr_a == nullptr) == (corr_b == nullptr)) && (!corr_a || kscale), LPM_ERR_BADARG,
                "lpm_mha_bn_corrections: null pointer");
    LPM_REQUIRE(nblk > 0 && L > 0 && n > 0, LPM_ERR_BADARG, "lpm_mha_bn_corrections: bad sizes");
    if (partial_colsums_cw(nblk) == 4)
        hipLaunchKernelGGL(mha_bn_corrections_kernel<4>, dim3((L + 3) / 4), dim3(1024), 0, (hipStream_t)stream, partial, nblk, L, mean, var,
                           kscale, eps, 1.0 / (double)n, dgamma, dbeta, corr_a, corr_b);
    else
        hipLaunchKernelGGL(mha_bn_corrections_kernel<16>, dim3((L + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partial, nblk, L, mean, var,
                           kscale, eps, 1.0 / (double)n, dgamma, dbeta, corr_a, corr_b);
    return check_launch("lpm_mha_bn_corrections");
}
