// K4 -- multi-head attention core, forward + backward, never materialising [B,h,L,L].
//   o = softmax(scale * q k^T) v                          transformer_utils.py:564-581
//   MultiHeadAttentionBN variant: z = (q k^T) * key_scale[j] + key_shift[j] (folded logits_bn over the
//   key-position channel, :652-659); lpm_mha_logit_stats reduces the column statistics it needs.
//
// gfx950 mapping.  One 256-thread workgroup per (batch, head); K and V (L x d, d in {8,16}) are staged
// once into LDS (row stride 20 floats: both the "row-chunk" and the "column" MFMA operand walks are
// at most 2-way bank conflicted).  Each wave owns 16-query tiles and works on TRANSPOSED score tiles
// S^T[key, q] = K Q^T with v_mfma_f32_16x16x4_f32 (exact fp32): in that layout a lane owns one query
// column (q = lane & 15) and 4 keys per register, so (a) the softmax reductions are register-local plus
// two cross-lane steps, and (b) the probability tile is ALREADY the B operand of the next MFMA
// (O^T = V^T P^T) -- no LDS round trip, no transposes.  The backward recomputes the tiles from the
// saved log-sum-exp in two sweeps: per query tile (dQ) and per key tile (dK, dV), each reduction-free
// across waves.
#include "lpm_common.h"

namespace lpm {

constexpr int MH_S = 20;  // LDS row stride (floats) of the staged [L, d<=16] operands

// stage rows [0, L) of a [B, L, h*d] tensor (head hh) into LDS [L16][MH_S]; columns >= d and rows >= L stay 0
__device__ __forceinline__ void mha_stage(float* dst, const float* __restrict__ src, int64_t ld, int b, int L,
                                          int hh, int d, int tid) {
    const int d4 = d >> 2;
    for (int i = tid; i < L * d4; i += 256) {
        const int row = i / d4, c = (i % d4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + ((int64_t)b * L + row) * ld + hh * d + c);
        *reinterpret_cast<float4*>(dst + row * MH_S + c) = v;
    }
}

// One workgroup per (batch, head).  Measured alternatives at the V1 video shape (B=80, L=256, h=64, d=16):
// first form 429 us; + query fragments prefetched ahead of the K/V staging, two PV accumulator chains, affine /
// ragged handling compiled out -> 334 us (this form); persistent workgroups with the next item's K/V/Q fetched
// into registers during compute -> 400-456 us (the register ring costs a wave per SIMD or spills).
template <int NKT, bool AFFINE>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int64_t ld, int L, int h, int d,
                                                      float scale, const float* __restrict__ key_scale,
                                                      const float* __restrict__ key_shift, float* __restrict__ o,
                                                      int64_t ldo, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NQ = (NKT + 3) / 4;    // query tiles per wave
    const int nkt = (L + 15) >> 4, L16 = nkt * 16;
    float* Ks = smem;
    float* Vs = Ks + L16 * MH_S;
    float* ksc = Vs + L16 * MH_S;
    float* ksh = ksc + L16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int ns = d >> 2;  // reduction steps of 4 over the head dimension
    const bool ragged = (L != L16);

    // every query fragment this wave will need, issued before the K/V staging so the loads overlap it
    float qf[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int qrow = (wave + 4 * i) * 16 + l15;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[i][s] = (s < ns && qrow < L) ? q[((int64_t)b * L + qrow) * ld + hh * d + 4 * s + g] : 0.f;
    }
    if (ragged || d < 16) {               // pad rows / columns must read as zero; otherwise every word is overwritten
        for (int i = tid; i < 2 * L16 * MH_S; i += 256) smem[i] = 0.f;
        __syncthreads();
    }
    if (AFFINE) {
        for (int i = tid; i < L16; i += 256) {
            ksc[i] = (i < L) ? key_scale[i] : 1.f;
            ksh[i] = (i < L) ? key_shift[i] : 0.f;
        }
    }
    mha_stage(Ks, k, ld, b, L, hh, d, tid);
    mha_stage(Vs, v, ld, b, L, hh, d, tid);
    __syncthreads();

#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int qt = wave + 4 * i;
        if (qt >= nkt) break;
        const int qrow = qt * 16 + l15;
        float qs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qs[s] = qf[i][s] * scale;
        f32x4 p[NKT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt < nkt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* ka = Ks + (kt * 16 + l15) * MH_S + g;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < ns) acc = mfma16(ka[4 * s], qs[s], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    float z = acc[r];
                    if (AFFINE) z = fmaf(z, ksc[key], ksh[key]);
                    if (ragged && key >= L) z = -INFINITY;
                    acc[r] = z;
                    m = fmaxf(m, z);
                }
                p[kt] = acc;
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(p[kt][r] - m);
                    p[kt][r] = e;
                    sum += e;
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};   // two chains: the 40-cycle dependent latency hides
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt < nkt) {
                const float* va = Vs + (kt * 16 + 4 * g) * MH_S + l15;
                oa = mfma16(va[0], p[kt][0], oa);
                ob = mfma16(va[MH_S], p[kt][1], ob);
                oa = mfma16(va[2 * MH_S], p[kt][2], oa);
                ob = mfma16(va[3 * MH_S], p[kt][3], ob);
            }
        }
        const float inv = 1.f / sum;
        if (qrow < L) {
            if (4 * g < d) {
                float4 ov = make_float4((oa[0] + ob[0]) * inv, (oa[1] + ob[1]) * inv, (oa[2] + ob[2]) * inv, (oa[3] + ob[3]) * inv);
                *reinterpret_cast<float4*>(o + ((int64_t)b * L + qrow) * ldo + hh * d + 4 * g) = ov;
            }
            if (g == 0) lse[((int64_t)b * h + hh) * L + qrow] = m + __logf(sum);
        }
    }
}

// per (batch, head) column statistics of the raw logits s = q k^T over the query axis:
// partial[(b*h+hh)][0][key] = sum_q s, [1][key] = sum_q s^2
template <int NKT>
__global__ __launch_bounds__(256) void mha_logit_stats_kernel(const float* __restrict__ q,
                                                              const float* __restrict__ k, int64_t ld, int L, int h,
                                                              int d, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nkt = (L + 15) >> 4, L16 = nkt * 16;
    float* Ks = smem;
    float* red = Ks + L16 * MH_S;   // [4 waves][2][L16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int ns = d >> 2;
    for (int i = tid; i < L16 * MH_S; i += 256) smem[i] = 0.f;
    __syncthreads();
    mha_stage(Ks, k, ld, b, L, hh, d, tid);
    __syncthreads();
    f32x4 cs[NKT], cq[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        cs[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        cq[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int qt = wave; qt < nkt; qt += 4) {
        const int qrow = qt * 16 + l15;
        float qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[s] = (s < ns && qrow < L) ? q[((int64_t)b * L + qrow) * ld + hh * d + 4 * s + g] : 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt < nkt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* ka = Ks + (kt * 16 + l15) * MH_S + g;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < ns) acc = mfma16(ka[4 * s], qf[s], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cs[kt][r] += acc[r];            // rows q >= L have q == 0 -> contribute 0
                    cq[kt][r] = fmaf(acc[r], acc[r], cq[kt][r]);
                }
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = cs[kt][r], c = cq[kt][r];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o, 64);
                    c += __shfl_xor(c, o, 64);
                }
                if (l15 == 0) {
                    const int key = kt * 16 + 4 * g + r;
                    red[(wave * 2 + 0) * L16 + key] = a;
                    red[(wave * 2 + 1) * L16 + key] = c;
                }
            }
        }
    }
    __syncthreads();
    float* out = partial + ((int64_t)b * h + hh) * 2 * L;
    for (int key = tid; key < L; key += 256) {
        out[key] = red[0 * L16 + key] + red[2 * L16 + key] + red[4 * L16 + key] + red[6 * L16 + key];
        out[L + key] = red[1 * L16 + key] + red[3 * L16 + key] + red[5 * L16 + key] + red[7 * L16 + key];
    }
}

// backward, two kernels so that each stages only TWO operands (L16 x 20 floats each: 41 KB at L = 256, three
// workgroups per CU instead of one with all four staged):
//   mha_bwd_dq_kernel : K, V in LDS; one query tile per wave iteration; Q / dO / O fragments and lse from global
//   mha_bwd_dkv_kernel: Q, dO (+ lse, D_q) in LDS; one key tile per wave iteration; K / V fragments from global;
//                       also the per-(batch, head) column sums of dz and dz*s for the logits_bn backward
//                       (dz_partial [(b*h+hh)][0][key] = sum_q dz, [1][key] = sum_q dz * s).
// D_q = rowsum(P * dP) = <dO_q, O_q>.  corr_a / corr_b: ds = key_scale*dz - corr_a[key] - s*corr_b[key].
template <bool AFFINE>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int64_t ld,
                                                         const float* __restrict__ o, const float* __restrict__ dout,
                                                         int64_t ldo, const float* __restrict__ lse, int L, int h, int d,
                                                         float scale, const float* __restrict__ key_scale,
                                                         const float* __restrict__ key_shift, float* __restrict__ dq,
                                                         int64_t ldd, const float* __restrict__ corr_a,
                                                         const float* __restrict__ corr_b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nkt = (L + 15) >> 4, L16 = nkt * 16;
    float* Ks = smem;
    float* Vs = Ks + L16 * MH_S;
    float* ksc = Vs + L16 * MH_S;
    float* ksh = ksc + L16;
    float* cas = ksh + L16;
    float* cbs = cas + L16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int ns = d >> 2;

    const bool ragged = (L != L16);
    if (ragged || d < 16) {               // pad rows / columns must read as zero; otherwise every word is overwritten
        for (int i = tid; i < 2 * L16 * MH_S; i += 256) smem[i] = 0.f;
        __syncthreads();
    }
    if (AFFINE) {
        for (int i = tid; i < L16; i += 256) {
            ksc[i] = (i < L) ? key_scale[i] : 1.f;
            ksh[i] = (i < L) ? key_shift[i] : 0.f;
            cas[i] = (corr_a && i < L) ? corr_a[i] : 0.f;
            cbs[i] = (corr_b && i < L) ? corr_b[i] : 0.f;
        }
    }
    mha_stage(Ks, k, ld, b, L, hh, d, tid);
    mha_stage(Vs, v, ld, b, L, hh, d, tid);
    __syncthreads();

    for (int qt = wave; qt < nkt; qt += 4) {
        const int qrow = qt * 16 + l15;
        const bool qok = qrow < L;
        float qf[4], gf[4], dpart = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = gf[s] = 0.f;
            if (s < ns && qok) {
                qf[s] = q[((int64_t)b * L + qrow) * ld + hh * d + 4 * s + g] * scale;
                const int64_t off = ((int64_t)b * L + qrow) * ldo + hh * d + 4 * s + g;
                gf[s] = dout[off];
                dpart = fmaf(gf[s], o[off], dpart);
            }
        }
        dpart += __shfl_xor(dpart, 16, 64);
        dpart += __shfl_xor(dpart, 32, 64);          // D_q for q = l15
        const float lq = qok ? lse[((int64_t)b * h + hh) * L + qrow] : INFINITY, dqv = dpart;
        f32x4 dqa = {0.f, 0.f, 0.f, 0.f}, dqb = {0.f, 0.f, 0.f, 0.f};   // two accumulation chains
        for (int kt = 0; kt < nkt; ++kt) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const float* ka = Ks + (kt * 16 + l15) * MH_S + g;
            const float* va = Vs + (kt * 16 + l15) * MH_S + g;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < ns) {
                    st = mfma16(ka[4 * s], qf[s], st);     // S^T[key, q]
                    dp = mfma16(va[4 * s], gf[s], dp);     // dP^T[key, q] = V dO^T
                }
            }
            const float* kc = Ks + (kt * 16 + 4 * g) * MH_S + l15;
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float z = st[r];
                if (AFFINE) z = fmaf(z, ksc[key], ksh[key]);
                if (ragged && key >= L) z = -INFINITY;
                const float p = __expf(z - lq);
                ds[r] = p * (dp[r] - dqv);
                if (AFFINE) ds[r] = ds[r] * ksc[key] - cas[key] - st[r] * cbs[key];
            }
            dqa = mfma16(kc[0], ds[0], dqa);               // dQ^T[dd, q] += K^T[dd, key] dS^T[key, q]
            dqb = mfma16(kc[MH_S], ds[1], dqb);
            dqa = mfma16(kc[2 * MH_S], ds[2], dqa);
            dqb = mfma16(kc[3 * MH_S], ds[3], dqb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dqa[r] += dqb[r];
        if (qok && 4 * g < d) {
            float4 ov = make_float4(dqa[0] * scale, dqa[1] * scale, dqa[2] * scale, dqa[3] * scale);
            *reinterpret_cast<float4*>(dq + ((int64_t)b * L + qrow) * ldd + hh * d + 4 * g) = ov;
        }
    }
}

template <bool AFFINE>
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, int64_t ld,
                                                          const float* __restrict__ o, const float* __restrict__ dout,
                                                          int64_t ldo, const float* __restrict__ lse, int L, int h, int d,
                                                          float scale, const float* __restrict__ key_scale,
                                                          const float* __restrict__ key_shift, float* __restrict__ dk,
                                                          float* __restrict__ dv, int64_t ldd,
                                                          const float* __restrict__ corr_a,
                                                          const float* __restrict__ corr_b,
                                                          float* __restrict__ dz_partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nkt = (L + 15) >> 4, L16 = nkt * 16;
    float* Qs = smem;
    float* Gs = Qs + L16 * MH_S;     // dO
    float* lses = Gs + L16 * MH_S;
    float* Dq = lses + L16;
    const bool stats_only = (dk == nullptr);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / h, hh = lid % h;
    const int ns = d >> 2, d4 = d >> 2;

    for (int i = tid; i < 2 * L16 * MH_S; i += 256) smem[i] = 0.f;
    for (int i = tid; i < L16; i += 256) {
        lses[i] = (i < L) ? lse[((int64_t)b * h + hh) * L + i] : INFINITY;   // padded queries -> p = 0
        Dq[i] = 0.f;
    }
    __syncthreads();
    mha_stage(Qs, q, ld, b, L, hh, d, tid);
    for (int i0 = 0; i0 < L * d4; i0 += 256) {     // dO and D_q: d4 consecutive threads share a row
        const int i = i0 + tid;
        float part = 0.f;
        int row = 0;
        if (i < L * d4) {
            row = i / d4;
            const int c = (i % d4) * 4;
            const int64_t off = ((int64_t)b * L + row) * ldo + hh * d + c;
            const float4 gv = *reinterpret_cast<const float4*>(dout + off);
            const float4 ov = *reinterpret_cast<const float4*>(o + off);
            *reinterpret_cast<float4*>(Gs + row * MH_S + c) = gv;
            part = gv.x * ov.x + gv.y * ov.y + gv.z * ov.z + gv.w * ov.w;
        }
        part += __shfl_xor(part, 1, 64);
        if (d4 == 4) part += __shfl_xor(part, 2, 64);
        if (i < L * d4 && (i % d4) == 0) Dq[row] = part;
    }
    __syncthreads();

    for (int kt = wave; kt < nkt; kt += 4) {
        const int krow = kt * 16 + l15;
        const bool kok = krow < L;
        float kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = vf[s] = 0.f;
            if (s < ns && kok) {
                const int64_t off = ((int64_t)b * L + krow) * ld + hh * d + 4 * s + g;
                kf[s] = k[off];
                vf[s] = v[off];
            }
        }
        const float sck = (key_scale && kok) ? key_scale[krow] : 1.f, shk = (key_shift && kok) ? key_shift[krow] : 0.f;
        const float cak = (corr_a && kok) ? corr_a[krow] : 0.f, cbk = (corr_b && kok) ? corr_b[krow] : 0.f;
        f32x4 dka = {0.f, 0.f, 0.f, 0.f}, dva = {0.f, 0.f, 0.f, 0.f};
        float zs = 0.f, zq = 0.f;
        for (int qt = 0; qt < nkt; ++qt) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const float* qa = Qs + (qt * 16 + l15) * MH_S + g;
            const float* ga = Gs + (qt * 16 + l15) * MH_S + g;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < ns) {
                    st = mfma16(qa[4 * s], kf[s], st);     // S[q, key]
                    dp = mfma16(ga[4 * s], vf[s], dp);     // dP[q, key] = dO V^T
                }
            }
            const float* qc = Qs + (qt * 16 + 4 * g) * MH_S + l15;
            const float* gc = Gs + (qt * 16 + 4 * g) * MH_S + l15;
            const float4 lq4v = *reinterpret_cast<const float4*>(lses + qt * 16 + 4 * g);    // this lane's 4 query rows
            const float4 dq4v = *reinterpret_cast<const float4*>(Dq + qt * 16 + 4 * g);
            const float lq4[4] = {lq4v.x, lq4v.y, lq4v.z, lq4v.w}, dq4[4] = {dq4v.x, dq4v.y, dq4v.z, dq4v.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qt * 16 + 4 * g + r;
                const float sraw = st[r] * scale;
                float z = sraw;
                if (AFFINE) z = fmaf(sraw, sck, shk);
                if (!kok) z = -INFINITY;
                const float p = __expf(z - lq4[r]);
                const float dz = p * (dp[r] - dq4[r]);
                if (AFFINE) {
                    zs += dz;
                    zq = fmaf(dz, sraw, zq);
                }
                if (!stats_only) {
                    float ds = dz;
                    if (AFFINE) ds = (qr < L) ? dz * sck - cak - sraw * cbk : 0.f;
                    dva = mfma16(gc[r * MH_S], p, dva);      // dV^T[dd, key] += dO^T[dd, q] P[q, key]
                    dka = mfma16(qc[r * MH_S], ds, dka);     // dK^T[dd, key] += Q^T[dd, q] dS[q, key]
                }
            }
        }
        if (!stats_only && kok && 4 * g < d) {
            const int64_t off = ((int64_t)b * L + krow) * ldd + hh * d + 4 * g;
            *reinterpret_cast<float4*>(dk + off) = make_float4(dka[0] * scale, dka[1] * scale, dka[2] * scale, dka[3] * scale);
            *reinterpret_cast<float4*>(dv + off) = make_float4(dva[0], dva[1], dva[2], dva[3]);
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
}

static inline size_t mha_fwd_lds(int L) { const int L16 = ((L + 15) / 16) * 16; return (size_t)(2 * L16 * MH_S + 2 * L16) * 4; }
static inline size_t mha_stats_lds(int L) { const int L16 = ((L + 15) / 16) * 16; return (size_t)(L16 * MH_S + 8 * L16) * 4; }
static inline size_t mha_bwd_lds(int L) { const int L16 = ((L + 15) / 16) * 16; return (size_t)(2 * L16 * MH_S + 4 * L16) * 4; }

template <typename KernT>
static int reserve_lds(KernT kern, size_t bytes, const char* what) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %zu bytes of LDS", what, bytes);
        return LPM_ERR_LAUNCH;
    }
    return LPM_OK;
}

// The same statistics WITHOUT the L x L logits: over the queries of one (batch, head),  sum_q q.k_j = k_j . s  with  s = sum_q q,  and
// sum_q (q.k_j)^2 = k_j^T G k_j  with  G = sum_q q q^T  [d, d].  One pass over q (the d x d moment matrix: thread (i, j) walks the
// queries staged in LDS, fixed order) and one over k (a thread per key: G k_j, then the two dot products), all fp32 FMAs:
// 2 L d^2 multiply-adds per head instead of L^2 d and the column reductions -- at cfg-3's video stream (80 x 64 heads, L = 300,
// d = 16) 470 us -> the time to read q and k once.
__global__ __launch_bounds__(256) void mha_logit_stats_quad_kernel(const float* __restrict__ q, const float* __restrict__ k, int64_t ld, int L,
                                                                   int h, int d, int nheads, float* __restrict__ partial,
                                                                   float* __restrict__ moments) {
    // one WAVE per (batch, head), four consecutive heads per workgroup.  G = Q^T Q on the exact-fp32 matrix pipe: for a group of four
    // queries lane (l15, g) holds Q[4 s + g][l15], which is its element of BOTH operands of v_mfma_f32_16x16x4_f32 (A = Q^T, B = Q);
    // acc[r] = G[4 g + r][l15].  No staging: a wave-load touches four 64-byte rows.
    __shared__ __attribute__((aligned(16))) float Gs[4][16 * 16 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int head = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (head >= nheads) return;                                  // (no workgroup barriers below: the tiles are wave-private)
    const int b = head / h, hh = head % h;
    const float* qp = q + (int64_t)b * L * ld + hh * d + l15;
    const bool col = l15 < d;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float sacc = 0.f;
    const int ns = (L + 3) >> 2;
    int s = 0;
    for (; s + 4 < ns; s += 5) {                                 // five groups' loads together
        float a[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int row = 4 * (s + u) + g;
            a[u] = (col && row < L) ? qp[(int64_t)row * ld] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            acc = mfma16(a[u], a[u], acc);
            sacc += a[u];
        }
    }
    for (; s < ns; ++s) {
        const int row = 4 * s + g;
        const float a = (col && row < L) ? qp[(int64_t)row * ld] : 0.f;
        acc = mfma16(a, a, acc);
        sacc += a;
    }
    sacc += __shfl_xor(sacc, 16, 64);
    sacc += __shfl_xor(sacc, 32, 64);                            // sum_q Q[q][l15] on every lane
    float* G = Gs[wave];
    float* sv = G + 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) G[(4 * g + r) * 16 + l15] = acc[r];
    if (g == 0) sv[l15] = sacc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (moments) {                                               // [d * d + d] per head: G, then s (lpm_mha_bn_dk_correct reads them in the backward)
        float* mo = moments + (int64_t)head * (d * d + d);
        for (int i = lane; i < d * d; i += 64) mo[i] = G[(i / d) * 16 + i % d];
        if (lane < d) mo[d * d + lane] = sv[lane];
    }
    float* out = partial + (int64_t)head * 2 * L;
    for (int key = lane; key < L; key += 64) {
        float kv[16];
#pragma unroll
        for (int c = 0; c < 16; c += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < d) v = *reinterpret_cast<const float4*>(k + ((int64_t)b * L + key) * ld + hh * d + c);
            kv[c] = v.x; kv[c + 1] = v.y; kv[c + 2] = v.z; kv[c + 3] = v.w;
        }
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const float4 gr = *reinterpret_cast<const float4*>(G + i * 16 + j);
                t = fmaf(gr.x, kv[j], t); t = fmaf(gr.y, kv[j + 1], t); t = fmaf(gr.z, kv[j + 2], t); t = fmaf(gr.w, kv[j + 3], t);
            }
            sq = fmaf(kv[i], t, sq);
            sum = fmaf(kv[i], sv[i], sum);
        }
        out[key] = sum;
        out[L + key] = sq;
    }
}

}  // namespace lpm

#define LPM_MHA_CHECK(name)                                                                                       \
    LPM_REQUIRE(B > 0 && L > 0 && h > 0 && (d == 8 || d == 16) && L <= 512, LPM_ERR_UNSUPPORTED_SHAPE,              \
                name ": need d in {8,16} and L <= 512 (L=%d d=%d)", L, d);                                          \
    LPM_REQUIRE(ld >= (int64_t)h * d && ld % 4 == 0, LPM_ERR_BADARG, name ": bad leading dimension")

extern "C" int lpm_mha_fwd(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d,
                           float scale, const float* key_scale, const float* key_shift, float* o, int64_t ldo, float* lse,
                           lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o && lse, LPM_ERR_BADARG, "lpm_mha_fwd: null pointer");
    LPM_MHA_CHECK("lpm_mha_fwd");
    LPM_REQUIRE((key_scale == nullptr) == (key_shift == nullptr), LPM_ERR_BADARG, "lpm_mha_fwd: key_scale/key_shift go together");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0, LPM_ERR_BADARG, "lpm_mha_fwd: bad ldo");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = mha_fwd_lds(L);
    const int nkt = (L + 15) / 16;
    dim3 grid(B * h);
#define LPM_MHA_FWD1(N, AFF)                                                                                     \
    do {                                                                                                         \
        auto kern = mha_fwd_kernel<N, AFF>;                                                                      \
        if (int rc = reserve_lds(kern, lds, "lpm_mha_fwd")) return rc;                                           \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q, k, v, ld, L, h, d, scale, key_scale, key_shift, o, ldo, lse); \
    } while (0)
#define LPM_MHA_FWD(N)                 \
    do {                               \
        if (key_scale) LPM_MHA_FWD1(N, true); \
        else LPM_MHA_FWD1(N, false);   \
    } while (0)
    if (nkt <= 4) LPM_MHA_FWD(4);
    else if (nkt <= 8) LPM_MHA_FWD(8);
    else if (nkt <= 16) LPM_MHA_FWD(16);
    else if (nkt <= 20) LPM_MHA_FWD(20);
    else LPM_MHA_FWD(32);
#undef LPM_MHA_FWD
#undef LPM_MHA_FWD1
    return check_launch("lpm_mha_fwd");
}

extern "C" size_t lpm_mha_logit_stats_workspace_bytes(int B, int L, int h) { return (size_t)B * h * 2 * L * sizeof(float); }

extern "C" int lpm_mha_logit_stats(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float* partial,
                                   lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && partial, LPM_ERR_BADARG, "lpm_mha_logit_stats: null pointer");
    LPM_MHA_CHECK("lpm_mha_logit_stats");
    hipStream_t s = (hipStream_t)stream;
    static const int quad = [] { const char* e = getenv("LPM_MHA_STATS_QUAD"); return (e && e[0] == '0') ? 0 : 1; }();   // 0: through the logits (A/B)
    if (quad && (((uintptr_t)q | (uintptr_t)k) & 15) == 0 && d % 4 == 0) {
        hipLaunchKernelGGL(mha_logit_stats_quad_kernel, dim3((B * h + 3) / 4), dim3(256), 0, s, q, k, ld, L, h, d, B * h, partial,
                           (float*)nullptr);
        return check_launch("lpm_mha_logit_stats");
    }
    const size_t lds = mha_stats_lds(L);
    const int nkt = (L + 15) / 16;
    dim3 grid(B * h);
#define LPM_MHA_ST(N)                                                                                   \
    do {                                                                                                \
        auto kern = mha_logit_stats_kernel<N>;                                                          \
        if (int rc = reserve_lds(kern, lds, "lpm_mha_logit_stats")) return rc;                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q, k, ld, L, h, d, partial);                  \
    } while (0)
    if (nkt <= 4) LPM_MHA_ST(4);
    else if (nkt <= 8) LPM_MHA_ST(8);
    else if (nkt <= 16) LPM_MHA_ST(16);
    else if (nkt <= 20) LPM_MHA_ST(20);
    else LPM_MHA_ST(32);
#undef LPM_MHA_ST
    return check_launch("lpm_mha_logit_stats");
}

extern "C" int lpm_mha_logit_stats_moments(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float* partial, float* moments,
                                           lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && partial, LPM_ERR_BADARG, "lpm_mha_logit_stats_moments: null pointer");
    LPM_MHA_CHECK("lpm_mha_logit_stats_moments");
    LPM_REQUIRE((((uintptr_t)q | (uintptr_t)k) & 15) == 0 && d % 4 == 0, LPM_ERR_BADARG,
                "lpm_mha_logit_stats_moments: q, k must be 16-byte aligned and d a multiple of 4");
    hipLaunchKernelGGL(mha_logit_stats_quad_kernel, dim3((B * h + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, k, ld, L, h, d, B * h, partial,
                       moments);
    return check_launch("lpm_mha_logit_stats_moments");
}

extern "C" int lpm_mha_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout,
                           int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, const float* key_scale,
                           const float* key_shift, float* dq, float* dk, float* dv, int64_t ldd, const float* corr_a,
                           const float* corr_b, float* dz_partial, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && k && v && o && dout && lse, LPM_ERR_BADARG, "lpm_mha_bwd: null pointer");
    LPM_REQUIRE((dq && dk && dv) || (!dq && !dk && !dv && dz_partial), LPM_ERR_BADARG,
                "lpm_mha_bwd: give dq, dk, dv together, or none of them (statistics-only pass needs dz_partial)");
    LPM_REQUIRE((corr_a == nullptr) == (corr_b == nullptr), LPM_ERR_BADARG, "lpm_mha_bwd: corr_a/corr_b go together");
    LPM_MHA_CHECK("lpm_mha_bwd");
    LPM_REQUIRE((key_scale == nullptr) == (key_shift == nullptr), LPM_ERR_BADARG, "lpm_mha_bwd: key_scale/key_shift go together");
    LPM_REQUIRE(ldo >= (int64_t)h * d && ldo % 4 == 0 && ldd >= (int64_t)h * d && ldd % 4 == 0, LPM_ERR_BADARG, "lpm_mha_bwd: bad ldo/ldd");
    const size_t lds = mha_bwd_lds(L);
    hipStream_t s = (hipStream_t)stream;
#define LPM_MHA_BWD(AFF)                                                                                                  \
    do {                                                                                                                  \
        if (dq) {                                                                                                         \
            if (int rc = reserve_lds(mha_bwd_dq_kernel<AFF>, lds, "lpm_mha_bwd")) return rc;                              \
            hipLaunchKernelGGL(mha_bwd_dq_kernel<AFF>, dim3(B * h), dim3(256), lds, s, q, k, v, ld, o, dout, ldo, lse, L, h, d, \
                               scale, key_scale, key_shift, dq, ldd, corr_a, corr_b);                                     \
        }                                                                                                                 \
        if (int rc = reserve_lds(mha_bwd_dkv_kernel<AFF>, lds, "lpm_mha_bwd")) return rc;                                 \
        hipLaunchKernelGGL(mha_bwd_dkv_kernel<AFF>, dim3(B * h), dim3(256), lds, s, q, k, v, ld, o, dout, ldo, lse, L, h, d,   \
                           scale, key_scale, key_shift, dk, dv, ldd, corr_a, corr_b, dz_partial);                         \
    } while (0)
    if (key_scale) LPM_MHA_BWD(true);
    else LPM_MHA_BWD(false);
#undef LPM_MHA_BWD
    return check_launch("lpm_mha_bwd");
}
