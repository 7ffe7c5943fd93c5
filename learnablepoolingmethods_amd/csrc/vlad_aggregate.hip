// K2 -- fused [BN-affine -> softmax] -> residual aggregation -> intra-normalisation, and the
// global-L2 finalize pass.
//
//   a[t,k]   = softmax_k(assign[t,k]*scale[k] + shift[k])        frame_level_models.py:2783-2798
//   U[d,k]   = sum_t a[t,k] * x[t,d]  -  (sum_t a[t,k]) * W2[d,k]               :2803-2817
//   N[:,k]   = U[:,k] * rsqrt(max(|U[:,k]|^2, 1e-12))                           :2819
//   out      = N * rsqrt(max(sum N^2, 1e-12))   (lpm_vlad_finalize_fwd)         :2821-2822
// and, with the SOFTMAX flag off, the NetVladAttenCluster form (video_pooling_modules.py:1646-1658).
//
// gfx950 mapping.  One 256-thread workgroup (4 waves, one per SIMD) owns (clip b, 32-cluster slab):
// the whole 32 x D tile of U lives in MFMA accumulators (wave w holds columns [w*D/4, (w+1)*D/4),
// D/128 tiles of v_mfma_f32_32x32x2_f32 = up to 128 accumulator VGPRs per lane), so the reduction over
// D for the intra-norm is workgroup-local and the [B,K,D] tensor is written exactly once.  Frames
// stream in chunks of 8 through a register-staged double-buffered LDS pipeline: x rows as float4
// (one full 4 KB row per wave-quad instruction), the logits row-wise (each wave owns 2 rows of the
// chunk, softmax by wavefront reductions, only the slab's 32 columns are kept).  The softmax is
// recomputed by each of the K/32 slab workgroups of a clip: 8x the exp work, but no [B,T,K]
// assignment tensor ever reaches HBM; logits[b] and x[b] are re-read from the XCD's L2 because the
// slabs of one clip are mapped to consecutive workgroups of ONE XCD (xcd_remap).
// Algorithmic HBM bytes per clip: 4*(T*K + T*D + D*K) + D*K*4/B.
#include "lpm_common.h"

namespace lpm {

constexpr int VA_TC = 8;      // frames per LDS stage
constexpr int VA_SLAB = 32;   // clusters per workgroup

template <int NT, int KPL, bool SOFTMAX>
__global__ __launch_bounds__(256, 2) void vlad_aggregate_kernel(
    const float* __restrict__ assign, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ centres, int B, int T, int K, int nslab,
    int residual, float* __restrict__ nrm, float* __restrict__ asum, float* __restrict__ colsq,
    float* __restrict__ csq) {
    constexpr int D = NT * 128;
    constexpr int DW = NT * 32;  // columns of d per wave
    __shared__ __attribute__((aligned(16))) float xs[2][VA_TC * D];
    __shared__ float as[2][VA_TC * VA_SLAB];
    __shared__ float sred[4][VA_SLAB];
    __shared__ float ssum[VA_SLAB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / nslab, slab = lid % nslab;
    const int k0 = slab * VA_SLAB;

    const float* xb = x + (int64_t)b * T * ldx;
    const float* ab = assign + (int64_t)b * T * K;

    // per-thread affine for the columns this lane owns in a logits row: c = lane + 64*j
    float sc[KPL], sh[KPL];
    if (SOFTMAX) {
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int c = lane + 64 * j;
            sc[j] = (scale && c < K) ? scale[c] : 1.f;
            sh[j] = (shift && c < K) ? shift[c] : 0.f;
        }
    }
    const int jsel = slab >> 1;          // which 64-column group holds this slab
    const int hsel = slab & 1;           // which half-wave of that group

    float4 xreg[NT];
    float lreg[2][KPL];

    auto gload = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = tid + i * 256;
            const int r = f / (D / 4), c4 = (f % (D / 4)) * 4;
            const int t = t0 + r;
            xreg[i] = (t < T) ? *reinterpret_cast<const float4*>(xb + (int64_t)t * ldx + c4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (SOFTMAX) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int t = t0 + wave * 2 + rr;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int c = lane + 64 * j;
                    lreg[rr][j] = (t < T && c < K) ? ab[(int64_t)t * K + c] : 0.f;
                }
            }
        } else {
            const int t = t0 + wave * 2 + half;
            const int c = k0 + l31;
            lreg[0][0] = (t < T && c < K) ? ab[(int64_t)t * K + c] : 0.f;
        }
    };
    auto sstore = [&](int buf, int t0) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = tid + i * 256;
            *reinterpret_cast<float4*>(&xs[buf][f * 4]) = xreg[i];
        }
        if (SOFTMAX) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int rl = wave * 2 + rr;
                const int t = t0 + rl;
                float v[KPL];
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int c = lane + 64 * j;
                    v[j] = (c < K) ? fmaf(lreg[rr][j], sc[j], sh[j]) : -INFINITY;
                    m = fmaxf(m, v[j]);
                }
                m = wave_max(m);
                float e[KPL], sum = 0.f;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    e[j] = __expf(v[j] - m);
                    sum += e[j];
                }
                sum = wave_sum(sum);
                const float inv = 1.f / sum;
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < KPL; ++j) mine = (j == jsel) ? e[j] : mine;
                if (half == hsel) as[buf][rl * VA_SLAB + l31] = (t < T) ? mine * inv : 0.f;
            }
        } else {
            as[buf][(wave * 2 + half) * VA_SLAB + l31] = lreg[0][0];
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float asum_l = 0.f;

    const int nchunk = (T + VA_TC - 1) / VA_TC;
    gload(0);
    sstore(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload((c + 1) * VA_TC);
        const float* ap = &as[buf][half * VA_SLAB + l31];
        const float* xp = &xs[buf][half * D + wave * DW + l31];
#pragma unroll
        for (int tt = 0; tt < VA_TC; tt += 2) {
            const float a = ap[tt * VA_SLAB];
            asum_l += a;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(a, xp[tt * D + t * 32], acc[t]);
        }
        if (c + 1 < nchunk) sstore(buf ^ 1, (c + 1) * VA_TC);
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------------
    asum_l += __shfl_xor(asum_l, 32, 64);        // sum_t a[t, k0 + l31]
    if (wave == 0 && half == 0) ssum[l31] = asum_l;
    __syncthreads();

    const int dbase = wave * DW + l31;           // this lane's d inside tile t: dbase + 32*t
    float part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int d = dbase + t * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = 8 * q + 4 * half;     // slab-local cluster of reg 4q (+0..3)
            float4 w2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (residual && (k0 + kr) < K)
                w2 = *reinterpret_cast<const float4*>(centres + (int64_t)d * K + k0 + kr);
            const float w2v[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * q + j;
                const float u = acc[t][r] - ssum[kr + j] * w2v[j];
                acc[t][r] = u;
                part[r] += u * u;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = half_sum(part[r]);
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sred[wave][mfma32_row(r, lane)] = part[r];
    }
    __syncthreads();
    float inv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mfma32_row(r, lane);
        const float n = sred[0][row] + sred[1][row] + sred[2][row] + sred[3][row];
        inv[r] = rsqrtf(fmaxf(n, kL2Eps));
    }
    if (tid < VA_SLAB && (k0 + tid) < K) {
        const float n = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        const int64_t o = (int64_t)b * K + k0 + tid;
        asum[o] = ssum[tid];
        colsq[o] = n;
        csq[o] = n * iv * iv;
    }
    float* ob = nrm + (int64_t)b * D * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int d = dbase + t * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = 8 * q + 4 * half;
            if ((k0 + kr) < K) {
                float4 o;
                o.x = acc[t][4 * q + 0] * inv[4 * q + 0];
                o.y = acc[t][4 * q + 1] * inv[4 * q + 1];
                o.z = acc[t][4 * q + 2] * inv[4 * q + 2];
                o.w = acc[t][4 * q + 3] * inv[4 * q + 3];
                *reinterpret_cast<float4*>(ob + (int64_t)d * K + k0 + kr) = o;
            }
        }
    }
}

// gsq[b] = sum_k csq[b,k];  out = nrm * rsqrt(max(gsq, eps)) in d-major [B, D*K] or k-major [B,K,D].
// grid = (B, D/32): each block handles 32 d-rows of one clip through an LDS transpose tile.
template <bool KMAJOR>
__global__ __launch_bounds__(256) void vlad_finalize_kernel(const float* __restrict__ nrm,
                                                            const float* __restrict__ csq, int D, int K,
                                                            float* __restrict__ out, float* __restrict__ gsq) {
    __shared__ float tile[32][33];
    __shared__ float sg;
    const int b = blockIdx.x, d0 = blockIdx.y * 32, tid = threadIdx.x;
    float g = 0.f;
    for (int k = tid; k < K; k += 256) g += csq[(int64_t)b * K + k];
    g = wave_sum(g);
    __shared__ float wg[4];
    if ((tid & 63) == 0) wg[tid >> 6] = g;
    __syncthreads();
    if (tid == 0) {
        const float tot = wg[0] + wg[1] + wg[2] + wg[3];
        sg = rsqrtf(fmaxf(tot, kL2Eps));
        if (blockIdx.y == 0 && gsq) gsq[b] = tot;
    }
    __syncthreads();
    const float ig = sg;
    const float* src = nrm + ((int64_t)b * D + d0) * K;
    if (!KMAJOR) {
        float* dst = out + ((int64_t)b * D + d0) * K;
        const int n4 = 32 * K / 4;
        for (int i = tid; i < n4; i += 256) {
            float4 v = reinterpret_cast<const float4*>(src)[i];
            v.x *= ig; v.y *= ig; v.z *= ig; v.w *= ig;
            reinterpret_cast<float4*>(dst)[i] = v;
        }
    } else {
        float* dst = out + (int64_t)b * K * D;
        const int tx = tid & 31, ty = tid >> 5;  // 32 x 8
        for (int kb = 0; kb < K; kb += 32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dl = ty + 8 * i, k = kb + tx;
                tile[dl][tx] = (k < K) ? src[(int64_t)dl * K + k] * ig : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kl = ty + 8 * i, k = kb + kl;
                if (k < K) dst[(int64_t)k * D + d0 + tx] = tile[tx][kl];
            }
            __syncthreads();
        }
    }
}

template <int NT, bool SM>
static int launch_aggregate(int kpl, dim3 grid, hipStream_t s, const float* assign, const float* scale,
                            const float* shift, const float* x, int64_t ldx, const float* centres, int B, int T,
                            int K, int nslab, int residual, float* nrm, float* asum, float* colsq, float* csq) {
#define LPM_VA_LAUNCH(KPL)                                                                                        \
    hipLaunchKernelGGL((vlad_aggregate_kernel<NT, KPL, SM>), grid, dim3(256), 0, s, assign, scale, shift, x, ldx, \
                       centres, B, T, K, nslab, residual, nrm, asum, colsq, csq)
    if (!SM || kpl <= 1) LPM_VA_LAUNCH(1);
    else if (kpl <= 2) LPM_VA_LAUNCH(2);
    else if (kpl <= 4) LPM_VA_LAUNCH(4);
    else if (kpl <= 8) LPM_VA_LAUNCH(8);
    else LPM_VA_LAUNCH(16);
#undef LPM_VA_LAUNCH
    return check_launch("lpm_vlad_aggregate_fwd");
}

}  // namespace lpm

extern "C" int lpm_vlad_aggregate_fwd(const float* assign, const float* scale, const float* shift, const float* x,
                                      int64_t ldx, const float* centres, int B, int T, int D, int K, int flags,
                                      float* nrm, float* asum, float* colsq, float* csq, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(assign && x && nrm && asum && colsq && csq, LPM_ERR_BADARG, "lpm_vlad_aggregate_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_fwd: RESIDUAL needs centres");
    LPM_REQUIRE(B > 0 && T > 0 && D > 0 && K > 0 && ldx >= D, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_fwd: bad sizes B=%d T=%d D=%d K=%d ldx=%lld", B, T, D, K, (long long)ldx);
    LPM_REQUIRE((D == 128 || D == 256 || D == 512 || D == 1024) && K % 4 == 0 && K <= 1024 && ldx % 4 == 0,
                LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_fwd: need D in {128,256,512,1024}, K %% 4 == 0, K <= 1024, ldx %% 4 == 0 (D=%d K=%d)",
                D, K);
    LPM_REQUIRE((((uintptr_t)x | (uintptr_t)centres | (uintptr_t)nrm) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_fwd: x/centres/nrm must be 16-byte aligned");
    const int nslab = (K + VA_SLAB - 1) / VA_SLAB;
    const int kpl = (K + 63) / 64;
    dim3 grid(B * nslab);
    hipStream_t s = (hipStream_t)stream;
    const bool sm = (flags & LPM_VLAD_SOFTMAX) != 0;
#define LPM_VA_DISPATCH(NT)                                                                                         \
    return sm ? launch_aggregate<NT, true>(kpl, grid, s, assign, scale, shift, x, ldx, centres, B, T, K, nslab,     \
                                           residual, nrm, asum, colsq, csq)                                         \
              : launch_aggregate<NT, false>(kpl, grid, s, assign, scale, shift, x, ldx, centres, B, T, K, nslab,    \
                                            residual, nrm, asum, colsq, csq)
    switch (D) {
        case 128: LPM_VA_DISPATCH(1);
        case 256: LPM_VA_DISPATCH(2);
        case 512: LPM_VA_DISPATCH(4);
        default: LPM_VA_DISPATCH(8);
    }
#undef LPM_VA_DISPATCH
}

extern "C" int lpm_vlad_finalize_fwd(const float* nrm, const float* csq, int B, int D, int K, int flags, float* out,
                                     float* gsq, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(nrm && csq && out, LPM_ERR_BADARG, "lpm_vlad_finalize_fwd: null pointer");
    LPM_REQUIRE(B > 0 && D > 0 && K > 0 && D % 32 == 0 && K % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_finalize_fwd: need D %% 32 == 0 and K %% 4 == 0 (D=%d K=%d)", D, K);
    dim3 grid(B, D / 32);
    if (flags & LPM_VLAD_OUT_KMAJOR)
        hipLaunchKernelGGL(vlad_finalize_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, nrm, csq, D, K, out, gsq);
    else
        hipLaunchKernelGGL(vlad_finalize_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, nrm, csq, D, K, out, gsq);
    return check_launch("lpm_vlad_finalize_fwd");
}
