// K3 -- backward of K2 (the reference has no backward code: TF autodiff of
// frame_level_models.py:2798-2822 / video_pooling_modules.py:1646-1658; closed forms in SURVEY.md
// App. F.1-F.3, restated and checked against autograd in oracle/numpy_ref.py).
//
// With N the intra-normalised descriptor saved by K2 and per-(clip,cluster) scalars
//   p = <dO_k, N_k>, alpha = <dO, O>:    dU[:,k] = u_k * dO[:,k] - v_k * N[:,k]
//   u_k = inv_n_k * inv_g,   v_k = u_k * (m_g * alpha * inv_g * (1 - m_k c_k) + m_k p_k)
// (m_* = "clamp inactive").  Then
//   dA[t,k] = sum_d x[t,d] dU[d,k] - ctil_k,  ctil_k = sum_d dU[d,k] W2[d,k]
//   dx[t,d] = sum_k A[t,k] dU[d,k]
//   dW2[d,k] = - sum_b s_b[k] dU_b[d,k]
//   dlogit~[t,k] = A (dA - sum_j A_j dA_j)                      (softmax, F.3)
//
// Launch sequence: (0) optional k-major -> d-major transpose of dO; (1) column dots, split 4-way over
// D for parallelism; (2) coefficients; (3) main kernel, one workgroup per (clip, 32-frame slab): both
// GEMMs share one pass over dU, which is formed on the fly from dO and N while staging 32-row d-chunks
// into LDS (dU never reaches HBM), v_mfma_f32_32x32x2_f32 accumulation, the softmax backward is done
// on the workgroup's 32 x K tile in LDS; (4) dW2 as a clip-loop per (d,k) float4.
#include "tile_gemm.h"

namespace lpm {

constexpr int VB_TS = 32;     // frames per workgroup in the main kernel
constexpr int VB_DC = 32;     // d rows per staged chunk
constexpr int VB_DSPLIT = 16; // column-dot split over D (B x 16 workgroups keep enough loads in flight: 58 -> see DESIGN)

// [B,K,D] -> [B,D,K]
__global__ __launch_bounds__(256) void vlad_kmajor_to_dmajor_kernel(const float* __restrict__ src, int D, int K,
                                                                    float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, d0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* s = src + (int64_t)b * K * D;
    float* o = dst + (int64_t)b * D * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i;
        tile[ty + 8 * i][tx] = (k < K) ? s[(int64_t)k * D + d0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = d0 + ty + 8 * i, k = k0 + tx;
        if (k < K) o[(int64_t)d * K + k] = tile[tx][ty + 8 * i];
    }
}

// [B,K,D] -> [B,D,K] for D % 64 == 0 and K % 64 == 0: 64 x 64 tiles, 16-byte global accesses on both sides; the LDS row
// stride 65 keeps the scalar tile writes (bank = k + 4 q + j) and the transposed reads (bank = 4 kq + j + d) conflict-free.
__global__ __launch_bounds__(256) void vlad_kmajor_to_dmajor64_kernel(const float* __restrict__ src, int D, int K,
                                                                      float* __restrict__ dst) {
    __shared__ float tile[64 * 65];
    const int b = blockIdx.z, d0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const float* s = src + (int64_t)b * K * D;
    float* o = dst + (int64_t)b * D * K;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + 256 * it, k = i >> 4, q = i & 15;
        const float4 v = *reinterpret_cast<const float4*>(s + (int64_t)(k0 + k) * D + d0 + 4 * q);
        float* t = tile + k * 65 + 4 * q;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + 256 * it, d = i >> 4, kq = i & 15;
        const float* t = tile + (4 * kq) * 65 + d;
        *reinterpret_cast<float4*>(o + (int64_t)(d0 + d) * K + k0 + 4 * kq) = make_float4(t[0], t[65], t[130], t[195]);
    }
}

static void launch_kmajor_to_dmajor(const float* src, int B, int D, int K, float* dst, hipStream_t s) {
    if (D % 64 == 0 && K % 64 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
        hipLaunchKernelGGL(vlad_kmajor_to_dmajor64_kernel, dim3(D / 64, K / 64, B), dim3(256), 0, s, src, D, K, dst);
    else
        hipLaunchKernelGGL(vlad_kmajor_to_dmajor_kernel, dim3(D / 32, (K + 31) / 32, B), dim3(256), 0, s, src, D, K, dst);
}

// bf16 storage: N points at bf16 values (the un-normalised sums as lpm_vlad_aggregate_tiles3_fwd_bf16 stores them)
__device__ __forceinline__ float vb_ld(const float* p, int64_t i, int bf16) {
    return bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16) : p[i];
}
__device__ __forceinline__ float4 vb_ld4(const float* p, int64_t i4, int bf16) {       // i4: index in units of 4 elements
    if (!bf16) return reinterpret_cast<const float4*>(p)[i4];
    const uint2 q = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                       __uint_as_float(q.y & 0xffff0000u));
}

// dots[b][split][0..2][k]: <dO_k,N_k>, <dO_k,W2_k>, <N_k,W2_k> over the split's rows of D.
// Threads are (row group, 4 consecutive clusters): a row of K clusters is K / 4 16-byte loads per array (8-byte for bf16 N), 256 / (K / 4)
// row groups walk the split's rows side by side with four rows' loads issued together, and meet through LDS in a fixed order.  (One
// thread per cluster with one 4-byte load per array in flight: 202 us for the 400 MB of cfg-5's video stream, 2 TB/s.)
__global__ __launch_bounds__(256) void vlad_bwd_coldots_kernel(const float* __restrict__ dO,
                                                               const float* __restrict__ N,
                                                               const float* __restrict__ W2, int D, int K,
                                                               float* __restrict__ dots, const float* __restrict__ colsq_raw, int n_bf16,
                                                               int64_t dob) {
    // colsq_raw != NULL (LPM_VLAD_NRM_RAW): N holds the un-normalised sums U and N = U * rsqrt(max(colsq, eps)) per column --
    // the product vlad_finalize2 would have stored.  dob: distance between the clips' gradients in dO, in elements (>= D * K)
    __shared__ float4 red[3][256];
    const int b = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
    // split sp takes rows sp, sp + VB_DSPLIT, ...: the splits of a clip read VB_DSPLIT consecutive rows at a time (contiguous
    // quarter-ranges put every workgroup of the grid at the same offset of a 64 KB-aligned range: HBM channel aliasing)
    const int K4 = K >> 2;
    const int RG = (K4 < 256 && 256 % K4 == 0) ? 256 / K4 : 1;
    const int rg = RG > 1 ? tid / K4 : 0;
    const int dper = D / VB_DSPLIT;
    float* out = dots + ((int64_t)b * VB_DSPLIT + sp) * 3 * K;
    for (int c4 = RG > 1 ? tid % K4 : tid; c4 < K4; c4 += (RG > 1 ? K4 : 256)) {        // RG > 1: exactly one pass, every thread in it
        float4 iv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (colsq_raw) {
            const float4 cq = *reinterpret_cast<const float4*>(colsq_raw + (int64_t)b * K + 4 * c4);
            iv = make_float4(rsqrtf(fmaxf(cq.x, kL2Eps)), rsqrtf(fmaxf(cq.y, kL2Eps)), rsqrtf(fmaxf(cq.z, kL2Eps)), rsqrtf(fmaxf(cq.w, kL2Eps)));
        }
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), dw = p, nw = p;
        auto add = [&](const float4 a, float4 n, const float4 w) {
            n.x *= iv.x; n.y *= iv.y; n.z *= iv.z; n.w *= iv.w;
            p.x = fmaf(a.x, n.x, p.x); p.y = fmaf(a.y, n.y, p.y); p.z = fmaf(a.z, n.z, p.z); p.w = fmaf(a.w, n.w, p.w);
            dw.x = fmaf(a.x, w.x, dw.x); dw.y = fmaf(a.y, w.y, dw.y); dw.z = fmaf(a.z, w.z, dw.z); dw.w = fmaf(a.w, w.w, dw.w);
            nw.x = fmaf(n.x, w.x, nw.x); nw.y = fmaf(n.y, w.y, nw.y); nw.z = fmaf(n.z, w.z, nw.z); nw.w = fmaf(n.w, w.w, nw.w);
        };
        // row d of the split = row sp + d * VB_DSPLIT of the clip; in units of float4: ((b * D + sp + d * VB_DSPLIT) * K4 + c4)
        const int64_t base4 = ((int64_t)b * D + sp) * K4 + c4;              // N (contiguous clips)
        const int64_t obase4 = (int64_t)b * (dob >> 2) + (int64_t)sp * K4 + c4;  // dO
        const int64_t rs4 = (int64_t)VB_DSPLIT * K4;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        int d = rg;
        for (; d + 3 * RG < dper; d += 4 * RG) {
            float4 a[4], n[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i4 = base4 + (int64_t)(d + u * RG) * rs4;
                a[u] = reinterpret_cast<const float4*>(dO)[obase4 + (int64_t)(d + u * RG) * rs4];
                n[u] = vb_ld4(N, i4, n_bf16);
                w[u] = W2 ? reinterpret_cast<const float4*>(W2)[(int64_t)(sp + (d + u * RG) * VB_DSPLIT) * K4 + c4] : zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(a[u], n[u], w[u]);
        }
        for (; d < dper; d += RG) {
            const int64_t i4 = base4 + (int64_t)d * rs4;
            add(reinterpret_cast<const float4*>(dO)[obase4 + (int64_t)d * rs4], vb_ld4(N, i4, n_bf16),
                W2 ? reinterpret_cast<const float4*>(W2)[(int64_t)(sp + d * VB_DSPLIT) * K4 + c4] : zero);
        }
        if (RG > 1) {
            red[0][tid] = p; red[1][tid] = dw; red[2][tid] = nw;
            __syncthreads();
            if (rg == 0)
                for (int i = 1; i < RG; ++i) {
                    const float4 x0 = red[0][i * K4 + c4], x1 = red[1][i * K4 + c4], x2 = red[2][i * K4 + c4];
                    p.x += x0.x; p.y += x0.y; p.z += x0.z; p.w += x0.w;
                    dw.x += x1.x; dw.y += x1.y; dw.z += x1.z; dw.w += x1.w;
                    nw.x += x2.x; nw.y += x2.y; nw.z += x2.z; nw.w += x2.w;
                }
        }
        if (rg == 0) {
            *reinterpret_cast<float4*>(out + 4 * c4) = p;
            *reinterpret_cast<float4*>(out + K + 4 * c4) = dw;
            *reinterpret_cast<float4*>(out + 2 * K + 4 * c4) = nw;
        }
    }
}

// per clip: alpha, then u, v, ctil per cluster
__global__ __launch_bounds__(256) void vlad_bwd_coeff_kernel(const float* __restrict__ dots,
                                                             const float* __restrict__ colsq,
                                                             const float* __restrict__ csq,
                                                             const float* __restrict__ gsq, int K,
                                                             float* __restrict__ u, float* __restrict__ v,
                                                             float* __restrict__ ctil, int nsplit) {
    __shared__ float wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* dp = dots + (int64_t)b * nsplit * 3 * K;
    float psum = 0.f;
    for (int k = tid; k < K; k += 256) {
        float p = 0.f;
        for (int s = 0; s < nsplit; ++s) p += dp[s * 3 * K + k];
        psum += p;
    }
    psum = wave_sum(psum);
    if ((tid & 63) == 0) wsum[tid >> 6] = psum;
    __syncthreads();
    const float g = gsq[b];
    const float inv_g = rsqrtf(fmaxf(g, kL2Eps));
    const float m_g = (g >= kL2Eps) ? 1.f : 0.f;
    const float alpha = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_g;   // <dO, O>
    for (int k = tid; k < K; k += 256) {
        float p = 0.f, dw = 0.f, nw = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            p += dp[s * 3 * K + k];
            dw += dp[s * 3 * K + K + k];
            nw += dp[s * 3 * K + 2 * K + k];
        }
        const float n = colsq[(int64_t)b * K + k], c = csq[(int64_t)b * K + k];
        const float m_k = (n >= kL2Eps) ? 1.f : 0.f;
        const float uu = rsqrtf(fmaxf(n, kL2Eps)) * inv_g;
        const float vv = uu * (m_g * alpha * inv_g * (1.f - m_k * c) + m_k * p);
        u[(int64_t)b * K + k] = uu;
        v[(int64_t)b * K + k] = vv;
        ctil[(int64_t)b * K + k] = uu * dw - vv * nw;
    }
}

// main backward kernel: workgroup = (clip, 32-frame slab).  KTW = k-tiles (32 clusters) per wave.
template <int KTW, bool SOFTMAX>
__global__ __launch_bounds__(256) void vlad_bwd_main_kernel(
    const float* __restrict__ dO, const float* __restrict__ N, const float* __restrict__ ug,
    const float* __restrict__ vg, const float* __restrict__ cg, const float* __restrict__ assign,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ x, int64_t ldx,
    int T, int D, int K, int nts, float* __restrict__ dassign, float* __restrict__ dx, int64_t lddx,
    int accumulate) {
    constexpr int KPL = 2 * KTW;          // logits columns per lane in a row pass
    constexpr int NP = 4 * KTW;           // float4 staging passes per d-chunk (32 rows x K/4 float4 / 256 thr)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KT = (K + 31) / 32;
    const int KS = KT * 32 + 1;           // row stride: K rounded up to whole 32-cluster tiles (+1: column-wise
                                          // MFMA operand reads hit distinct banks); pad columns hold zeros
    float* As = smem;                     // [32][KS]  assignment tile
    float* dUs = As + VB_TS * KS;         // [32][KS]  dU chunk; later dA - ctil
    float* xs = dUs + VB_DC * KS;         // [32][33]  x chunk
    float* red = xs + 32 * 33;            // [4][32][33] per-wave partial dx tiles
    float* cu = red + 4 * 32 * 33;        // [K] u, [K] v, [K] ctil
    float* cv = cu + K;
    float* cc = cv + K;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / nts, t0 = (lid % nts) * VB_TS;
    const int K4 = K / 4;

    for (int i = tid; i < 2 * VB_TS * KS; i += 256) smem[i] = 0.f;   // As and dUs, including the pad columns
    __syncthreads();

    for (int k = tid; k < K; k += 256) {
        cu[k] = ug[(int64_t)b * K + k];
        cv[k] = vg[(int64_t)b * K + k];
        cc[k] = cg[(int64_t)b * K + k];
    }
    // ---- prologue: assignment tile (softmax recomputed from the logits) ----------------------
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, t = t0 + row;
        const float* ar = assign + ((int64_t)b * T + t) * K;
        float vals[KPL];
        if (SOFTMAX) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int c = lane + 64 * j;
                float lv = -INFINITY;
                if (t < T && c < K) lv = fmaf(ar[c], scale ? scale[c] : 1.f, shift ? shift[c] : 0.f);
                vals[j] = lv;
                m = fmaxf(m, lv);
            }
            m = wave_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                vals[j] = (t < T) ? __expf(vals[j] - m) : 0.f;
                sum += vals[j];
            }
            sum = wave_sum(sum);
            const float inv = (t < T) ? 1.f / sum : 0.f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int c = lane + 64 * j;
                if (c < K) As[row * KS + c] = vals[j] * inv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int c = lane + 64 * j;
                if (c < K) As[row * KS + c] = (t < T) ? ar[c] : 0.f;
            }
        }
    }

    // ---- staging helpers -----------------------------------------------------------------------
    float4 rdo[NP], rn[NP], rx;
    const float* dOb = dO + (int64_t)b * D * K;
    const float* Nb = N + (int64_t)b * D * K;
    const int xt = tid >> 3, xc = (tid & 7) * 4;
    const bool xok = (t0 + xt) < T;
    const float* xrow = x + ((int64_t)b * T + t0 + xt) * ldx + xc;
    auto gload = [&](int d0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = tid + 256 * i;
            const int r = f / K4;
            if (r < VB_DC) {
                const int64_t off = (int64_t)(d0 + r) * K + (f % K4) * 4;
                rdo[i] = *reinterpret_cast<const float4*>(dOb + off);
                rn[i] = *reinterpret_cast<const float4*>(Nb + off);
            }
        }
        rx = xok ? *reinterpret_cast<const float4*>(xrow + d0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = tid + 256 * i;
            const int r = f / K4;
            if (r < VB_DC) {
                const int k = (f % K4) * 4;
                float* dst = dUs + r * KS + k;
                dst[0] = cu[k + 0] * rdo[i].x - cv[k + 0] * rn[i].x;
                dst[1] = cu[k + 1] * rdo[i].y - cv[k + 1] * rn[i].y;
                dst[2] = cu[k + 2] * rdo[i].z - cv[k + 2] * rn[i].z;
                dst[3] = cu[k + 3] * rdo[i].w - cv[k + 3] * rn[i].w;
            }
        }
        float* xd = xs + xt * 33 + xc;
        xd[0] = rx.x; xd[1] = rx.y; xd[2] = rx.z; xd[3] = rx.w;
    };

    f32x16 accA[KTW];
#pragma unroll
    for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[i][r] = 0.f;

    const int nchunk = D / VB_DC;
    gload(0);
    __syncthreads();          // cu/cv visible, As complete
    sstore();
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int d0 = c * VB_DC;
        if (c + 1 < nchunk) gload(d0 + VB_DC);
        f32x16 accX;
#pragma unroll
        for (int r = 0; r < 16; ++r) accX[r] = 0.f;
#pragma unroll
        for (int i = 0; i < KTW; ++i) {
            const int tile = wave + 4 * i;
            if (tile < KT) {
                // dA[t, tile] += x[t, dchunk] . dU[dchunk, tile]
                const float* xa = xs + l31 * 33 + half;
                const float* ub = dUs + half * KS + tile * 32 + l31;
#pragma unroll
                for (int dd = 0; dd < VB_DC; dd += 2) accA[i] = mfma32(xa[dd], ub[dd * KS], accA[i]);
                // dx[t, dchunk] += A[t, tile] . dU[dchunk, tile]^T
                const float* aa = As + l31 * KS + tile * 32 + half;
                const float* ud = dUs + l31 * KS + tile * 32 + half;
#pragma unroll
                for (int kk = 0; kk < 32; kk += 2) accX = mfma32(aa[kk], ud[kk], accX);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 32 * 33 + mfma32_row(r, lane) * 33 + l31] = accX[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int tt = e >> 5, dd = e & 31;
            const float s = red[tt * 33 + dd] + red[32 * 33 + tt * 33 + dd] + red[2 * 32 * 33 + tt * 33 + dd] +
                            red[3 * 32 * 33 + tt * 33 + dd];
            if (t0 + tt < T) {
                float* p = dx + ((int64_t)b * T + t0 + tt) * lddx + d0 + dd;
                *p = accumulate ? (*p + s) : s;
            }
        }
        if (c + 1 < nchunk) sstore();
        __syncthreads();
    }

    // ---- epilogue: dA - ctil -> LDS, softmax backward row-wise ---------------------------------
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
        const int tile = wave + 4 * i;
        if (tile < KT) {
            const int k = tile * 32 + l31;
            if (k < K) {
                const float ct = cc[k];
#pragma unroll
                for (int r = 0; r < 16; ++r) dUs[mfma32_row(r, lane) * KS + k] = accA[i][r] - ct;
            }
        }
    }
    __syncthreads();
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, t = t0 + row;
        if (t >= T) continue;   // wave-uniform
        float* out = dassign + ((int64_t)b * T + t) * K;
        if (SOFTMAX) {
            float a[KPL], g[KPL], dot = 0.f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int c = lane + 64 * j;
                a[j] = (c < K) ? As[row * KS + c] : 0.f;
                g[j] = (c < K) ? dUs[row * KS + c] : 0.f;
                dot = fmaf(a[j], g[j], dot);
            }
            dot = wave_sum(dot);
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int c = lane + 64 * j;
                if (c < K) out[c] = a[j] * (g[j] - dot);
            }
        } else {
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int c = lane + 64 * j;
                if (c < K) out[c] = dUs[row * KS + c];
            }
        }
    }
}

// dW2[d,k] = - sum_b s[b,k] * (u[b,k] dO[b,d,k] - v[b,k] N[b,d,k]).  blockIdx.y splits the clips (a thread per float4 of
// [D, K] alone is 256 workgroups at cfg-2's video shape and 8 at the audio shape: too few loads in flight); the partial sums
// are added in split order by vlad_bwd_dcentres_reduce_kernel, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void vlad_bwd_dcentres_kernel(const float* __restrict__ dO,
                                                                const float* __restrict__ N,
                                                                const float* __restrict__ asum,
                                                                const float* __restrict__ u,
                                                                const float* __restrict__ v, int B, int D, int K,
                                                                int bper, float* __restrict__ dW2,
                                                                const float* __restrict__ colsq_raw, int n_bf16, int64_t dob) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index into [D,K]
    const int64_t n4 = (int64_t)D * K / 4;
    if (i >= n4) return;
    const int k = (int)((i * 4) % K);
    const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
        const float4 a = reinterpret_cast<const float4*>(dO + (int64_t)b * dob)[i];
        float4 n = vb_ld4(N, (int64_t)b * n4 + i, n_bf16);
        if (colsq_raw) {                   // N = U * rsqrt(max(colsq, eps)): see vlad_bwd_coldots_kernel
            const float4 c = *reinterpret_cast<const float4*>(colsq_raw + (int64_t)b * K + k);
            n.x *= rsqrtf(fmaxf(c.x, kL2Eps)); n.y *= rsqrtf(fmaxf(c.y, kL2Eps));
            n.z *= rsqrtf(fmaxf(c.z, kL2Eps)); n.w *= rsqrtf(fmaxf(c.w, kL2Eps));
        }
        const float4 s = *reinterpret_cast<const float4*>(asum + (int64_t)b * K + k);
        const float4 uu = *reinterpret_cast<const float4*>(u + (int64_t)b * K + k);
        const float4 vv = *reinterpret_cast<const float4*>(v + (int64_t)b * K + k);
        acc.x -= s.x * (uu.x * a.x - vv.x * n.x);
        acc.y -= s.y * (uu.y * a.y - vv.y * n.y);
        acc.z -= s.z * (uu.z * a.z - vv.z * n.z);
        acc.w -= s.w * (uu.w * a.w - vv.w * n.w);
    }
    reinterpret_cast<float4*>(dW2)[(int64_t)blockIdx.y * n4 + i] = acc;
}

__global__ __launch_bounds__(256) void vlad_bwd_dcentres_reduce_kernel(const float4* __restrict__ part, int Z, int64_t n4,
                                                                       float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = part[i];
    for (int z = 1; z < Z; ++z) {
        const float4 p = part[(int64_t)z * n4 + i];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    out[i] = s;
}

constexpr int VB_DC_SPLITS = 16;          // most clip splits of the centres' gradient (its partial sums live in the workspace)
static int dcentres_splits(int B, int D, int K) {
    const int64_t wgx = ((int64_t)D * K / 4 + 255) / 256;
    int z = (int)((1024 + wgx - 1) / wgx);
    if (z > VB_DC_SPLITS) z = VB_DC_SPLITS;
    if (z > B) z = B;
    return z < 1 ? 1 : z;
}
// part: room for dcentres_splits(B, D, K) x [D, K] floats, or null (then one pass straight into dW2)
// In-place second half of dA -> dlogit~ when the GEMM ran with the plain store epilogue (K > 256): row (b, t) of `dl` holds dA; one
// wave per row:  g = dA - ctil[b];  softmax: dl = a (g - sum_j a_j g_j) with a = softmax(logits * scale + shift) recomputed;
// otherwise dl = g.  The arithmetic of tile_gemm's fused epilogue, KPL = K / 64 values per lane.
template <int KPL>
__global__ __launch_bounds__(256) void vlad_softmax_bwd_rows_kernel(float* __restrict__ dl, const float* __restrict__ logits, int logits_bf16,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ ctil, int T, int64_t rows, int softmax) {
    constexpr int K = 64 * KPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / T);
    float* out = dl + row * K;
    const float* ct = ctil + (int64_t)b * K;
    float a[KPL], gg[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) gg[j] = out[lane + 64 * j] - ct[lane + 64 * j];
    if (!softmax) {
#pragma unroll
        for (int j = 0; j < KPL; ++j) out[lane + 64 * j] = gg[j];
        return;
    }
    const float* lr = logits + row * K;
    const unsigned short* lrb = reinterpret_cast<const unsigned short*>(logits) + row * K;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int c = lane + 64 * j;
        const float lv = logits_bf16 ? __uint_as_float((unsigned)lrb[c] << 16) : lr[c];
        a[j] = fmaf(lv, scale ? scale[c] : 1.f, shift ? shift[c] : 0.f);
        mx = fmaxf(mx, a[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        a[j] = __expf(a[j] - mx);
        sum += a[j];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        a[j] *= inv;
        dot = fmaf(a[j], gg[j], dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < KPL; ++j) out[lane + 64 * j] = a[j] * (gg[j] - dot);
}

static void launch_dcentres(const float* dO, const float* N, const float* asum, const float* u, const float* v, int B, int D, int K,
                            float* part, float* dW2, hipStream_t s, const float* colsq_raw = nullptr, int n_bf16 = 0, int64_t dob = 0) {
    if (dob == 0) dob = (int64_t)D * K;
    const int64_t n4 = (int64_t)D * K / 4;
    const unsigned wgx = (unsigned)((n4 + 255) / 256);
    const int Z = part ? dcentres_splits(B, D, K) : 1;
    if (Z == 1) {
        hipLaunchKernelGGL(vlad_bwd_dcentres_kernel, dim3(wgx), dim3(256), 0, s, dO, N, asum, u, v, B, D, K, B, dW2, colsq_raw, n_bf16, dob);
        return;
    }
    const int bper = (B + Z - 1) / Z, Zeff = (B + bper - 1) / bper;
    hipLaunchKernelGGL(vlad_bwd_dcentres_kernel, dim3(wgx, (unsigned)Zeff), dim3(256), 0, s, dO, N, asum, u, v, B, D, K, bper, part,
                       colsq_raw, n_bf16, dob);
    hipLaunchKernelGGL(vlad_bwd_dcentres_reduce_kernel, dim3(wgx), dim3(256), 0, s, (const float4*)part, Zeff, n4, (float4*)dW2);
}

// ---- split-bf16 tile form of the main step (VLAD_PRECISION bf16x3) --------------------------------------------------
// dU = u dO - v N is formed ONCE per clip and written as B-operand fragment tiles in both orientations; the two GEMMs
// of the backward then run on the bf16 pipe through the tile GEMM (tile_gemm.h):
//   dA[t,k] = sum_d x[t,d] dU[d,k]      A = row tiles of x,            B = ub1[b][d-step][k-tile]   + softmax backward epilogue
//   dx[t,d] = sum_k a[t,k] dU[d,k]      A = row tiles of the assignment, B = ub2[b][k-step][d-tile]  (+ dl . W^T as a second
//                                       reduction segment when the caller chains the assignment GEMM's backward)
// grid (D/32, B): one workgroup stages a [32 d][K] chunk of dU in LDS and emits both tile forms from it.
// 512 threads: the [32][K + 1] staging tiles (69 KB with the g0 products at K = 256) allow two workgroups per CU whatever
// their size, and a streaming kernel wants more than two waves per SIMD
constexpr int VB_DU_NT = 512;
__global__ __launch_bounds__(VB_DU_NT) void vlad_bwd_du_tiles_kernel(const float* __restrict__ dO, const float* __restrict__ N,
                                                                const float* __restrict__ ug, const float* __restrict__ vg,
                                                                int D, int K, uint4* __restrict__ ub1, uint4* __restrict__ ub2,
                                                                const float* __restrict__ colsq, float* __restrict__ g0, int raw, int planes,
                                                                int rw, int64_t dob) {
    // raw (LPM_VLAD_NRM_RAW): N holds the un-normalised sums U; N = U * rsqrt(max(colsq, eps)), and g0's U is read as it is
    // rw: rows of D per workgroup -- 32, or 16 when only the d-reduction tiles are wanted (ub2 == NULL): half the LDS, so that two
    // workgroups share a CU at K = 512 as well (cfg-5: 137 KB -> 69 KB with the g0 products)
    extern __shared__ float dus[];           // [rw][K+1], then u[K], v[K], rn[K] (, then prod [rw][K+1] when g0)
    const int KS = K + 1;
    float* cu = dus + rw * KS;
    float* cv = cu + K;
    float* rn = cv + K;                      // g0 && !raw: norm of the un-normalised column, U = N * rn;  raw: its inverse
    float* prod = rn + K;                    // g0 only: dU * U
    const int tid = threadIdx.x, b = blockIdx.y, d0 = blockIdx.x * rw;
    for (int k = tid; k < K; k += VB_DU_NT) {
        cu[k] = ug[(int64_t)b * K + k];
        cv[k] = vg[(int64_t)b * K + k];
        if (raw) rn[k] = rsqrtf(fmaxf(colsq[(int64_t)b * K + k], kL2Eps));
        else if (g0) rn[k] = sqrtf(fmaxf(colsq[(int64_t)b * K + k], kL2Eps));
    }
    __syncthreads();
    const int K4 = K / 4;
    const float* ob = dO + (int64_t)b * dob + (int64_t)d0 * K;
    const int64_t nb4 = ((int64_t)b * D + d0) * K4;
    const int n_bf16 = planes == 1;          // bf16 storage: N holds the sums as bf16
    for (int i = tid; i < rw * K4; i += VB_DU_NT) {
        const int r = i / K4, k = (i % K4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(ob + (int64_t)r * K + k);
        float4 n = vb_ld4(N, nb4 + (int64_t)r * K4 + k / 4, n_bf16);
        const float4 nu = n;               // raw: the un-normalised value
        if (raw) {
            n.x *= rn[k + 0]; n.y *= rn[k + 1]; n.z *= rn[k + 2]; n.w *= rn[k + 3];
        }
        float* dst = dus + r * KS + k;
        dst[0] = cu[k + 0] * a.x - cv[k + 0] * n.x;
        dst[1] = cu[k + 1] * a.y - cv[k + 1] * n.y;
        dst[2] = cu[k + 2] * a.z - cv[k + 2] * n.z;
        dst[3] = cu[k + 3] * a.w - cv[k + 3] * n.w;
        if (g0) {
            float* pd = prod + r * KS + k;
            pd[0] = raw ? dst[0] * nu.x : dst[0] * n.x * rn[k + 0];
            pd[1] = raw ? dst[1] * nu.y : dst[1] * n.y * rn[k + 1];
            pd[2] = raw ? dst[2] * nu.z : dst[2] * n.z * rn[k + 2];
            pd[3] = raw ? dst[3] * nu.w : dst[3] * n.w * rn[k + 3];
        }
    }
    __syncthreads();
    if (g0) {
        // g0[b][d] = sum_k dU[d,k] U[d,k]: the only thing the input batch norm's gamma gradient needs from this clip when the
        // frames themselves need no gradient (see ops._NetVLAD.backward); fixed summation order (lane-strided, then the wave tree)
        const int lane = tid & 63, wave = tid >> 6;
        const int RPW = rw / (VB_DU_NT / 64);           // rows per wave
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            float acc = 0.f;
            for (int k = lane; k < K; k += 64) acc += prod[r * KS + k];
            acc = wave_sum(acc);
            if (lane == 0) g0[(int64_t)b * D + d0 + r] = acc;
        }
    }
    const int KT = K / 32, DS = D / 16, KS16 = K / 16, DT = D / 32;
    for (int it = tid; it < (rw / 16) * KT * 64; it += VB_DU_NT) {    // reduction over d, columns k
        const int lane = it & 63, kt = (it >> 6) % KT, dsl = (it >> 6) / KT;
        const int k = kt * 32 + (lane & 31), dl = dsl * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dus[(dl + e) * KS + k];
        uint4 hi, lo;
        tg_split8(v, hi, lo);
        if (planes == 1) {                 // bf16 storage: plain bf16 tiles
            ub1[(((int64_t)b * DS + d0 / 16 + dsl) * KT + kt) * 64 + lane] = hi;
            continue;
        }
        const int64_t base = ((((int64_t)b * DS + d0 / 16 + dsl) * KT + kt) * 2) * 64 + lane;
        ub1[base] = hi;
        ub1[base + 64] = lo;
    }
    if (ub2 == nullptr) return;                                  // no input gradient wanted: the dx GEMM's operand is not needed
    // (rw == 32 here: a tile of this form is 32 rows of D)
    for (int it = tid; it < KS16 * 64; it += VB_DU_NT) {              // reduction over k, columns d
        const int lane = it & 63, ks = it >> 6;
        const int dl = lane & 31, k = ks * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dus[dl * KS + k + e];
        uint4 hi, lo;
        tg_split8(v, hi, lo);
        const int64_t base = ((((int64_t)b * KS16 + ks) * DT + d0 / 32) * 2) * 64 + lane;
        ub2[base] = hi;
        ub2[base + 64] = lo;
    }
}

// assignment (softmax recomputed from the logits, or the similarities themselves) -> row tiles [b][mt][ks][plane][lane].
// grid (MT, B), one workgroup per 32-frame row tile; K <= 512.
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void vlad_bwd_assign_rows_kernel(const float* __restrict__ assign, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, int T, int K, int MT,
                                                                   uint4* __restrict__ ar) {
    extern __shared__ float as[];            // [32][K+1]
    const int KS = K + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = blockIdx.x, b = blockIdx.y;
    for (int i = tid; i < 32 * KS; i += 256) as[i] = 0.f;
    __syncthreads();
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, t = mt * 32 + row;
        if (t >= T) continue;                // wave-uniform
        const float* arow = assign + ((int64_t)b * T + t) * K;
        if (SOFTMAX) {
            float v[8];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = lane + 64 * j;
                v[j] = (c < K) ? fmaf(arow[c], scale ? scale[c] : 1.f, shift ? shift[c] : 0.f) : -INFINITY;
                m = fmaxf(m, v[j]);
            }
            m = wave_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = __expf(v[j] - m);
                sum += v[j];
            }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = lane + 64 * j;
                if (c < K) as[row * KS + c] = v[j] * inv;
            }
        } else {
            for (int c = lane; c < K; c += 64) as[row * KS + c] = arow[c];
        }
    }
    __syncthreads();
    const int KS16 = K / 16;
    for (int it = tid; it < KS16 * 64; it += 256) {
        const int ln = it & 63, ks = it >> 6;
        const int row = ln & 31, k = ks * 16 + 8 * (ln >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = as[row * KS + k + e];
        uint4 hi, lo;
        tg_split8(v, hi, lo);
        const int64_t base = ((((int64_t)b * MT + mt) * KS16 + ks) * 2) * 64 + ln;
        ar[base] = hi;
        ar[base + 64] = lo;
    }
}

// ---- k-major forms (LPM_VLAD_RAW_KMAJOR): dO and the un-normalised sums U are [B, K, D], as the NetVladV1 cluster encoder hands the
// gradient back and as lpm_vlad_aggregate_raw_kmajor_fwd stored the sums -- no transpose of dO, no d-major copy of U -----------------------
// (Round 3: with packed fp32 instructions allowed, the compiler paired dw and nw into v_pk_fma_f32 and the nw half came out wrong by a
// few per cent in ~3 % of training steps with a second process on the GPU -- tools/determinism_check.py; the library is built without
// packed fp32 operations since, _build.py.)
// one wave per (clip, cluster) row: dots[b][0][0..2][k] = <dO_k, N_k>, <dO_k, W2_k>, <N_k, W2_k>,  N_k = U_k rsqrt(max(colsq, eps))
__global__ __launch_bounds__(256) void vlad_bwd_coldots_k_kernel(const float* __restrict__ dO, const float* __restrict__ U,
                                                                 const float* __restrict__ W2T, const float* __restrict__ colsq, int D, int K,
                                                                 int64_t rows, float* __restrict__ dots) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);           // b * K + k
    if (row >= rows) return;                                                     // wave-uniform
    const int k = (int)(row % K);
    const int64_t b = row / K;
    const float4* pd = reinterpret_cast<const float4*>(dO + row * D);
    const float4* pu = reinterpret_cast<const float4*>(U + row * D);
    const float4* pw = W2T ? reinterpret_cast<const float4*>(W2T + (int64_t)k * D) : nullptr;
    float p = 0.f, dw = 0.f, nw = 0.f;
    for (int i = lane; i < D / 4; i += 64) {
        const float4 a = pd[i], n = pu[i];
        p = fmaf(a.x, n.x, p); p = fmaf(a.y, n.y, p); p = fmaf(a.z, n.z, p); p = fmaf(a.w, n.w, p);
        if (pw) {
            const float4 w = pw[i];
            dw = fmaf(a.x, w.x, dw); dw = fmaf(a.y, w.y, dw); dw = fmaf(a.z, w.z, dw); dw = fmaf(a.w, w.w, dw);
            nw = fmaf(n.x, w.x, nw); nw = fmaf(n.y, w.y, nw); nw = fmaf(n.z, w.z, nw); nw = fmaf(n.w, w.w, nw);
        }
    }
    p = wave_sum(p); dw = wave_sum(dw); nw = wave_sum(nw);
    if (lane == 0) {
        const float iv = rsqrtf(fmaxf(colsq[row], kL2Eps));
        float* out = dots + b * 3 * K;
        out[k] = p * iv;
        out[K + k] = dw;
        out[2 * K + k] = nw * iv;
    }
}

// dU = u dO - v N as B-operand tiles ub1[b][d-step][k-tile] straight from the k-major tensors: a fragment lane (cluster, d half) wants 8
// consecutive d of its cluster row -- two float4 of dO and of U, no LDS staging.  grid (D/32, B), 512 threads; g0[b][d] = sum_k dU U (when
// wanted) is reduced through LDS in a fixed order.
__global__ __launch_bounds__(512) void vlad_bwd_du_tiles_k_kernel(const float* __restrict__ dO, const float* __restrict__ U,
                                                                  const float* __restrict__ ug, const float* __restrict__ vg,
                                                                  const float* __restrict__ colsq, int D, int K, uint4* __restrict__ ub1,
                                                                  float* __restrict__ g0) {
    extern __shared__ float pl[];            // g0 only: [32 d][KT * 32 + 1] products summed over nothing yet
    const int tid = threadIdx.x, b = blockIdx.y, d0 = blockIdx.x * 32;
    const int KT = K / 32, DS = D / 16;
    const int KS = K + 1;
    for (int it = tid; it < 2 * KT * 64; it += 512) {
        const int lane = it & 63, kt = (it >> 6) % KT, dsl = (it >> 6) / KT;
        const int k = kt * 32 + (lane & 31), dl = dsl * 16 + 8 * (lane >> 5);
        const int64_t row = (int64_t)b * K + k;
        const float uu = ug[row], vv = vg[row], rn = rsqrtf(fmaxf(colsq[row], kL2Eps));
        const float* po = dO + row * D + d0 + dl;
        const float* pu = U + row * D + d0 + dl;
        const float4 a0 = *reinterpret_cast<const float4*>(po), a1 = *reinterpret_cast<const float4*>(po + 4);
        const float4 n0 = *reinterpret_cast<const float4*>(pu), n1 = *reinterpret_cast<const float4*>(pu + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float nv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = uu * av[e] - vv * (nv[e] * rn);
            if (g0) pl[(dl + e) * KS + k] = v[e] * nv[e];
        }
        uint4 hi, lo;
        tg_split8(v, hi, lo);
        const int64_t base = ((((int64_t)b * DS + d0 / 16 + dsl) * KT + kt) * 2) * 64 + lane;
        ub1[base] = hi;
        ub1[base + 64] = lo;
    }
    if (g0) {
        __syncthreads();
        const int lane = tid & 63, wave = tid >> 6;
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float acc = 0.f;
            for (int k = lane; k < K; k += 64) acc += pl[r * KS + k];
            acc = wave_sum(acc);
            if (lane == 0) g0[(int64_t)b * D + d0 + r] = acc;
        }
    }
}

// dW2^T[k, d] = - sum_b s[b,k] (u[b,k] dO[b,k,d] - v[b,k] N[b,k,d]) over a range of clips per blockIdx.y (partial sums added in split
// order by vlad_bwd_dcentres_reduce_kernel, then transposed to [D, K])
__global__ __launch_bounds__(256) void vlad_bwd_dcentres_k_kernel(const float* __restrict__ dO, const float* __restrict__ U,
                                                                  const float* __restrict__ asum, const float* __restrict__ u,
                                                                  const float* __restrict__ v, const float* __restrict__ colsq, int B, int D,
                                                                  int K, int bper, float* __restrict__ part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index into [K, D]
    const int64_t n4 = (int64_t)K * D / 4;
    if (i >= n4) return;
    const int k = (int)((i * 4) / D);
    const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = b0; b < b1; ++b) {
        const int64_t row = (int64_t)b * K + k;
        const float4 a = reinterpret_cast<const float4*>(dO + (int64_t)b * K * D)[i];
        const float4 n = reinterpret_cast<const float4*>(U + (int64_t)b * K * D)[i];
        const float s = asum[row], uu = u[row], vr = v[row] * rsqrtf(fmaxf(colsq[row], kL2Eps));
        acc.x -= s * (uu * a.x - vr * n.x);
        acc.y -= s * (uu * a.y - vr * n.y);
        acc.z -= s * (uu * a.z - vr * n.z);
        acc.w -= s * (uu * a.w - vr * n.w);
    }
    reinterpret_cast<float4*>(part)[(int64_t)blockIdx.y * n4 + i] = acc;
}

static size_t bwd_main_lds_bytes(int K) {
    const int KS = (K + 31) / 32 * 32 + 1;
    return (size_t)(2 * 32 * KS + 32 * 33 + 4 * 32 * 33 + 3 * K) * sizeof(float);
}

}  // namespace lpm

extern "C" size_t lpm_vlad_bwd_workspace_bytes(int B, int D, int K) {
    return ((size_t)B * lpm::VB_DSPLIT * 3 * K + 3 * (size_t)B * K + (size_t)B * D * K) * sizeof(float);
}

extern "C" int lpm_vlad_aggregate_bwd(const float* dout, const float* nrm, const float* asum, const float* colsq,
                                      const float* csq, const float* gsq, const float* assign, const float* scale,
                                      const float* shift, const float* x, int64_t ldx, const float* centres, int B,
                                      int T, int D, int K, int flags, float* dassign, float* dx, int64_t lddx,
                                      int accumulate_dx, float* dcentres, void* workspace, size_t workspace_bytes,
                                      lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dout && nrm && asum && colsq && csq && gsq && assign && x && dassign && dx && workspace,
                LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd: null pointer");
    const bool residual = (flags & LPM_VLAD_RESIDUAL) != 0;
    const bool sm = (flags & LPM_VLAD_SOFTMAX) != 0;
    LPM_REQUIRE(!residual || (centres && dcentres), LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd: RESIDUAL needs centres/dcentres");
    LPM_REQUIRE(!(flags & LPM_VLAD_NRM_RAW), LPM_ERR_BADARG,
                "lpm_vlad_aggregate_bwd: LPM_VLAD_NRM_RAW (un-normalised nrm) is understood by lpm_vlad_aggregate_bwd_tiles only");
    LPM_REQUIRE(B > 0 && T > 0 && ldx >= D && lddx >= D, LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd: bad sizes");
    LPM_REQUIRE(D % 128 == 0 && K % 4 == 0 && K <= 512 && ldx % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_bwd: need D %% 128 == 0, K %% 4 == 0, K <= 512, ldx %% 4 == 0 (D=%d K=%d)", D, K);
    LPM_REQUIRE(workspace_bytes >= lpm_vlad_bwd_workspace_bytes(B, D, K), LPM_ERR_WORKSPACE,
                "lpm_vlad_aggregate_bwd: workspace too small");
    LPM_REQUIRE((((uintptr_t)x | (uintptr_t)dout | (uintptr_t)nrm | (uintptr_t)workspace) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_bwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    float* dots = ws;
    float* u = dots + (size_t)B * VB_DSPLIT * 3 * K;
    float* v = u + (size_t)B * K;
    float* ctil = v + (size_t)B * K;
    float* dod = ctil + (size_t)B * K;   // d-major copy of dout when it arrives k-major
    const float* dO = dout;
    if (flags & LPM_VLAD_OUT_KMAJOR) {
        launch_kmajor_to_dmajor(dout, B, D, K, dod, s);
        dO = dod;
    }
    hipLaunchKernelGGL(vlad_bwd_coldots_kernel, dim3(B, VB_DSPLIT), dim3(256), 0, s, dO, nrm, residual ? centres : nullptr,
                       D, K, dots, (const float*)nullptr, 0, (int64_t)D * K);
    hipLaunchKernelGGL(vlad_bwd_coeff_kernel, dim3(B), dim3(256), 0, s, dots, colsq, csq, gsq, K, u, v, ctil, VB_DSPLIT);
    const int nts = (T + VB_TS - 1) / VB_TS;
    const size_t lds = bwd_main_lds_bytes(K);
    dim3 grid(B * nts);
#define LPM_VB_LAUNCH(KTW, SM)                                                                                        \
    do {                                                                                                              \
        auto kern = vlad_bwd_main_kernel<KTW, SM>;                                                                    \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            (void)hipGetLastError();                                                                                  \
            set_error("lpm_vlad_aggregate_bwd: cannot reserve %zu bytes of LDS", lds);                                \
            return LPM_ERR_LAUNCH;                                                                                    \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, dO, nrm, u, v, ctil, assign, scale, shift, x, ldx, T, D, K, \
                           nts, dassign, dx, lddx, accumulate_dx);                                                    \
    } while (0)
    if (K <= 128) { if (sm) LPM_VB_LAUNCH(1, true); else LPM_VB_LAUNCH(1, false); }
    else if (K <= 256) { if (sm) LPM_VB_LAUNCH(2, true); else LPM_VB_LAUNCH(2, false); }
    else { if (sm) LPM_VB_LAUNCH(4, true); else LPM_VB_LAUNCH(4, false); }
#undef LPM_VB_LAUNCH
    if (residual) {
        launch_dcentres(dO, nrm, asum, u, v, B, D, K, nullptr, dcentres, s);
    }
    return check_launch("lpm_vlad_aggregate_bwd");
}

// ---- tile form: workspace = [dots | u | v | ctil | d-major dout copy | ub1 | ub2 | ar] ------------------------------
namespace lpm {
struct BwdTilesLayout {
    size_t dots, u, v, ctil, dod, ub1, ub2, ar, dcp, total;     // byte offsets
    int MT;
};
static BwdTilesLayout bwd_tiles_layout(int B, int T, int D, int K) {
    BwdTilesLayout L;
    size_t o = 0;
    L.dots = o; o += (size_t)B * VB_DSPLIT * 3 * K * 4;
    L.u = o; o += (size_t)B * K * 4;
    L.v = o; o += (size_t)B * K * 4;
    L.ctil = o; o += (size_t)B * K * 4;
    o = (o + 255) / 256 * 256;
    L.dod = o; o += (size_t)B * D * K * 4;
    L.ub1 = o; o += (size_t)B * D * K * 4;             // D/16 steps x K/32 tiles x 2 KB = 4 bytes per element
    L.ub2 = o; o += (size_t)B * D * K * 4;
    L.MT = 2 * ((T + 63) / 64);
    L.ar = o; o += (size_t)B * L.MT * (K / 16) * 2048;
    L.dcp = o; o += (size_t)dcentres_splits(B, D, K) * D * K * 4;       // partial sums of the centres' gradient
    L.total = o;
    return L;
}
}  // namespace lpm

extern "C" size_t lpm_vlad_bwd_tiles_workspace_bytes(int B, int T, int D, int K) { return lpm::bwd_tiles_layout(B, T, D, K).total; }

static int vlad_aggregate_bwd_tiles_impl(const float* dout, int64_t dob, const float* nrm, const float* asum, const float* colsq,
                                         const float* csq, const float* gsq, const float* assign, const float* scale,
                                         const float* shift, const void* xr, const float* centres, int B, int T, int D,
                                         int K, int flags, float* dassign, float* dcentres, float* g0, void* workspace,
                                         size_t workspace_bytes, lpm_stream_t stream);
extern "C" int lpm_vlad_aggregate_bwd_tiles(const float* dout, const float* nrm, const float* asum, const float* colsq,
                                            const float* csq, const float* gsq, const float* assign, const float* scale,
                                            const float* shift, const void* xr, const float* centres, int B, int T, int D,
                                            int K, int flags, float* dassign, float* dcentres, float* g0, void* workspace,
                                            size_t workspace_bytes, lpm_stream_t stream) {
    return vlad_aggregate_bwd_tiles_impl(dout, (int64_t)D * K, nrm, asum, colsq, csq, gsq, assign, scale, shift, xr, centres, B, T, D, K, flags,
                                         dassign, dcentres, g0, workspace, workspace_bytes, stream);
}
// ... with the clips' gradients dout_batch_stride elements apart (>= D * K, a multiple of 4): dout is a column slice of the gradient
// of the concatenated descriptors (ops.DescriptorSlots), read in place.  d-major gradients only (no LPM_VLAD_OUT_KMAJOR / RAW_KMAJOR).
extern "C" int lpm_vlad_aggregate_bwd_tiles_ld(const float* dout, int64_t dout_batch_stride, const float* nrm, const float* asum,
                                               const float* colsq, const float* csq, const float* gsq, const float* assign,
                                               const float* scale, const float* shift, const void* xr, const float* centres, int B, int T,
                                               int D, int K, int flags, float* dassign, float* dcentres, float* g0, void* workspace,
                                               size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dout_batch_stride >= (int64_t)D * K && dout_batch_stride % 4 == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_bwd_tiles_ld: the batch stride must be >= D * K and a multiple of 4");
    LPM_REQUIRE(dout_batch_stride == (int64_t)D * K || !(flags & (LPM_VLAD_OUT_KMAJOR | LPM_VLAD_RAW_KMAJOR)), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_bwd_tiles_ld: a strided gradient is read in the d-major layout only");
    return vlad_aggregate_bwd_tiles_impl(dout, dout_batch_stride, nrm, asum, colsq, csq, gsq, assign, scale, shift, xr, centres, B, T, D, K,
                                         flags, dassign, dcentres, g0, workspace, workspace_bytes, stream);
}
static int vlad_aggregate_bwd_tiles_impl(const float* dout, int64_t dob, const float* nrm, const float* asum, const float* colsq,
                                         const float* csq, const float* gsq, const float* assign, const float* scale,
                                         const float* shift, const void* xr, const float* centres, int B, int T, int D,
                                         int K, int flags, float* dassign, float* dcentres, float* g0, void* workspace,
                                         size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dout && nrm && asum && colsq && csq && gsq && assign && xr && dassign && workspace, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_bwd_tiles: null pointer");
    LPM_REQUIRE(!g0 || dcentres, LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd_tiles: g0 needs a dcentres buffer (with or without RESIDUAL)");
    const bool residual = (flags & LPM_VLAD_RESIDUAL) != 0;
    const bool sm = (flags & LPM_VLAD_SOFTMAX) != 0;
    LPM_REQUIRE(!residual || (centres && dcentres), LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd_tiles: RESIDUAL needs centres/dcentres");
    LPM_REQUIRE(B > 0 && T > 0 && D > 0 && D % 32 == 0 && K > 0 && K % 32 == 0 && K <= 512, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_bwd_tiles: need D %% 32 == 0, K %% 32 == 0, K <= 512 (D=%d K=%d)", D, K);
    const BwdTilesLayout L = bwd_tiles_layout(B, T, D, K);
    LPM_REQUIRE(workspace_bytes >= L.total, LPM_ERR_WORKSPACE, "lpm_vlad_aggregate_bwd_tiles: workspace too small");
    LPM_REQUIRE((((uintptr_t)xr | (uintptr_t)dout | (uintptr_t)nrm | (uintptr_t)workspace) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_bwd_tiles: pointers must be 16-byte aligned");
    const int planes = (flags & LPM_VLAD_TILES_BF16) ? 1 : 2;
    LPM_REQUIRE(planes == 2 || g0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_bwd_tiles: the bf16-storage form has no input-gradient path (it needs g0: frames without a gradient)");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* dots = (float*)(ws + L.dots);
    float* u = (float*)(ws + L.u);
    float* v = (float*)(ws + L.v);
    float* ctil = (float*)(ws + L.ctil);
    float* dod = (float*)(ws + L.dod);
    uint4* ub1 = (uint4*)(ws + L.ub1);
    uint4* ub2 = (uint4*)(ws + L.ub2);
    uint4* ar = (uint4*)(ws + L.ar);
    const float* dO = dout;
    const bool rk = (flags & LPM_VLAD_RAW_KMAJOR) != 0;      // dout AND nrm (the un-normalised sums) are k-major [B, K, D]
    LPM_REQUIRE(!rk || (g0 && planes == 2 && D % 32 == 0), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_bwd_tiles: the k-major form is the split-bf16, no-input-gradient form (needs g0)");
    if ((flags & LPM_VLAD_OUT_KMAJOR) && !rk) {
        launch_kmajor_to_dmajor(dout, B, D, K, dod, s);
        dO = dod;
    }
    const bool raw = (flags & LPM_VLAD_NRM_RAW) != 0 || rk;
    const float* colsq_raw = raw ? colsq : nullptr;
    float* w2t = dod;                                        // k-major form: centres^T [K, D] (the d-major copy of dout is not needed there)
    if (rk) {
        if (residual) launch_kmajor_to_dmajor(centres, 1, /*D=*/K, /*K=*/D, w2t, s);        // [D, K] -> [K, D]
        const int64_t rows = (int64_t)B * K;
        hipLaunchKernelGGL(vlad_bwd_coldots_k_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dO, nrm, residual ? w2t : nullptr,
                           colsq, D, K, rows, dots);
        hipLaunchKernelGGL(vlad_bwd_coeff_kernel, dim3(B), dim3(256), 0, s, dots, colsq, csq, gsq, K, u, v, ctil, 1);
        const size_t lds = (size_t)32 * (K + 1) * sizeof(float);
        auto kern = vlad_bwd_du_tiles_k_kernel;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("lpm_vlad_aggregate_bwd_tiles: cannot reserve %zu bytes of LDS", lds);
            return LPM_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(D / 32, B), dim3(512), lds, s, dO, nrm, u, v, colsq, D, K, ub1, g0);
    } else {
    hipLaunchKernelGGL(vlad_bwd_coldots_kernel, dim3(B, VB_DSPLIT), dim3(256), 0, s, dO, nrm, residual ? centres : nullptr, D, K, dots,
                       colsq_raw, planes == 1 ? 1 : 0, dob);
    hipLaunchKernelGGL(vlad_bwd_coeff_kernel, dim3(B), dim3(256), 0, s, dots, colsq, csq, gsq, K, u, v, ctil, VB_DSPLIT);
    {
        const int rw = (g0 && K >= 512) ? 16 : 32;          // only the d-reduction tiles: 16-row workgroups keep two per CU at K = 512
        const size_t lds = (size_t)(rw * (K + 1) + 3 * K + (g0 ? rw * (K + 1) : 0)) * sizeof(float);
        auto kern = vlad_bwd_du_tiles_kernel;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("lpm_vlad_aggregate_bwd_tiles: cannot reserve %zu bytes of LDS", lds);
            return LPM_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(D / rw, B), dim3(VB_DU_NT), lds, s, dO, nrm, u, v, D, K, ub1, g0 ? (uint4*)nullptr : ub2, colsq, g0,
                           raw ? 1 : 0, planes, rw, dob);
    }
    }
    if (!g0) {       // the assignment's row tiles are the A operand of the dx GEMM only
        const size_t lds = (size_t)32 * (K + 1) * sizeof(float);
        if (sm) {
            auto kern = vlad_bwd_assign_rows_kernel<true>;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(L.MT, B), dim3(256), lds, s, assign, scale, shift, T, K, L.MT, ar);
        } else {
            auto kern = vlad_bwd_assign_rows_kernel<false>;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(L.MT, B), dim3(256), lds, s, assign, scale, shift, T, K, L.MT, ar);
        }
    }
    const int DS = D / 16, KT = K / 32;
    const int64_t U = 64 * planes;
    TileGemmArgs g{};
    g.a = (const uint4*)xr; g.a_tile = (int64_t)DS * U; g.a_step = U; g.a_batch = (int64_t)L.MT * DS * U; g.a_tiles = L.MT;
    g.b = ub1; g.b_tile = U; g.b_step = (int64_t)KT * U; g.b_batch = (int64_t)DS * KT * U; g.b_tiles = KT;
    g.rb_per_batch = L.MT / 2; g.steps_per_split = DS; g.total_steps = DS;
    g.out = dassign; g.rows_valid = T; g.cols_valid = K;
    g.logits = assign; g.logits_bf16 = planes == 1 ? 1 : 0; g.scale = scale; g.shift = shift; g.ctil = ctil; g.softmax = sm ? 1 : 0;
    // K = 512 (cfg-5's video stream): the fused epilogue needs all K columns of a row in one workgroup -- 64 x 512 tiles, 131 KB of
    // LDS, one workgroup per CU: 367 us.  Two launches instead: the GEMM on 64 x 256 tiles with the plain store (two per CU), then the
    // row-wise softmax backward in place (LPM_DA_SPLIT=0: the fused form, A/B)
    static const int da_split = [] { const char* e = getenv("LPM_DA_SPLIT"); return (e && e[0] == '0') ? 0 : 1; }();
    int rc;
    if (da_split && K == 512) {
        g.ldo = K; g.out_batch = (int64_t)T * K; g.out_split = 0; g.accumulate = 0;
        rc = tile_gemm_store(g, B, 1, s, "lpm_vlad_aggregate_bwd_tiles", 2, planes);
        if (rc == LPM_OK) {
            const int64_t rows = (int64_t)B * T;
            hipLaunchKernelGGL(vlad_softmax_bwd_rows_kernel<8>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dassign, assign,
                               planes == 1 ? 1 : 0, scale, shift, ctil, T, rows, sm ? 1 : 0);
        }
    } else {
        rc = tile_gemm_softmax_bwd(g, B, s, "lpm_vlad_aggregate_bwd_tiles", planes);
    }
    if (rc != LPM_OK) return rc;
    if ((residual || g0) && rk) {
        // k-major form: partial sums over clip ranges [z][K][D] -> fixed-order reduce -> [K, D] -> transpose into dcentres [D, K]
        const int64_t n4 = (int64_t)D * K / 4;
        const unsigned wgx = (unsigned)((n4 + 255) / 256);
        const int Z = dcentres_splits(B, D, K), bper = (B + Z - 1) / Z, Zeff = (B + bper - 1) / bper;
        float* part = (float*)(ws + L.dcp);
        float* tmp = (float*)(ws + L.ub2);                    // [K, D] (no second dU tile orientation in this form: the region is free)
        hipLaunchKernelGGL(vlad_bwd_dcentres_k_kernel, dim3(wgx, (unsigned)Zeff), dim3(256), 0, s, dO, nrm, asum, u, v, colsq, B, D, K, bper,
                           part);
        hipLaunchKernelGGL(vlad_bwd_dcentres_reduce_kernel, dim3(wgx), dim3(256), 0, s, (const float4*)part, Zeff, n4, (float4*)tmp);
        launch_kmajor_to_dmajor(tmp, 1, D, K, dcentres, s);
    } else if (residual || g0) {        // dcentres = - sum_b asum_b dU_b: the centres' gradient, and (g0) the input batch norm's beta term
        launch_dcentres(dO, nrm, asum, u, v, B, D, K, (float*)(ws + L.dcp), dcentres, s, colsq_raw, planes == 1 ? 1 : 0, dob);
    }
    return check_launch("lpm_vlad_aggregate_bwd_tiles");
}

// ---- input_bn's gamma / beta gradients from K3's by-products ("no input gradient" mode, see lpm_hip.h) ---------------------
namespace lpm {
// one wave per feature column d:
//   dbeta[d]  = -sum_k dcentres[d,k] + sum_k W[d,k] cs[k]
//   dgamma[d] = (sum_b g0[b,d] + sum_k W[d,k] dW[d,k] - sum_k dcentres[d,k] centres[d,k] - beta[d] dbeta[d]) / gamma[d]
__global__ __launch_bounds__(256) void input_bn_grads_kernel(const float* __restrict__ dcentres, const float* __restrict__ centres,
                                                             const float* __restrict__ W, const float* __restrict__ dW,
                                                             const float* __restrict__ g0, const float* __restrict__ cs,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int B,
                                                             int D, int K, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= D) return;                                    // wave-uniform
    float s1 = 0.f, wc = 0.f, wdw = 0.f, dc = 0.f, g = 0.f;
    for (int k = lane; k < K; k += 64) {
        const int64_t i = (int64_t)d * K + k;
        const float dce = dcentres[i], w = W[i];
        s1 -= dce;
        if (cs) wc = fmaf(w, cs[k], wc);
        wdw = fmaf(w, dW[i], wdw);
        if (centres) dc = fmaf(dce, centres[i], dc);
    }
    for (int b = lane; b < B; b += 64) g += g0[(int64_t)b * D + d];
    s1 = wave_sum(s1); wc = wave_sum(wc); wdw = wave_sum(wdw); dc = wave_sum(dc); g = wave_sum(g);
    if (lane == 0) {
        const float db = s1 + wc;
        dbeta[d] = db;
        dgamma[d] = (g + wdw - dc - beta[d] * db) / gamma[d];
    }
}
}  // namespace lpm

extern "C" int lpm_input_bn_grads(const float* dcentres, const float* centres, const float* W, const float* dW, const float* g0,
                                  const float* colsum_dl, const float* gamma, const float* beta, int B, int D, int K, float* dgamma,
                                  float* dbeta, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dcentres && W && dW && g0 && gamma && beta && dgamma && dbeta, LPM_ERR_BADARG, "lpm_input_bn_grads: null pointer");
    LPM_REQUIRE(B > 0 && D > 0 && K > 0, LPM_ERR_BADARG, "lpm_input_bn_grads: bad sizes");
    hipLaunchKernelGGL(input_bn_grads_kernel, dim3((D + 3) / 4), dim3(256), 0, (hipStream_t)stream, dcentres, centres, W, dW, g0, colsum_dl,
                       gamma, beta, B, D, K, dgamma, dbeta);
    return check_launch("lpm_input_bn_grads");
}

extern "C" int lpm_vlad_aggregate_bwd_tiles_dx(const void* workspace, size_t workspace_bytes, const void* dlr, const void* wtt, int B,
                                               int T, int D, int K, float* dx, int64_t lddx, int accumulate_dx,
                                               lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(workspace && dx, LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd_tiles_dx: null pointer");
    LPM_REQUIRE((dlr == nullptr) == (wtt == nullptr), LPM_ERR_BADARG, "lpm_vlad_aggregate_bwd_tiles_dx: dlr and wtt go together");
    LPM_REQUIRE(B > 0 && T > 0 && D > 0 && D % 32 == 0 && K > 0 && K % 32 == 0 && K <= 512 && lddx >= D, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_bwd_tiles_dx: need D %% 32 == 0, K %% 32 == 0, K <= 512 (D=%d K=%d)", D, K);
    const BwdTilesLayout L = bwd_tiles_layout(B, T, D, K);
    LPM_REQUIRE(workspace_bytes >= L.total, LPM_ERR_WORKSPACE, "lpm_vlad_aggregate_bwd_tiles_dx: workspace too small");
    const char* ws = (const char*)workspace;
    const int KS16 = K / 16, DT = D / 32;
    TileGemmArgs g{};
    g.a = (const uint4*)(ws + L.ar); g.a_tile = (int64_t)KS16 * 128; g.a_step = 128; g.a_batch = (int64_t)L.MT * KS16 * 128;
    g.a_tiles = L.MT;
    g.b = (const uint4*)(ws + L.ub2); g.b_tile = 128; g.b_step = (int64_t)DT * 128; g.b_batch = (int64_t)KS16 * DT * 128; g.b_tiles = DT;
    g.rb_per_batch = L.MT / 2; g.steps_per_split = KS16; g.total_steps = KS16;
    if (dlr) {
        g.a2 = (const uint4*)dlr; g.a2_tile = g.a_tile; g.a2_step = 128; g.a2_batch = g.a_batch;
        g.b2 = (const uint4*)wtt; g.b2_tile = 128; g.b2_step = (int64_t)DT * 128; g.b2_batch = 0; g.b2_tiles = DT;
        g.steps2 = KS16;
    }
    g.out = dx; g.ldo = lddx; g.out_batch = (int64_t)T * lddx; g.rows_valid = T; g.cols_valid = D; g.accumulate = accumulate_dx ? 1 : 0;
    return tile_gemm_store(g, B, 1, (hipStream_t)stream, "lpm_vlad_aggregate_bwd_tiles_dx");
}
