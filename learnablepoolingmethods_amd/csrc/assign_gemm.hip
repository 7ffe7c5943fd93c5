// K1 -- soft-assignment GEMM with batch-norm statistics epilogue.
//   logits[M,K] = x[M,D] . w[D,K]     (reference: tf.matmul, frame_level_models.py:2781)
//   partial[blk, 0, k] = sum_rows logits, partial[blk, 1, k] = sum_rows logits^2
//                                       (the reduction half of cluster_bn, :2783-2789)
//
// gfx950 mapping: 256-thread workgroups (4 waves, one per SIMD) own a 64 x (64*NT) output tile;
// waves sit 2 (rows) x 2 (cols), each accumulating NT 32x32 tiles with v_mfma_f32_32x32x2_f32
// (exact fp32 -- bitwise an fmaf chain -- so the 1e-3 parity bar is met with ~4 decades to spare).
// x and w stream through a register-staged, double-buffered LDS pipeline (one barrier per
// 16-deep K-step); x is read from HBM exactly once when K <= 256, w (<= 1 MB) stays L2-resident.
#include "lpm_common.h"

namespace lpm {

constexpr int AG_BM = 64;   // rows per block
constexpr int AG_BK = 16;   // reduction depth per LDS stage

template <int NT>
__global__ __launch_bounds__(256) void assign_gemm_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ w, int M, int D, int K,
                                                              float* __restrict__ logits,
                                                              float* __restrict__ partial) {
    constexpr int BN = 64 * NT;
    constexpr int XS = AG_BK + 1;  // padded row stride: A-operand column reads hit 32 distinct banks
    __shared__ float xs[2][AG_BM * XS];
    __shared__ float ws[2][AG_BK * BN];
    __shared__ float red[2][2][BN];  // [sum|sumsq][wave row][col]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * AG_BM, n0 = blockIdx.y * BN;

    // global->register staging coordinates
    const int xr = tid >> 2, xc = (tid & 3) * 4;             // x: 64 rows x 16 floats, one float4 per thread
    const bool xvalid = (m0 + xr) < M;
    const float* xp = x + (int64_t)(m0 + xr) * ldx + xc;
    float4 xreg;
    float4 wreg[NT];

    auto gload = [&](int k0) {
        xreg = xvalid ? *reinterpret_cast<const float4*>(xp + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = tid + i * 256;                    // float4 index inside the 16 x BN tile
            const int r = f / (BN / 4), c = (f % (BN / 4)) * 4;
            const int col = n0 + c;
            wreg[i] = (col < K) ? *reinterpret_cast<const float4*>(w + (int64_t)(k0 + r) * K + col)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
        float* xd = &xs[buf][xr * XS + xc];
        xd[0] = xreg.x; xd[1] = xreg.y; xd[2] = xreg.z; xd[3] = xreg.w;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = tid + i * 256;
            *reinterpret_cast<float4*>(&ws[buf][f * 4]) = wreg[i];
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = D / AG_BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload((c + 1) * AG_BK);
        const float* xa = &xs[buf][(wm * 32 + (lane & 31)) * XS + (lane >> 5)];
        const float* wb = &ws[buf][(lane >> 5) * BN + wn * (NT * 32) + (lane & 31)];
#pragma unroll
        for (int kk = 0; kk < AG_BK; kk += 2) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(a, wb[kk * BN + t * 32], acc[t]);
        }
        if (c + 1 < nchunk) sstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: logits + per-column partial statistics
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cb = wn * (NT * 32) + t * 32 + (lane & 31);   // column inside the block tile
        const int col = n0 + cb;
        float cs = 0.f, cq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + mfma32_row(r, lane);
            const float v = acc[t][r];
            if (row < M && col < K) logits[(int64_t)row * K + col] = v;
            cs += v;
            cq += v * v;
        }
        cs += __shfl_xor(cs, 32, 64);
        cq += __shfl_xor(cq, 32, 64);
        if (lane < 32) {
            red[0][wm][cb] = cs;
            red[1][wm][cb] = cq;
        }
    }
    __syncthreads();
    for (int cb = tid; cb < BN; cb += 256) {
        const int col = n0 + cb;
        if (col < K) {
            float* p = partial + (int64_t)blockIdx.x * 2 * K;
            p[col] = red[0][0][cb] + red[0][1][cb];
            p[K + col] = red[1][0][cb] + red[1][1][cb];
        }
    }
}

}  // namespace lpm

extern "C" int lpm_assign_gemm_nblk(int M) { return (M + lpm::AG_BM - 1) / lpm::AG_BM; }

extern "C" int lpm_assign_gemm_fwd(const float* x, int64_t ldx, const float* w, int M, int D, int K, int precision,
                                   float* logits, float* partial, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && w && logits && partial, LPM_ERR_BADARG, "lpm_assign_gemm_fwd: null pointer");
    LPM_REQUIRE(M > 0 && D > 0 && K > 0 && ldx >= D, LPM_ERR_BADARG, "lpm_assign_gemm_fwd: bad sizes M=%d D=%d K=%d ldx=%lld",
                M, D, K, (long long)ldx);
    LPM_REQUIRE(D % AG_BK == 0 && K % 4 == 0 && ldx % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_assign_gemm_fwd: need D %% 16 == 0, K %% 4 == 0, ldx %% 4 == 0 (D=%d K=%d ldx=%lld)", D, K,
                (long long)ldx);
    LPM_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, LPM_ERR_BADARG, "lpm_assign_gemm_fwd: x/w must be 16-byte aligned");
    LPM_REQUIRE(precision == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_assign_gemm_fwd: precision %d not built", precision);
    hipStream_t s = (hipStream_t)stream;
    const int nb = lpm_assign_gemm_nblk(M);
    if (K > 128) {
        dim3 grid(nb, (K + 255) / 256);
        hipLaunchKernelGGL(assign_gemm_f32_kernel<4>, grid, dim3(256), 0, s, x, ldx, w, M, D, K, logits, partial);
    } else if (K > 64) {
        hipLaunchKernelGGL(assign_gemm_f32_kernel<2>, dim3(nb, 1), dim3(256), 0, s, x, ldx, w, M, D, K, logits, partial);
    } else {
        hipLaunchKernelGGL(assign_gemm_f32_kernel<1>, dim3(nb, 1), dim3(256), 0, s, x, ldx, w, M, D, K, logits, partial);
    }
    return check_launch("lpm_assign_gemm_fwd");
}
