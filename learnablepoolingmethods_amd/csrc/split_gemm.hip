// Split-bf16 operand preparation for the encoder's dense layers (transformer_utils.py:559-561,583,701,708).
//
// The dense GEMMs themselves are plain library GEMMs (hipBLASLt through torch.mm, as the brief prescribes); what
// is hand-written here is the operand format that lets them run on the bf16 matrix pipe WITHOUT leaving the fp32
// parity bar: x = xh + xl, W = Wh + Wl (bf16 planes, 2^-17 residual) and
//     x W  ~=  xh Wh + xl Wh + xh Wl  =  [xh | xl | xh] . [Wh ; Wh ; Wl]
// i.e. ONE bf16 GEMM with a 3x longer reduction and fp32 accumulation inside the GEMM (no partial-sum passes).
//   lpm_split_rows    x [M,K] fp32 (optionally relu(x + bias) fused)  -> X3 [M,3K] bf16 = [hi | lo | hi]
//   lpm_split_weight  W [K,N] fp32 -> W3 [3K,N] = [Wh;Wh;Wl]  and  W3T [3N,K] = [Wh^T;Wh^T;Wl^T] (for dX = dY W^T)
#include "lpm_common.h"

namespace lpm {

__device__ __forceinline__ unsigned sg_bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float sg_bf16_f32(unsigned h) { return __uint_as_float(h << 16); }

// one thread = 8 consecutive columns of one row
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int K,
                                                         const float* __restrict__ bias, int relu,
                                                         unsigned short* __restrict__ out3) {
    const int K8 = K / 8;
    const int64_t total = M * K8;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int64_t m = w / K8;
        const int c = (int)(w % K8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + m * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(x + m * ldx + c + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (bias) {
            const float4 ba = *reinterpret_cast<const float4*>(bias + c), bb = *reinterpret_cast<const float4*>(bias + c + 4);
            v[0] += ba.x; v[1] += ba.y; v[2] += ba.z; v[3] += ba.w;
            v[4] += bb.x; v[5] += bb.y; v[6] += bb.z; v[7] += bb.w;
        }
        unsigned h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (relu) v[e] = fmaxf(v[e], 0.f);
            h[e] = sg_bf16_rne(v[e]);
            l[e] = sg_bf16_rne(v[e] - sg_bf16_f32(h[e]));
        }
        const uint4 hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        const uint4 lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
        unsigned short* row = out3 + m * 3 * (int64_t)K;
        *reinterpret_cast<uint4*>(row + c) = hi;
        *reinterpret_cast<uint4*>(row + K + c) = lo;
        *reinterpret_cast<uint4*>(row + 2 * (int64_t)K + c) = hi;
    }
}

// grid (K/32, N/32); 32x32 tile through LDS for the transposed copy
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, int K, int N,
                                                           unsigned short* __restrict__ w3,
                                                           unsigned short* __restrict__ w3t) {
    __shared__ unsigned short th[32][33], tl[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, n = n0 + tx;
        unsigned h = 0, l = 0;
        if (k < K && n < N) {
            const float v = W[(int64_t)k * N + n];
            h = sg_bf16_rne(v);
            l = sg_bf16_rne(v - sg_bf16_f32(h));
            w3[(int64_t)k * N + n] = (unsigned short)h;
            w3[((int64_t)K + k) * N + n] = (unsigned short)h;
            w3[(2 * (int64_t)K + k) * N + n] = (unsigned short)l;
        }
        th[ty + 8 * i][tx] = (unsigned short)h;
        tl[ty + 8 * i][tx] = (unsigned short)l;
    }
    __syncthreads();
    if (w3t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + ty + 8 * i, k = k0 + tx;
            if (k < K && n < N) {
                const unsigned short h = th[tx][ty + 8 * i], l = tl[tx][ty + 8 * i];
                w3t[(int64_t)n * K + k] = h;
                w3t[((int64_t)N + n) * K + k] = h;
                w3t[(2 * (int64_t)N + n) * K + k] = l;
            }
        }
    }
}

}  // namespace lpm

extern "C" int lpm_split_rows(const float* x, int64_t ldx, int64_t M, int K, const float* bias, int relu, void* out3,
                              lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && out3, LPM_ERR_BADARG, "lpm_split_rows: null pointer");
    LPM_REQUIRE(M > 0 && K > 0 && ldx >= K, LPM_ERR_BADARG, "lpm_split_rows: bad sizes");
    LPM_REQUIRE(K % 8 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)out3 | (uintptr_t)bias) & 15) == 0,
                LPM_ERR_UNSUPPORTED_SHAPE, "lpm_split_rows: need K %% 8 == 0, ldx %% 4 == 0, 16-byte aligned pointers (K=%d)", K);
    const int64_t total = M * (K / 8);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       M, K, bias, relu, (unsigned short*)out3);
    return check_launch("lpm_split_rows");
}

extern "C" int lpm_split_weight(const float* W, int K, int N, void* w3, void* w3t, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(W && w3, LPM_ERR_BADARG, "lpm_split_weight: null pointer");
    LPM_REQUIRE(K > 0 && N > 0, LPM_ERR_BADARG, "lpm_split_weight: bad sizes");
    hipLaunchKernelGGL(split_weight_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, (hipStream_t)stream, W, K, N,
                       (unsigned short*)w3, (unsigned short*)w3t);
    return check_launch("lpm_split_weight");
}
