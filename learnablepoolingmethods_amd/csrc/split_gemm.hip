// Split-bf16 operand preparation for the encoder's dense layers (transformer_utils.py:559-561,583,701,708).
//
// The dense GEMMs themselves are plain library GEMMs (hipBLASLt through torch.mm, as the brief prescribes); what
// is hand-written here is the operand format that lets them run on the bf16 matrix pipe WITHOUT leaving the fp32
// parity bar: x = xh + xl, W = Wh + Wl (bf16 planes, 2^-17 residual) and
//     x W  ~=  xh Wh + xl Wh + xh Wl  =  [xh | xl | xh] . [Wh ; Wh ; Wl]      (the weight image is stored transposed)
// i.e. ONE bf16 GEMM with a 3x longer reduction and fp32 accumulation inside the GEMM (no partial-sum passes).
//   lpm_split_rows    x [M,K] fp32 (optionally relu(x + bias) fused)  -> X3 [M,3K] bf16 = [hi | lo | hi]   (activations)
//                                                                      or  [hi | hi | lo]   (gradients, order = 1)
//   lpm_split_weight  W [K,N] fp32 -> w3n [N,3K] (rows [Wh^T|Wh^T|Wl^T]: y = X3 w3n^T)  and
//                                      w3k [K,3N] (rows [Wh|Wl|Wh]:       dx = DY3 w3k^T)
// The two plane orders pair up row by row: seen as [3M, K] and [3M, N] matrices (row 3m+p = plane p of row m), an
// activation image and a gradient image give the weight gradient as ONE long-reduction GEMM
//     dW = X3[3M,K]^T . DY3[3M,N] = xh^T dyh + xl^T dyh + xh^T dyl
// (a 61440-deep reduction at cfg-2: hipBLASLt runs it at ~2x the rate of three separate 20480-deep GEMMs).
// Round 5: every image here also exists in the fp16 two-product format (operand_format.h): [hi | lo] planes of v * scale.
#include "lpm_common.h"
#include "operand_format.h"

namespace lpm {

__device__ __forceinline__ unsigned sg_bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float sg_bf16_f32(unsigned h) { return __uint_as_float(h << 16); }

// one thread = 8 consecutive columns of one row
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int K,
                                                         const float* __restrict__ bias, int relu, int order,
                                                         unsigned short* __restrict__ out3, const float* __restrict__ row_scale,
                                                         const OperandFmt fmt) {
    const int K8 = K / 8;
    const int64_t total = M * K8;
    float vmax = 0.f;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int64_t m = w / K8;
        const int c = (int)(w % K8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + m * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(x + m * ldx + c + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (row_scale) {                       // a lazily normalised pooled descriptor: raw sums times one factor per row
            const float rs = row_scale[m];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= rs;
        }
        if (bias) {
            const float4 ba = *reinterpret_cast<const float4*>(bias + c), bb = *reinterpret_cast<const float4*>(bias + c + 4);
            v[0] += ba.x; v[1] += ba.y; v[2] += ba.z; v[3] += ba.w;
            v[4] += bb.x; v[5] += bb.y; v[6] += bb.z; v[7] += bb.w;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        vmax = of_amax8(vmax, v);
        uint4 hi, lo;
        if (fmt.f16) {
            of_split8(v, 1, fmt.scale, hi, lo);
        } else {
            unsigned h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = sg_bf16_rne(v[e]);
                l[e] = sg_bf16_rne(v[e] - sg_bf16_f32(h[e]));
            }
            hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
        }
        of_store_row8(out3 + m * of_row_stride(K, fmt.planes), K, c, hi, lo, fmt.planes, order);
    }
    of_amax_commit(fmt.amax, vmax);
}

// Backward companion of the fused relu(x + bias) split: g = df * [act > 0] where `act3` is the [M,3K] split image of
// the forward activation (its hi plane is > 0 exactly where the activation was), out3 = split(g) in the gradient plane
// order [hi | hi | lo], and per-block
// column partial sums of g (the bias gradient) -> colpart [gridDim.x][K].  One workgroup = SR_ROWS rows, all columns.
constexpr int SR_ROWS = 32;
__global__ __launch_bounds__(256) void split_rows_relu_bwd_kernel(const float* __restrict__ df, int64_t M, int K,
                                                                  const unsigned short* __restrict__ act3, int act_planes,
                                                                  unsigned short* __restrict__ out3,
                                                                  float* __restrict__ colpart, float alpha, const OperandFmt fmt) {
    const int K8 = K / 8;
    float vmax = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * SR_ROWS, r1 = min(M, r0 + SR_ROWS);
    for (int cg = threadIdx.x; cg < K8; cg += 256) {
        const int c = cg * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t m = r0; m < r1; ++m) {
            const float4 a = *reinterpret_cast<const float4*>(df + m * K + c);
            const float4 b = *reinterpret_cast<const float4*>(df + m * K + c + 4);
            const uint4 hm = *reinterpret_cast<const uint4*>(act3 + m * of_row_stride(K, act_planes) + c);   // hi plane of the activation
            float v[8] = {a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha, b.x * alpha, b.y * alpha, b.z * alpha, b.w * alpha};
            const unsigned mw[4] = {hm.x, hm.y, hm.z, hm.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned ah = (mw[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                v[e] = of_positive(ah) ? v[e] : 0.f;                            // activation > 0
                acc[e] += v[e];
            }
            vmax = of_amax8(vmax, v);
            uint4 hi, lo;
            if (fmt.f16) {
                of_split8(v, 1, fmt.scale, hi, lo);
            } else {
                unsigned h[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = sg_bf16_rne(v[e]);
                    l[e] = sg_bf16_rne(v[e] - sg_bf16_f32(h[e]));
                }
                hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            }
            of_store_row8(out3 + m * of_row_stride(K, fmt.planes), K, c, hi, lo, fmt.planes, 1);     // (three planes: gradient order [hi | hi | lo])
        }
        float* cp = colpart + (int64_t)blockIdx.x * K + c;
        *reinterpret_cast<float4*>(cp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(cp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    of_amax_commit(fmt.amax, vmax);
}

// colpart [nblk][K] -> out [K]  (fp64 accumulation; 1024 threads per 16 columns, partial_colsums16)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ colpart, int nblk, int K,
                                                             float* __restrict__ out) {
    double s, q;
    int c;
    partial_colsums16(colpart, nblk, (int64_t)K, 0, K, s, q, c);
    if (threadIdx.x < 16 && c < K) out[c] = (float)s;
}

// grid (K/32, N/32); 32x32 tile through LDS for the transposed image.
//   w3n [N, 3K]: row n = [Wh[:,n] | Wh[:,n] | Wl[:,n]]   (forward: y = X3 . w3n^T, X3 = [xh|xl|xh])
//   w3k [K, 3N]: row k = [Wh[k,:] | Wl[k,:] | Wh[k,:]]   (input gradient: dx = DY3 . w3k^T, DY3 = [dyh|dyh|dyl])
// Both are "B stored transposed" (NT) operands: hipBLASLt runs that layout 8-10 % faster than NN at these shapes.
// f16: fp16 planes -- wn [N, 3K] = [Wh^T | Wh^T | Wl^T] (the forward's three-term product), wk [K, 2N] = [Wh | Wh] (the input gradient's
// two-term product: the weight rounded once).
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, int K, int N,
                                                           unsigned short* __restrict__ w3n,
                                                           unsigned short* __restrict__ w3k, int f16) {
    __shared__ unsigned short th[32][33], tl[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, n = n0 + tx;
        unsigned h = 0, l = 0;
        if (k < K && n < N) {
            const float v = W[(int64_t)k * N + n];
            if (f16) {
                of_split1_f16(v, h, l);
                if (w3k) {
                    unsigned short* row = w3k + (int64_t)k * 2 * N;
                    row[n] = (unsigned short)h;
                    row[N + n] = (unsigned short)h;
                }
            } else {
                h = sg_bf16_rne(v);
                l = sg_bf16_rne(v - sg_bf16_f32(h));
                if (w3k) {
                    unsigned short* row = w3k + (int64_t)k * 3 * N;
                    row[n] = (unsigned short)h;
                    row[N + n] = (unsigned short)l;
                    row[2 * N + n] = (unsigned short)h;
                }
            }
        }
        th[ty + 8 * i][tx] = (unsigned short)h;
        tl[ty + 8 * i][tx] = (unsigned short)l;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        if (k < K && n < N) {
            const unsigned short h = th[tx][ty + 8 * i], l = tl[tx][ty + 8 * i];
            unsigned short* row = w3n + (int64_t)n * 3 * K;
            row[k] = h;
            row[K + k] = h;
            row[2 * K + k] = l;
        }
    }
}

}  // namespace lpm

extern "C" int lpm_split_rows_fmt(const float* x, int64_t ldx, int64_t M, int K, const float* bias, int relu, int order, const float* row_scale,
                                  void* out3, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && out3, LPM_ERR_BADARG, "lpm_split_rows: null pointer");
    LPM_REQUIRE(M > 0 && K > 0 && ldx >= K, LPM_ERR_BADARG, "lpm_split_rows: bad sizes");
    LPM_REQUIRE(K % 8 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)out3 | (uintptr_t)bias) & 15) == 0,
                LPM_ERR_UNSUPPORTED_SHAPE, "lpm_split_rows: need K %% 8 == 0, ldx %% 4 == 0, 16-byte aligned pointers (K=%d)", K);
    if (const int rc = operand_fmt_check(fmt, "lpm_split_rows")) return rc;
    const int64_t total = M * (K / 8);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       M, K, bias, relu, order ? 1 : 0, (unsigned short*)out3, row_scale, operand_fmt(fmt));
    return check_launch("lpm_split_rows");
}
extern "C" int lpm_split_rows(const float* x, int64_t ldx, int64_t M, int K, const float* bias, int relu, int order, void* out3,
                              lpm_stream_t stream) {
    return lpm_split_rows_fmt(x, ldx, M, K, bias, relu, order, nullptr, out3, nullptr, stream);
}

// x[m, :] * row_scale[m] -> activation image [hi | lo | hi]: the operand of the q/k/v GEMM when x is the pooled descriptor in its
// lazily normalised form (lpm_vlad_aggregate_raw_kmajor_fwd + lpm_vlad_row_scales)
extern "C" int lpm_split_rows_scaled(const float* x, int64_t ldx, int64_t M, int K, const float* row_scale, void* out3, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && out3 && row_scale, LPM_ERR_BADARG, "lpm_split_rows_scaled: null pointer");
    LPM_REQUIRE(M > 0 && K > 0 && ldx >= K, LPM_ERR_BADARG, "lpm_split_rows_scaled: bad sizes");
    LPM_REQUIRE(K % 8 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)out3) & 15) == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_split_rows_scaled: need K %% 8 == 0, ldx %% 4 == 0, 16-byte aligned pointers (K=%d)", K);
    return lpm_split_rows_fmt(x, ldx, M, K, nullptr, 0, 0, row_scale, out3, nullptr, stream);
}

namespace lpm {
// ---- bias (+ ReLU) of a dense layer whose output goes to something other than the next GEMM's operand split (tf.layers.dense with
// use_bias / activation=relu, e.g. FeedForwardNetworkMod's layers in front of their batch norms, transformer_utils.py:741-760) ----------
// forward: y <- act(y + bias) in place, one pass;  backward: dx = dy * [y > 0] (ReLU; y is the saved OUTPUT) and per-32-row column
// partial sums of dx for the bias gradient (then colsum_reduce_kernel), one pass -- instead of add + relu and threshold + reduce.
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(float* __restrict__ y, const float* __restrict__ bias, int relu, int64_t total4,
                                                           int C4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float4 b = reinterpret_cast<const float4*>(bias)[i % C4];
        float4 v = reinterpret_cast<float4*>(y)[i];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}
// one workgroup = SR_ROWS rows, all columns (4 per thread and pass); dx may alias dy
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int relu, int64_t M, int C,
                                                           float* __restrict__ dx, float* __restrict__ colpart) {
    const int C4 = C / 4;
    const int64_t r0 = (int64_t)blockIdx.x * SR_ROWS, r1 = min(M, r0 + SR_ROWS);
    for (int c4 = threadIdx.x; c4 < C4; c4 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t m = r0;
        for (; m + 3 < r1; m += 4) {                 // four rows' loads together
            float4 g[4], a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = reinterpret_cast<const float4*>(dy + (m + u) * C)[c4];
                a[u] = relu ? reinterpret_cast<const float4*>(y + (m + u) * C)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 d = make_float4(a[u].x > 0.f ? g[u].x : 0.f, a[u].y > 0.f ? g[u].y : 0.f, a[u].z > 0.f ? g[u].z : 0.f,
                                             a[u].w > 0.f ? g[u].w : 0.f);
                acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
                if (relu) reinterpret_cast<float4*>(dx + (m + u) * C)[c4] = d;
            }
        }
        for (; m < r1; ++m) {
            const float4 g = reinterpret_cast<const float4*>(dy + m * C)[c4];
            const float4 a = relu ? reinterpret_cast<const float4*>(y + m * C)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 d = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
            acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
            if (relu) reinterpret_cast<float4*>(dx + m * C)[c4] = d;
        }
        reinterpret_cast<float4*>(colpart + (int64_t)blockIdx.x * C)[c4] = acc;
    }
}
}  // namespace lpm

extern "C" size_t lpm_split_rows_relu_bwd_workspace_bytes(int64_t M, int K) {
    return (size_t)((M + lpm::SR_ROWS - 1) / lpm::SR_ROWS) * K * sizeof(float);
}

extern "C" int lpm_split_rows_relu_bwd(const float* df, int64_t M, int K, const void* act3, void* out3, float* dbias,
                                       void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    return lpm_split_rows_relu_bwd_fmt(df, M, K, 1.f, act3, LPM_OPERAND_BF16X3, out3, dbias, workspace, workspace_bytes, nullptr, stream);
}
extern "C" int lpm_split_rows_relu_bwd_fmt(const float* df, int64_t M, int K, float alpha, const void* act3, int act_kind, void* out3, float* dbias,
                                           void* workspace, size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    if (const int rc = operand_fmt_check(fmt, "lpm_split_rows_relu_bwd")) return rc;
    LPM_REQUIRE(alpha > 0.f && operand_kind_ok(act_kind), LPM_ERR_BADARG,
                "lpm_split_rows_relu_bwd: bad alpha / act_kind");
    LPM_REQUIRE(df && act3 && out3 && dbias && workspace, LPM_ERR_BADARG, "lpm_split_rows_relu_bwd: null pointer");
    LPM_REQUIRE(M > 0 && K > 0 && K % 8 == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_split_rows_relu_bwd: need K %% 8 == 0 (K=%d)", K);
    LPM_REQUIRE(workspace_bytes >= lpm_split_rows_relu_bwd_workspace_bytes(M, K), LPM_ERR_WORKSPACE,
                "lpm_split_rows_relu_bwd: workspace too small");
    const int nblk = (int)((M + SR_ROWS - 1) / SR_ROWS);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(split_rows_relu_bwd_kernel, dim3(nblk), dim3(256), 0, s, df, M, K, (const unsigned short*)act3,
                       operand_kind_planes(act_kind), (unsigned short*)out3, (float*)workspace, alpha, operand_fmt(fmt));
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((K + 15) / 16), dim3(1024), 0, s, (const float*)workspace, nblk, K, dbias);
    return check_launch("lpm_split_rows_relu_bwd");
}

extern "C" int lpm_split_weight_fmt(const float* W, int K, int N, void* w3n, void* w3k, int kind, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(W && w3n, LPM_ERR_BADARG, "lpm_split_weight: null pointer");
    LPM_REQUIRE(K > 0 && N > 0, LPM_ERR_BADARG, "lpm_split_weight: bad sizes");
    LPM_REQUIRE(operand_kind_ok(kind), LPM_ERR_BADARG, "lpm_split_weight: unknown operand format %d", kind);
    hipLaunchKernelGGL(split_weight_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, (hipStream_t)stream, W, K, N,
                       (unsigned short*)w3n, (unsigned short*)w3k, operand_kind_f16(kind));
    return check_launch("lpm_split_weight");
}
extern "C" int lpm_split_weight(const float* W, int K, int N, void* w3n, void* w3k, lpm_stream_t stream) {
    return lpm_split_weight_fmt(W, K, N, w3n, w3k, LPM_OPERAND_BF16X3, stream);
}

extern "C" int lpm_bias_act_fwd(float* y, const float* bias, int relu, int64_t M, int C, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(y && bias, LPM_ERR_BADARG, "lpm_bias_act_fwd: null pointer");
    LPM_REQUIRE(M > 0 && C > 0 && C % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_bias_act_fwd: need C %% 4 == 0 (C=%d)", C);
    LPM_REQUIRE((((uintptr_t)y | (uintptr_t)bias) & 15) == 0, LPM_ERR_BADARG, "lpm_bias_act_fwd: pointers must be 16-byte aligned");
    const int64_t total4 = M * (C / 4);
    const int64_t want = (total4 + 255) / 256;
    hipLaunchKernelGGL(bias_act_fwd_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, y, bias, relu, total4,
                       C / 4);
    return check_launch("lpm_bias_act_fwd");
}
extern "C" size_t lpm_bias_act_bwd_workspace_bytes(int64_t M, int C) { return (size_t)((M + lpm::SR_ROWS - 1) / lpm::SR_ROWS) * C * sizeof(float); }
/* y: the forward's OUTPUT (only read when relu); dx: dy masked (not written without relu: the gradient passes through, dx may be NULL);
 * dbias [C] = column sums of the masked gradient */
extern "C" int lpm_bias_act_bwd(const float* dy, const float* y, int relu, int64_t M, int C, float* dx, float* dbias, void* workspace,
                                size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dy && dbias && workspace && (!relu || (y && dx)), LPM_ERR_BADARG, "lpm_bias_act_bwd: null pointer");
    LPM_REQUIRE(M > 0 && C > 0 && C % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_bias_act_bwd: need C %% 4 == 0 (C=%d)", C);
    LPM_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)workspace) & 15) == 0, LPM_ERR_BADARG,
                "lpm_bias_act_bwd: pointers must be 16-byte aligned");
    LPM_REQUIRE(workspace_bytes >= lpm_bias_act_bwd_workspace_bytes(M, C), LPM_ERR_WORKSPACE, "lpm_bias_act_bwd: workspace too small");
    const int nblk = (int)((M + SR_ROWS - 1) / SR_ROWS);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(nblk), dim3(256), 0, s, dy, y, relu, M, C, dx, (float*)workspace);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, (const float*)workspace, nblk, C, dbias);
    return check_launch("lpm_bias_act_bwd");
}
