// a2 + a3 -- SampleUniformFrames fused with the input batch-norm.
//   idx[b,j] = int32(fp32(j * fp32(1/S)) * fp32(num_frames[b]))      model_utils.py:112-118
//   y[b*S+j, :] = raw[b, idx[b,j], :] * scale + shift                 model_utils.py:119-122 +
//                                                                      frame_level_models.py:2265-2271
// The gathered [B,S,F] tensor is never materialised un-normalised: one pass reduces the column
// statistics of the gathered rows (per-block partials -> lpm_bn_fold), a second pass gathers again
// and writes the normalised rows that K1 and K2 consume.  Both are pure HBM streams (float4,
// row-contiguous 4.6 KB reads).
#include "lpm_common.h"

namespace lpm {

constexpr int FP_ROWS = 32;  // gathered rows per statistics block

__device__ __forceinline__ int sample_index(int j, float step, int nf) {
    // fp32 product then truncation toward zero, exactly as tf.linspace * num_frames -> tf.cast(int32)
    const float v = __fmul_rn((float)j, step);
    return (int)__fmul_rn(fminf(v, 1.0f), (float)nf);
}

__global__ __launch_bounds__(256) void frame_stats_kernel(const float* __restrict__ raw,
                                                          const int32_t* __restrict__ num_frames, int B,
                                                          int max_frames, int F, int S, float step,
                                                          float* __restrict__ partial) {
    const int r0 = blockIdx.x * FP_ROWS;
    const int rows = B * S;
    const int r1 = min(rows, r0 + FP_ROWS);
    __shared__ int64_t base[FP_ROWS];
    if (threadIdx.x < FP_ROWS) {
        const int r = r0 + threadIdx.x;
        if (r < rows) {
            const int b = r / S, j = r % S;
            int idx = sample_index(j, step, num_frames[b]);
            idx = max(0, min(idx, max_frames - 1));
            base[threadIdx.x] = ((int64_t)b * max_frames + idx) * F;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += 256) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < r1 - r0; ++r) {
            const float v = raw[base[r] + c];
            s += v;
            q = fmaf(v, v, q);
        }
        float* p = partial + (int64_t)blockIdx.x * 2 * F;
        p[c] = s;
        p[F + c] = q;
    }
}

__global__ __launch_bounds__(256) void frame_apply_kernel(const float* __restrict__ raw,
                                                          const int32_t* __restrict__ num_frames, int B,
                                                          int max_frames, int F, int S, float step,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ y) {
    const int F4 = F / 4;
    const int64_t total = (int64_t)B * S * F4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / F4), c = (int)(i % F4) * 4;
        const int b = r / S, j = r % S;
        int idx = sample_index(j, step, num_frames[b]);
        idx = max(0, min(idx, max_frames - 1));
        float4 v = *reinterpret_cast<const float4*>(raw + ((int64_t)b * max_frames + idx) * F + c);
        if (scale) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + c);
            const float4 sh = *reinterpret_cast<const float4*>(shift + c);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        *reinterpret_cast<float4*>(y + (int64_t)r * F + c) = v;
    }
}

// column partials of (sum dy, sum dy * x) over the gathered rows, for dgamma/dbeta of input_bn
__global__ __launch_bounds__(256) void frame_bn_bwd_partial_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                   const float* __restrict__ raw,
                                                                   const int32_t* __restrict__ num_frames, int B,
                                                                   int max_frames, int F, int S, float step,
                                                                   float* __restrict__ partial) {
    const int r0 = blockIdx.x * FP_ROWS;
    const int rows = B * S;
    const int r1 = min(rows, r0 + FP_ROWS);
    __shared__ int64_t base[FP_ROWS];
    if (threadIdx.x < FP_ROWS) {
        const int r = r0 + threadIdx.x;
        if (r < rows) {
            const int b = r / S, j = r % S;
            int idx = sample_index(j, step, num_frames[b]);
            idx = max(0, min(idx, max_frames - 1));
            base[threadIdx.x] = ((int64_t)b * max_frames + idx) * F;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += 256) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < r1 - r0; ++r) {
            const float g = dy[(int64_t)(r0 + r) * lddy + c];
            s += g;
            q = fmaf(g, raw[base[r] + c], q);
        }
        float* p = partial + (int64_t)blockIdx.x * 2 * F;
        p[c] = s;
        p[F + c] = q;
    }
}

__global__ __launch_bounds__(1024) void frame_bn_bwd_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                                   int F, const float* __restrict__ mean,
                                                                   const float* __restrict__ var, float eps,
                                                                   float* dgamma, float* dbeta) {
    __shared__ double sh[2][16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, q = 0.0;
    if (c < F) {
        for (int b = rg; b < nblk; b += 16) {
            const float* p = partial + (int64_t)b * 2 * F;
            s += (double)p[c];
            q += (double)p[F + c];
        }
    }
    sh[0][rg][cl] = s;
    sh[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < F) {
        for (int i = 1; i < 16; ++i) {
            s += sh[0][i][cl];
            q += sh[1][i][cl];
        }
        // sum dy * xhat = rstd * (sum dy*x - mean * sum dy)
        const double rstd = 1.0 / sqrt((double)var[c] + (double)eps);
        dbeta[c] = (float)s;
        dgamma[c] = (float)(rstd * (q - (double)mean[c] * s));
    }
}

}  // namespace lpm

static inline int fp_nblk(int B, int S) { return (B * S + lpm::FP_ROWS - 1) / lpm::FP_ROWS; }

extern "C" size_t lpm_frame_stats_workspace_bytes(int B, int S, int F) {
    return (size_t)fp_nblk(B, S) * 2 * F * sizeof(float);
}
extern "C" int lpm_frame_stats_nblk(int B, int S) { return fp_nblk(B, S); }

#define LPM_FRAME_CHECK(name)                                                                                       \
    LPM_REQUIRE(raw && num_frames, LPM_ERR_BADARG, name ": null pointer");                                          \
    LPM_REQUIRE(B > 0 && max_frames > 0 && F > 0 && S > 0, LPM_ERR_BADARG, name ": bad sizes");                      \
    LPM_REQUIRE(F % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE, name ": need F %% 4 == 0 (F=%d)", F)

extern "C" int lpm_frame_stats(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                               float* partial, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_stats");
    LPM_REQUIRE(partial, LPM_ERR_BADARG, "lpm_frame_stats: null workspace");
    const float step = 1.0f / (float)S;
    hipLaunchKernelGGL(frame_stats_kernel, dim3(fp_nblk(B, S)), dim3(256), 0, (hipStream_t)stream, raw, num_frames, B,
                       max_frames, F, S, step, partial);
    return check_launch("lpm_frame_stats");
}

extern "C" int lpm_frame_apply(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                               const float* scale, const float* shift, float* y, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_apply");
    LPM_REQUIRE(y && ((scale == nullptr) == (shift == nullptr)), LPM_ERR_BADARG, "lpm_frame_apply: bad pointers");
    const float step = 1.0f / (float)S;
    const int64_t total = (int64_t)B * S * (F / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(frame_apply_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, raw,
                       num_frames, B, max_frames, F, S, step, scale, shift, y);
    return check_launch("lpm_frame_apply");
}

extern "C" int lpm_frame_bn_bwd(const float* dy, int64_t lddy, const float* raw, const int32_t* num_frames, int B,
                                int max_frames, int F, int S, const float* mean, const float* var, float eps,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_bn_bwd");
    LPM_REQUIRE(dy && mean && var && dgamma && dbeta && workspace && lddy >= F, LPM_ERR_BADARG, "lpm_frame_bn_bwd: bad pointers");
    LPM_REQUIRE(workspace_bytes >= lpm_frame_stats_workspace_bytes(B, S, F), LPM_ERR_WORKSPACE, "lpm_frame_bn_bwd: workspace too small");
    const float step = 1.0f / (float)S;
    const int nblk = fp_nblk(B, S);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_bn_bwd_partial_kernel, dim3(nblk), dim3(256), 0, s, dy, lddy, raw, num_frames, B, max_frames, F, S,
                       step, (float*)workspace);
    hipLaunchKernelGGL(frame_bn_bwd_reduce_kernel, dim3((F + 63) / 64), dim3(1024), 0, s, (const float*)workspace, nblk, F, mean,
                       var, eps, dgamma, dbeta);
    return check_launch("lpm_frame_bn_bwd");
}
