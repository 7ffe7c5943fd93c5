// a14 + a15 -- per-variable clip_by_norm fused with TF1-style Adam over a flat parameter arena.
//   utils.clip_gradient_norms (utils.py:170-189): g *= c / max(||g||_2, c), per variable
//   tf.train.AdamOptimizer (train.py:252,336): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
//                                              p -= lr_t * m / (sqrt(v) + eps)
// Every variable starts at a multiple of LPM_ARENA_ALIGN floats inside the arena, so each
// 4096-float chunk belongs to exactly one variable: pass 1 writes one partial sum of squares per
// chunk, pass 2 reduces each variable's chunks in a fixed order (deterministic norms), pass 3
// streams p/g/m/v once as float4 (HBM-bound: 28 B per parameter).
#include "lpm_common.h"

namespace lpm {

constexpr int CA_CHUNK = 4096;   // == LPM_ARENA_ALIGN

// which variable owns the chunk that starts at `base`: binary search on the (chunk-aligned) offsets
__device__ __forceinline__ int ca_owner(const int64_t* __restrict__ offsets, int ntensors, int64_t base) {
    int lo = 0, hi = ntensors;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= base) lo = mid; else hi = mid;
    }
    return lo;
}
// l2 (optional, [ntensors]): the analytic gradient of a variable's L2 penalty, coefficient * w (slim.l2_regularizer on the MoE weights,
// video_level_models.py:84-100: part of the loss whose gradient is clipped), formed HERE and in ca_apply_kernel from the parameter both
// passes can read, instead of by an add pass over the gradient arena before them (round 6: 68 us per cfg-5 step for two MoE matrices)
__global__ __launch_bounds__(256) void ca_chunk_sumsq_kernel(const float* __restrict__ g, int64_t total,
                                                             float* __restrict__ chunk_ss, const float* __restrict__ p,
                                                             const int64_t* __restrict__ offsets, int ntensors,
                                                             const float* __restrict__ l2) {
    const int64_t base = (int64_t)blockIdx.x * CA_CHUNK;
    const float c = l2 ? l2[ca_owner(offsets, ntensors, base)] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CA_CHUNK / (256 * 4); ++i) {
        const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
        if (e + 3 < total) {
            float4 v = *reinterpret_cast<const float4*>(g + e);
            if (c != 0.f) {
                const float4 w = *reinterpret_cast<const float4*>(p + e);
                v.x = fmaf(c, w.x, v.x); v.y = fmaf(c, w.y, v.y); v.z = fmaf(c, w.z, v.z); v.w = fmaf(c, w.w, v.w);
            }
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    }
    s = wave_sum(s);
    __shared__ float w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) chunk_ss[blockIdx.x] = (w[0] + w[1]) + (w[2] + w[3]);
}

// one block per variable: factor[t] = clip / max(||g_t||, clip)   (1 when clip <= 0)
__global__ __launch_bounds__(1024) void ca_tensor_factor_kernel(const float* __restrict__ chunk_ss,
                                                                const int64_t* __restrict__ offsets, float clip,
                                                                float* __restrict__ factor) {
    const int t = blockIdx.x;
    const int64_t c0 = offsets[t] / CA_CHUNK, c1 = (offsets[t + 1] + CA_CHUNK - 1) / CA_CHUNK;
    // 1024 threads, eight independent loads per thread and round: the 33.8 k chunk sums of hidden1_weights are five rounds (one
    // load in flight per thread of a 256-thread workgroup: 132 dependent rounds, 42 us of pure latency; four in flight: 36 us).
    // The order of the additions is fixed.
    double s = 0.0;
    for (int64_t c = c0 + threadIdx.x; c < c1; c += 1024 * 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (c + 1024 * u < c1) ? chunk_ss[c + 1024 * u] : 0.f;
        s += (((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) + (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double sh[16];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 16; ++i) tot += sh[i];
        const float nrm = (float)sqrt(tot);
        factor[t] = clip > 0.f ? clip / fmaxf(nrm, clip) : 1.f;
    }
}

__global__ __launch_bounds__(256) void ca_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const int64_t* __restrict__ offsets, int ntensors,
                                                       int64_t total, const float* __restrict__ factor, float lr_t,
                                                       float b1, float b2, float eps, int nt, const float* __restrict__ l2) {
    const int64_t base = (int64_t)blockIdx.x * CA_CHUNK;
    const int lo = ca_owner(offsets, ntensors, base);
    const float f = factor[lo];
    const float c2 = l2 ? l2[lo] : 0.f;
#pragma unroll
    for (int i = 0; i < CA_CHUNK / (256 * 4); ++i) {
        const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
        if (e + 3 < total) {
            // every byte of the four arenas is touched once per step and not again before the next step: non-temporal loads and
            // stores keep 2.6 GB (cfg-2) ... 9.5 GB (cfg-5) of dead lines out of the caches (LPM_ADAM_NT=0: plain accesses, A/B)
            float4 pp, gg, mm, vv;
            if (nt) {
                const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + e));
                const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + e));
                const f32x4 c = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + e));
                const f32x4 d = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + e));
                pp = make_float4(a[0], a[1], a[2], a[3]); gg = make_float4(b[0], b[1], b[2], b[3]);
                mm = make_float4(c[0], c[1], c[2], c[3]); vv = make_float4(d[0], d[1], d[2], d[3]);
            } else {
                pp = *reinterpret_cast<float4*>(p + e);
                gg = *reinterpret_cast<const float4*>(g + e);
                mm = *reinterpret_cast<float4*>(m + e);
                vv = *reinterpret_cast<float4*>(v + e);
            }
            if (c2 != 0.f) {                 // + the L2 penalty's gradient (the same fmaf as the norm pass)
                gg.x = fmaf(c2, pp.x, gg.x); gg.y = fmaf(c2, pp.y, gg.y); gg.z = fmaf(c2, pp.z, gg.z); gg.w = fmaf(c2, pp.w, gg.w);
            }
#define LPM_ADAM1(c) adam_element(gg.c * f, pp.c, mm.c, vv.c, lr_t, b1, b2, eps);
            LPM_ADAM1(x) LPM_ADAM1(y) LPM_ADAM1(z) LPM_ADAM1(w)
#undef LPM_ADAM1
            if (nt) {
                __builtin_nontemporal_store(f32x4{pp.x, pp.y, pp.z, pp.w}, reinterpret_cast<f32x4*>(p + e));
                __builtin_nontemporal_store(f32x4{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f32x4*>(m + e));
                __builtin_nontemporal_store(f32x4{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f32x4*>(v + e));
            } else {
                *reinterpret_cast<float4*>(p + e) = pp;
                *reinterpret_cast<float4*>(m + e) = mm;
                *reinterpret_cast<float4*>(v + e) = vv;
            }
        }
    }
}

}  // namespace lpm

extern "C" size_t lpm_clip_adam_scratch_bytes(int64_t total, int ntensors) {
    const int64_t nchunk = (total + lpm::CA_CHUNK - 1) / lpm::CA_CHUNK;
    return (size_t)(nchunk + ntensors) * sizeof(float);
}

static int clip_adam_impl(float* param, const float* grad, float* m, float* v, const int64_t* offsets, const float* l2coef, int ntensors,
                          int64_t total, float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step, float* scratch,
                          lpm_stream_t stream);
extern "C" int lpm_multi_tensor_clip_adam(float* param, const float* grad, float* m, float* v, const int64_t* offsets,
                                          int ntensors, int64_t total, float clip_norm, float lr, float beta1,
                                          float beta2, float eps, int64_t step, float* scratch, lpm_stream_t stream) {
    return clip_adam_impl(param, grad, m, v, offsets, nullptr, ntensors, total, clip_norm, lr, beta1, beta2, eps, step, scratch, stream);
}
// ... with the gradient of each variable's L2 penalty, l2coef[t] * w (device array [ntensors]; 0 = none), added on the fly in the norm
// pass and in the update pass: what `grad += l2coef[t] * param` in front of lpm_multi_tensor_clip_adam computes, without that pass
extern "C" int lpm_multi_tensor_clip_adam_l2(float* param, const float* grad, float* m, float* v, const int64_t* offsets, const float* l2coef,
                                             int ntensors, int64_t total, float clip_norm, float lr, float beta1, float beta2, float eps,
                                             int64_t step, float* scratch, lpm_stream_t stream) {
    return clip_adam_impl(param, grad, m, v, offsets, l2coef, ntensors, total, clip_norm, lr, beta1, beta2, eps, step, scratch, stream);
}
static int clip_adam_impl(float* param, const float* grad, float* m, float* v, const int64_t* offsets, const float* l2coef, int ntensors,
                          int64_t total, float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step, float* scratch,
                          lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(param && grad && m && v && offsets && scratch, LPM_ERR_BADARG, "lpm_multi_tensor_clip_adam: null pointer");
    LPM_REQUIRE(ntensors > 0 && total > 0 && step >= 1, LPM_ERR_BADARG, "lpm_multi_tensor_clip_adam: bad sizes (step is 1-based)");
    LPM_REQUIRE(total % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_multi_tensor_clip_adam: arena length must be a multiple of 4");
    LPM_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, LPM_ERR_BADARG,
                "lpm_multi_tensor_clip_adam: arenas must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nchunk = (total + CA_CHUNK - 1) / CA_CHUNK;
    float* chunk_ss = scratch;
    float* factor = scratch + nchunk;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    static const int adam_nt = [] { const char* e = getenv("LPM_ADAM_NT"); return (e && e[0] == '0') ? 0 : 1; }();
    hipLaunchKernelGGL(ca_chunk_sumsq_kernel, dim3((unsigned)nchunk), dim3(256), 0, s, grad, total, chunk_ss, (const float*)param, offsets, ntensors,
                       l2coef);
    hipLaunchKernelGGL(ca_tensor_factor_kernel, dim3(ntensors), dim3(1024), 0, s, chunk_ss, offsets, clip_norm, factor);
    hipLaunchKernelGGL(ca_apply_kernel, dim3((unsigned)nchunk), dim3(256), 0, s, param, grad, m, v, offsets, ntensors, total,
                       factor, (float)lr_t, beta1, beta2, eps, adam_nt, l2coef);
    return check_launch("lpm_multi_tensor_clip_adam");
}
