// MoeModel tail + CrossEntropyLoss fused (video_level_models.py:116-126, losses.py:41-51): per (clip, class)
//   g = softmax over the m+1 gate activations, e = sigmoid of the m expert activations, p = sum_{i<m} g_i e_i,
//   loss = mean_b sum_c -[ y log(p + eps) + (1 - y) log(1 - p + eps) ],  eps = 1e-5
// and its backward in closed form:
//   dL/dp = dloss * (-(y/(p+eps) - (1-y)/(1-p+eps)) / B) + dpred
//   d/d expert_act_i = dL/dp * g_i e_i (1 - e_i),    d/d gate_act_j = dL/dp * g_j (e_j [j<m] - p)
// 1 - p and 1 - e_i are formed WITHOUT cancellation: 1 - e_i = sigmoid(-expert_act_i) and, because the m+1 gates sum to one,
// q = 1 - p = g_m + sum_{i<m} g_i (1 - e_i).  A freshly initialised reference model saturates (hidden1_weights ~ N(0, 1/K) on a
// layer-normed descriptor gives |activation| ~ 30-70 and predictions within 1e-9 of 0 or 1): with q from the rounded p, fp32
// holds log(1 - p + eps) and 1 / (1 - p + eps) to 6e-3 only (eps = 1e-5 against a 6e-8 spacing of p next to 1); with q summed
// from its small terms they are good to fp32 rounding, like the fp64 evaluation of the same formula.
// ~35 elementwise / reduction launches of the host graph become three.  The two FC layers around it stay library GEMMs.
#include "lpm_common.h"

namespace lpm {

constexpr int MOE_MAX_MIX = 8;

// -> g (gates), e (experts), ne = 1 - e, p = sum g e, q = 1 - p
__device__ __forceinline__ void moe_mix(const float* __restrict__ ga, const float* __restrict__ ea, int m, float* g, float* e, float* ne,
                                        float& p, float& q) {
    float mx = ga[0];
    for (int i = 1; i <= m; ++i) mx = fmaxf(mx, ga[i]);
    float s = 0.f;
    for (int i = 0; i <= m; ++i) {
        g[i] = __expf(ga[i] - mx);
        s += g[i];
    }
    const float inv = 1.f / s;
    p = 0.f;
    for (int i = 0; i <= m; ++i) g[i] *= inv;
    q = g[m];
    for (int i = 0; i < m; ++i) {
        // sigmoid(a) and sigmoid(-a) from one exponential of -|a| (never overflows)
        const float a = ea[i], t = __expf(-fabsf(a)), r = 1.f / (1.f + t);
        const float big = r, small = t * r;
        e[i] = a >= 0.f ? big : small;
        ne[i] = a >= 0.f ? small : big;
        p = fmaf(g[i], e[i], p);
        q = fmaf(g[i], ne[i], q);
    }
}

__global__ __launch_bounds__(256) void moe_ce_fwd_kernel(const float* __restrict__ gate_act, const float* __restrict__ expert_act,
                                                         const float* __restrict__ labels, int64_t n, int m, float eps,
                                                         float* __restrict__ pred, float* __restrict__ loss_partial) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (i < n) {
        float g[MOE_MAX_MIX + 1], e[MOE_MAX_MIX], ne[MOE_MAX_MIX], p, q;
        moe_mix(gate_act + i * (m + 1), expert_act + i * m, m, g, e, ne, p, q);
        pred[i] = p;
        if (labels) {
            const float y = labels[i];
            l = -(y * __logf(p + eps) + (1.f - y) * __logf(q + eps));
        }
    }
    if (loss_partial) {
        l = wave_sum(l);
        __shared__ float w[4];
        if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = l;
        __syncthreads();
        if (threadIdx.x == 0) loss_partial[blockIdx.x] = (w[0] + w[1]) + (w[2] + w[3]);
    }
}

__global__ __launch_bounds__(256) void moe_ce_loss_reduce_kernel(const float* __restrict__ partial, int nblk, float inv_batch,
                                                                 float* __restrict__ loss) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] * (double)inv_batch);
}

__global__ __launch_bounds__(256) void moe_ce_bwd_kernel(const float* __restrict__ gate_act, const float* __restrict__ expert_act,
                                                         const float* __restrict__ labels, const float* __restrict__ dloss,
                                                         const float* __restrict__ dpred, int64_t n, int m, float eps,
                                                         float inv_batch, float* __restrict__ dgate, float* __restrict__ dexpert) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g[MOE_MAX_MIX + 1], e[MOE_MAX_MIX], ne[MOE_MAX_MIX], p, q;
    moe_mix(gate_act + i * (m + 1), expert_act + i * m, m, g, e, ne, p, q);
    float dp = dpred ? dpred[i] : 0.f;
    if (labels && dloss) {
        const float y = labels[i];
        dp += dloss[0] * inv_batch * -(y / (p + eps) - (1.f - y) / (q + eps));
    }
    for (int j = 0; j < m; ++j) {
        dexpert[i * m + j] = dp * g[j] * e[j] * ne[j];
        dgate[i * (m + 1) + j] = dp * g[j] * (q - ne[j]);          // e_j - p = (1 - p) - (1 - e_j)
    }
    dgate[i * (m + 1) + m] = dp * g[m] * (0.f - p);
}

}  // namespace lpm

extern "C" int lpm_moe_ce_nblk(int B, int V) { return (int)(((int64_t)B * V + 255) / 256); }

extern "C" int lpm_moe_ce_fwd(const float* gate_act, const float* expert_act, const float* labels, int B, int V, int num_mixtures,
                              float eps, float* predictions, float* loss, float* loss_partial, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(gate_act && expert_act && predictions, LPM_ERR_BADARG, "lpm_moe_ce_fwd: null pointer");
    LPM_REQUIRE((labels == nullptr) == (loss == nullptr) && (loss == nullptr) == (loss_partial == nullptr), LPM_ERR_BADARG,
                "lpm_moe_ce_fwd: labels, loss and loss_partial go together");
    LPM_REQUIRE(B > 0 && V > 0 && num_mixtures >= 1 && num_mixtures <= MOE_MAX_MIX, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_moe_ce_fwd: need 1 <= num_mixtures <= %d (got %d)", MOE_MAX_MIX, num_mixtures);
    const int64_t n = (int64_t)B * V;
    const int nblk = lpm_moe_ce_nblk(B, V);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_ce_fwd_kernel, dim3(nblk), dim3(256), 0, s, gate_act, expert_act, labels, n, num_mixtures, eps, predictions,
                       loss_partial);
    if (loss) hipLaunchKernelGGL(moe_ce_loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_partial, nblk, 1.f / (float)B, loss);
    return check_launch("lpm_moe_ce_fwd");
}

extern "C" int lpm_moe_ce_bwd(const float* gate_act, const float* expert_act, const float* labels, const float* dloss,
                              const float* dpredictions, int B, int V, int num_mixtures, float eps, float* dgate_act,
                              float* dexpert_act, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(gate_act && expert_act && dgate_act && dexpert_act, LPM_ERR_BADARG, "lpm_moe_ce_bwd: null pointer");
    LPM_REQUIRE((labels == nullptr) == (dloss == nullptr), LPM_ERR_BADARG, "lpm_moe_ce_bwd: labels and dloss go together");
    LPM_REQUIRE(B > 0 && V > 0 && num_mixtures >= 1 && num_mixtures <= MOE_MAX_MIX, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_moe_ce_bwd: need 1 <= num_mixtures <= %d (got %d)", MOE_MAX_MIX, num_mixtures);
    const int64_t n = (int64_t)B * V;
    hipLaunchKernelGGL(moe_ce_bwd_kernel, dim3(lpm_moe_ce_nblk(B, V)), dim3(256), 0, (hipStream_t)stream, gate_act, expert_act, labels, dloss,
                       dpredictions, n, num_mixtures, eps, 1.f / (float)B, dgate_act, dexpert_act);
    return check_launch("lpm_moe_ce_bwd");
}
