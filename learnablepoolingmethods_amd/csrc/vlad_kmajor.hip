// K2, fourth form (round 3; measured, NOT the default -- ops.VLAD_KMAJOR_SCALED): the residual aggregation of
// frame_level_models.py:2803-2822 for a consumer that applies the two normalisations itself (the NetVladV1 cluster encoders,
// App. C5: tokens = clusters) -- ONE launch writes the un-normalised residual sums k-major [B, K, D] straight from the accumulators
// AND the per-(clip, cluster) row scale 1 / (n_k sqrt(g)) that normalises them (lpm_vlad_aggregate_raw_kmajor_fwd +
// lpm_vlad_row_scales of vlad_tiles3.hip are two launches).
//
// Idea: the third form (128 clusters x 128 columns per workgroup) is bound by the L2 -> LDS delivery -- per clip the assignment tiles
// are re-read by D/128 = 8 column slabs and the frame tiles by K/128 = 2 cluster slabs, 4.8 MB of LDS-DMA per clip for 2.6 MB
// algorithmic.  Here, for K = 256, a workgroup owns ALL 256 clusters x 128 columns of a clip ("wide" item): the frame tiles are
// fetched once (3.6 MB per clip, -25 %), and a wave's 64 x 64 accumulator tile reads 8 KB of fragments per 12 MFMAs (-33 % LDS bytes
// per MFMA).  24 KB per 16-frame step in a 3-stage ring = 72 KB of LDS, two workgroups per CU, 127 VGPRs.  80 clips x 8 slabs = 640
// wide items do not divide over 2 x 256 slots, so the grid can be MIXED: whole rounds of clips as wide items (dispatched first), the
// remaining clips as 128 x 128 items in the tail of the same launch.
//
// Measured (tools/time_k2_forms.py, tools/k2_ablate.sh, cfg-2 video shape, one box; the two-launch chain on the same box: 69 + 4.4 us):
//   all clips wide 78.5 us, whole rounds wide 78.0 us, no clip wide 82.8 us;  main loop alone 64 / 59 / 62 us (third form: ~50),
//   epilogue alone 30 us, without the stores -10 us, without the norm / arrival section -10 us.
// The wide loop moves 25 % fewer bytes at the SAME ~7.4 TB/s of L2 -> LDS delivery the third form reaches (8.1), but 640 items on 512
// slots leave the second round a quarter full, and the mixed grid's tail (256 lone 128 x 128 workgroups) cannot pull that rate
// either; two workgroups per CU instead of three cost the 128 x 128 items 12 us.  The per-clip arrival (three barriers, a drained
// vmcnt behind 128 KB of tile stores, one atomic) costs a workgroup more than the 4.4 us launch it replaces.  First attempt of the
// arrival with an acq_rel agent-scope fetch_add: 105-140 us -- at agent scope the release writes back the XCD's whole L2 (tens of MB
// of dirty descriptor lines) once per workgroup; relaxed atomic + write-through stores + L1-bypassing loads (guide 5) is the form
// kept.  Writing the residual back into the accumulators for a separate store pass behind the arrival: 70-136 spilled registers.
// Kept as an A/B form (LPM_VLAD_KMAJOR_SCALED=1) and for its epilogue technique:
//
// Epilogue, one pass in the accumulators' own layout (acc[r]: column d = lane & 31, cluster 8 (r >> 2) + 4 (lane >> 5) + (r & 3)): the
// centres are read as float4 along k, the residual and the store happen in registers -- for a fixed register the 32 lanes of a
// half-wave hold 32 consecutive d of one cluster row, a 128-byte segment of the k-major result; addresses are (uniform SGPR base) +
// (one per-lane offset) -- and the partial column norms (sum over the wave's columns = over its lanes) come from a 16-value / 32-lane
// transpose-reduction per cluster tile (15 shuffles + 1, fixed order) instead of a butterfly per accumulator register.  No LDS
// transposes.
// Row scales: the workgroups of a clip publish their partial norms with write-through stores and count themselves in on a per-clip
// counter (zeroed by the launcher's memset node; relaxed agent-scope fetch_add behind a drained vmcnt); whoever arrives last adds
// the clip's partial norms in slab order (L1-bypassing loads) and writes scale, colsq, csq, gsq -- nobody waits, and who is last
// changes no bit of the result.  Which form a clip takes changes the summation order of its assignment sums in the last bit.
#include "lpm_common.h"
#include <type_traits>

namespace lpm {

typedef __bf16 vk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned vk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 vk_mfma(vk_u32x4 a, vk_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vk_bf16x8, a), __builtin_bit_cast(vk_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float vk_bf(unsigned h) { return __uint_as_float(h << 16); }

constexpr int VK_NS = 3;                          // ring stages
constexpr int VK_RING = VK_NS * 24 * 1024;        // the wide item's ring; the 128 x 128 item uses 48 KB of it

struct VKArgs {
    const uint4* at;            // assignment tiles [b][K/32][S][plane][lane]   (lpm_assign_tiles)
    const uint4* xt;            // frame tiles      [b][S][D/32][plane][lane]   (lpm_split_frames / lpm_frame_apply_tiles)
    const float* centres;       // [D, K] (cluster_weights2) or null
    int T, D, K, S, residual;
    int B_large, n_large;       // clips [0, B_large) run as wide items; n_large = B_large * D/128 workgroups
    float* out;                 // [B, K, D] un-normalised residual sums
    float* asum;                // [B, K]
    float* colsq_part;          // [B, D/128, K]
    float* scale;               // [B, K] 1 / (n_k sqrt(g))
    float* colsq;               // [B, K]
    float* csq;                 // [B, K]
    float* gsq;                 // [B]
    unsigned* arrive;           // [B], zero at launch
    int dbg;                    // measurement only (LPM_VK_DBG): 1 no main loop, 2 no stores, 8 no norms / arrival
};

// one item: clip b, clusters [kb * 128 CW, + 128 CW), columns [ds * 128, + 128).  CW = 2: wide (needs K == 256, kb == 0).
// waves: kw = 64-cluster group (2 CW of them), dw = (32 CW)-column group (4 / CW of them); accumulators [2 cluster tiles][CW column tiles]
template <int CW>
__device__ __forceinline__ void vk_item(const VKArgs& g, const int b, const int kb, const int ds, unsigned char* smem) {
    constexpr int NA = 4 * CW;                    // assignment tiles per stage
    constexpr int NPIECE = 2 * NA + 8;            // 1 KB pieces per stage (hi and lo planes of NA + 4 tiles)
    constexpr int PW = NPIECE / 8;                // pieces per wave
    constexpr int STAGE = NPIECE * 1024;
    constexpr int XOFF = 2 * NA * 1024;
    constexpr int NDW = 4 / CW;                   // column groups
    constexpr int NPART = NDW / 2;                // partial assignment sums per cluster (waves sharing a cluster tile's sum)
    constexpr int NCL = 128 * CW;                 // clusters per item
    static_assert(VK_NS * STAGE <= VK_RING, "ring");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int kw = wave / NDW, dw = wave % NDW;
    const int D = g.D, K = g.K, S = g.S;
    const int DT = D >> 5, KT = K >> 5, P = D >> 7;

    const uint4* src[PW];
    int64_t sstep[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int p = wave + 8 * j;
        if (p < 2 * NA) {
            src[j] = g.at + ((((int64_t)b * KT + kb * NA + (p >> 1)) * S) * 2 + (p & 1)) * 64 + lane;
            sstep[j] = 128;
        } else {
            const int q = p - 2 * NA;
            src[j] = g.xt + ((((int64_t)b * S) * DT + ds * 4 + (q >> 1)) * 2 + (q & 1)) * 64 + lane;
            sstep[j] = (int64_t)DT * 128;
        }
    }
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % VK_NS) * STAGE;
#pragma unroll
        for (int j = 0; j < PW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (int64_t)s * sstep[j]),
                                             (__attribute__((address_space(3))) void*)(st + (wave + 8 * j) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][CW];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < CW; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][e][r] = 0.f;
    float asum_w = 0.f;      // partial assignment sum of cluster row l31 of tile (kw, dw / NPART), this lane's 8 frames of every step

#pragma unroll
    for (int s = 0; s < VK_NS - 1; ++s)
        if (s < S && !(g.dbg & 1)) issue(s);
    for (int s = 0; s < ((g.dbg & 1) ? 0 : S); ++s) {
        if (s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");      // one younger step stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // everyone's pieces of step s are in LDS; stage (s - 1) % NS is free
        asm volatile("" ::: "memory");
        if (s + VK_NS - 1 < S) issue(s + VK_NS - 1);
        const unsigned char* st = smem + (s % VK_NS) * STAGE;
        const vk_u32x4* af = reinterpret_cast<const vk_u32x4*>(st) + lane;
        const vk_u32x4* xf = reinterpret_cast<const vk_u32x4*>(st + XOFF) + lane;
        vk_u32x4 xh[CW], xl[CW], ah[2], al[2];
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            xh[e] = xf[((dw * CW + e) * 2 + 0) * 64];
            xl[e] = xf[((dw * CW + e) * 2 + 1) * 64];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ah[c] = af[((kw * 2 + c) * 2 + 0) * 64];
            al[c] = af[((kw * 2 + c) * 2 + 1) * 64];
        }
        // term-major: 2 CW independent accumulators between dependent MFMAs; every accumulator still sees ah.xh, ah.xl, al.xh in order
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < CW; ++e) acc[c][e] = vk_mfma(ah[c], xh[e], acc[c][e]);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < CW; ++e) acc[c][e] = vk_mfma(ah[c], xl[e], acc[c][e]);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < CW; ++e) acc[c][e] = vk_mfma(al[c], xh[e], acc[c][e]);
        // assignment sums: the NDW waves of a 64-cluster group share its 2 tiles x 4 fragment words (2 frames each)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c == dw / NPART) {
#pragma unroll
                for (int w = 0; w < 4 / NPART; ++w) {
                    const int q2 = (dw % NPART) * (4 / NPART) + w;
                    const unsigned h = ah[c][q2], l = al[c][q2];
                    asum_w += (vk_bf(h & 0xffffu) + vk_bf(l & 0xffffu)) + (vk_bf(h >> 16) + vk_bf(l >> 16));
                }
            }
        }
    }
    __syncthreads();         // no DMA in flight (the last step waited for vmcnt(0)), all fragment reads done: the ring is scratch now

    float* red_a = reinterpret_cast<float*>(smem);          // [NPART][NCL] partial assignment sums
    float* ssum = red_a + 2 * 256;                          // [NCL]
    float* red_n = ssum + 256;                              // [NDW][NCL] partial square norms
    float* invn = red_n + 4 * 256;                          // [K <= 1024] (last arriver)
    float* wg = invn + 1024;                                // [4]
    int* flag = reinterpret_cast<int*>(wg + 8);
    const int k0 = kb * NCL + kw * 64;                      // first cluster of this wave
    // addresses as (wave-uniform base) + (one per-lane byte offset): SGPR bases, a single VGPR -- 64 independent 64-bit per-lane
    // addresses would not fit next to 64 accumulators
    constexpr int NTILE = 2 * CW;
    const unsigned cen_voff = (unsigned)(l31 * K + 4 * half) * 4u;
    // the centres of accumulator tile t = (c, e) = (t / CW, t % CW): 4 float4 along k per lane (its column d, clusters 8 q + 4 half + 0..3)
    auto load_centres = [&](int t, float4 (&dst)[4]) {
        const int c = t / CW, e = t % CW;
        const float* cb = g.centres + ((int64_t)(ds * 128 + (dw * CW + e) * 32) * K + k0 + c * 32);       // uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cb + 8 * q) + cen_voff);
    };
    float4 cen[4];       // (one tile's worth: a second buffer would not fit 128 registers next to 64 accumulators)
    if (g.residual) load_centres(0, cen);       // requested before the assignment sums meet in LDS: that L2 round trip is not exposed
    asum_w += __shfl_xor(asum_w, 32, 64);
    if (half == 0) red_a[(dw % NPART) * NCL + (kw * 2 + dw / NPART) * 32 + l31] = asum_w;
    __syncthreads();
    if (tid < NCL) ssum[tid] = NPART == 2 ? red_a[tid] + red_a[NCL + tid] : red_a[tid];
    __syncthreads();

    // One pass over the accumulators: residual, squares and the k-major store of every tile (for a fixed register the 32 lanes of a
    // half-wave write 128 contiguous bytes of one cluster row; (uniform base per register) + (one per-lane offset)).  Per cluster
    // tile c: nv[r] = sum over this wave's columns (lanes x CW tiles) of u^2, then 16 values x 32 lanes -> lane i of each half-wave
    // holds the total of value i >> 1: at the step with lane bit m a lane keeps the half of its values whose index bit equals its
    // lane bit and hands the other half to its partner (15 shuffles, fixed order), the last step adds the two lanes that hold the
    // same value.  (Writing u back into the accumulators for a later store pass costs ~70-130 spilled registers with hipcc 7.2.)
    float ntot[2];
    const unsigned out_voff = (unsigned)(4 * half * D + l31) * 4u;
    const bool do_store = !(g.dbg & 2);
    auto pass1 = [&](auto res_tag) {
        constexpr bool RES = decltype(res_tag)::value;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float nv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) nv[i] = 0.f;
#pragma unroll
            for (int e = 0; e < CW; ++e) {
                const int t = c * CW + e;
                float* ob = g.out + (((int64_t)b * K + k0 + c * 32) * D + ds * 128 + (dw * CW + e) * 32);            // uniform
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float sv[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (RES) {
                        const float4 s4 = *reinterpret_cast<const float4*>(ssum + kw * 64 + c * 32 + 8 * q + 4 * half);
                        sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
                        cv[0] = cen[q].x; cv[1] = cen[q].y; cv[2] = cen[q].z; cv[3] = cen[q].w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float u = acc[c][e][4 * q + j];
                        if (RES) u -= sv[j] * cv[j];
                        nv[4 * q + j] = fmaf(u, u, nv[4 * q + j]);
                        if (do_store) *reinterpret_cast<float*>(reinterpret_cast<char*>(ob + (int64_t)(8 * q + j) * D) + out_voff) = u;
                    }
                }
                if (RES && t + 1 < NTILE) load_centres(t + 1, cen);     // in flight under the reduction below / the other waves' work
            }
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int m = 16 >> step, n2 = 8 >> step;
                const bool up = (l31 & m) != 0;
#pragma unroll
                for (int i = 0; i < n2; ++i) {
                    const float keep = up ? nv[n2 + i] : nv[i];
                    const float send = up ? nv[i] : nv[n2 + i];
                    nv[i] = keep + __shfl_xor(send, m, 64);
                }
            }
            ntot[c] = nv[0] + __shfl_xor(nv[0], 1, 64);
        }
    };
    if (g.residual) pass1(std::true_type{});
    else pass1(std::false_type{});
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) unsigned gu32;
    if (!(g.dbg & 8)) {
        if ((l31 & 1) == 0) {                           // lanes 2 r, 2 r + 1 hold value r of both cluster tiles
            const int r = l31 >> 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) red_n[dw * NCL + kw * 64 + c * 32 + 8 * (r >> 2) + 4 * half + (r & 3)] = ntot[c];
        }
        __syncthreads();
        if (tid < NCL) {
            const float pn = NDW == 4 ? (red_n[tid] + red_n[NCL + tid]) + (red_n[2 * NCL + tid] + red_n[3 * NCL + tid]) : red_n[tid] + red_n[NCL + tid];
            __hip_atomic_store((gfloat*)(g.colsq_part + ((int64_t)b * P + ds) * K + kb * NCL + tid), pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ds == 0) g.asum[(int64_t)b * K + kb * NCL + tid] = ssum[tid];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) {
            const unsigned need = (unsigned)(P * (K / NCL));
            // relaxed: the partial norms went out as write-through (sc1) stores that this workgroup has waited for, and the last
            // arriver reads them back with L1-bypassing (sc1) loads -- no release / acquire fence, which at agent scope would write back
            // and invalidate the XCD's whole L2 (tens of MB of dirty descriptor lines) once per workgroup (guide 5, sampling-GEMM item 2)
            *flag = __hip_atomic_fetch_add((gu32*)(g.arrive + b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == need - 1u;
        }
        __syncthreads();
        if (*flag) {
            // ---- the clip is complete and this workgroup arrived last: n_k, 1 / n_k, c_k, g and the row scales (vlad_row_scales_kernel's
            // arithmetic and order of operations: 256 threads, k = tid, tid + 256, ...)
            float gs = 0.f;
            if (tid < 256) {
                for (int k = tid; k < K; k += 256) {
                    float n = 0.f;
                    for (int p = 0; p < P; ++p)
                        n += __hip_atomic_load((gfloat*)(g.colsq_part + ((int64_t)b * P + p) * K + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float iv = rsqrtf(fmaxf(n, kL2Eps));
                    const float cc = n * iv * iv;
                    invn[k] = iv;
                    gs += cc;
                    g.colsq[(int64_t)b * K + k] = n;
                    g.csq[(int64_t)b * K + k] = cc;
                }
                gs = wave_sum(gs);
                if (lane == 0) wg[wave] = gs;
            }
            __syncthreads();
            const float tot = (wg[0] + wg[1]) + (wg[2] + wg[3]);
            const float ig = rsqrtf(fmaxf(tot, kL2Eps));
            if (tid == 0) g.gsq[b] = tot;
            if (tid < 256)
                for (int k = tid; k < K; k += 256) g.scale[(int64_t)b * K + k] = invn[k] * ig;
        }
    }
}

__global__ __launch_bounds__(512, 4) void vlad_kmajor_kernel(const VKArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // the ONLY LDS object (guide 5, trap (a))
    const int bid = blockIdx.x;
    const int P = g.D >> 7;
    if (bid < g.n_large) {                         // wave-uniform, workgroup-uniform
        const int lid = xcd_remap(bid, g.n_large);             // the P slabs of a clip: consecutive ids, one XCD, the same moment
        vk_item<2>(g, lid / P, 0, lid % P, smem);
    } else {
        const int KB = g.K >> 7;
        const int lid = xcd_remap(bid - g.n_large, (int)gridDim.x - g.n_large);
        const int rem = lid % (KB * P);
        vk_item<1>(g, g.B_large + lid / (KB * P), rem / P, rem % P, smem);
    }
}

// clips that run as wide items: whole rounds of 2 x 256 workgroup slots (64 clips at D = 1024), the rest as 128 x 128 items in the tail
static int vk_large_clips(int B, int D, int K, int flags) {
    static const int env_mode = [] { const char* e = getenv("LPM_K2_WIDE"); return e ? atoi(e) : 1; }();      // 0: none, 1: rounds, 2: all
    const int mode = (flags & LPM_VLAD_WIDE_ALL) ? 2 : ((flags & LPM_VLAD_WIDE_NONE) ? 0 : env_mode);
    if (K != 256 || mode == 0) return 0;
    if (mode == 2) return B;
    const int P = D / 128;
    const int per_round = 512 / P > 0 ? 512 / P : 1;
    return (B / per_round) * per_round;
}

}  // namespace lpm

extern "C" size_t lpm_vlad_kmajor_workspace_bytes(int B, int D, int K) {
    return ((size_t)B * (D / 128) * K + (size_t)B) * sizeof(float);        // colsq_part [B, D/128, K] | arrive [B]
}

// at / xt: lpm_assign_tiles / frame tiles; centres [D, K] when LPM_VLAD_RESIDUAL is set.  Outputs: raw_kmajor [B, K, D] (un-normalised
// residual sums), scale [B, K] (descriptor[b, k, :] = raw[b, k, :] * scale[b, k]), asum / colsq / csq [B, K], gsq [B] (for the backward).
extern "C" int lpm_vlad_aggregate_kmajor_scaled_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                                    int flags, float* raw_kmajor, float* scale, float* asum, float* colsq, float* csq,
                                                    float* gsq, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && raw_kmajor && scale && asum && colsq && csq && gsq && workspace, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_kmajor_scaled_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_kmajor_scaled_fwd: RESIDUAL needs centres");
    LPM_REQUIRE(B > 0 && T > 0 && D > 0 && K > 0 && D % 128 == 0 && K % 128 == 0 && K <= 1024, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_kmajor_scaled_fwd: need D %% 128 == 0, K %% 128 == 0, K <= 1024 (D=%d K=%d)", D, K);
    LPM_REQUIRE(workspace_bytes >= lpm_vlad_kmajor_workspace_bytes(B, D, K), LPM_ERR_WORKSPACE,
                "lpm_vlad_aggregate_kmajor_scaled_fwd: workspace too small");
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)raw_kmajor | (uintptr_t)workspace) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_kmajor_scaled_fwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int P = D / 128;
    VKArgs g{};
    g.at = (const uint4*)at; g.xt = (const uint4*)xt; g.centres = centres;
    g.T = T; g.D = D; g.K = K; g.S = (T + 15) / 16; g.residual = residual;
    g.B_large = vk_large_clips(B, D, K, flags); g.n_large = g.B_large * P;
    g.out = raw_kmajor; g.asum = asum; g.colsq_part = (float*)workspace;
    g.scale = scale; g.colsq = colsq; g.csq = csq; g.gsq = gsq;
    g.arrive = (unsigned*)(g.colsq_part + (size_t)B * P * K);
    static const int dbg = [] { const char* e = getenv("LPM_VK_DBG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    if (hipMemsetAsync(g.arrive, 0, (size_t)B * sizeof(unsigned), s) != hipSuccess) {        // the arrival counters: zero at every launch
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_kmajor_scaled_fwd: cannot clear the arrival counters");
        return LPM_ERR_LAUNCH;
    }
    const size_t lds = VK_RING;
    auto kern = vlad_kmajor_kernel;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_kmajor_scaled_fwd: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid((unsigned)(g.n_large + (B - g.B_large) * (K / 128) * P));
    hipEvent_t e0, e1;
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, s, e0, e1, 0, g);
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, g);
    return check_launch("lpm_vlad_aggregate_kmajor_scaled_fwd");
}
