// K2 for bf16 storage on clip-wide items (round 6) -- the residual aggregation of frame_level_models.py:2803-2817 at
// BASELINE configs[4]'s shape (K = 512 clusters, D = 1024, 300 frames, plain bf16 tiles: ONE MFMA per product) with the form that fixed
// cfg-2 in round 4 (vlad_clip.hip) instead of 128 x 128 items (vlad_tiles3.hip<false, 1>: 98 us = 0.32 of the HBM peak, 1.495 x the
// algorithmic traffic -- every clip's assignment tiles re-read by D / 128 = 8 column slabs, its frame tiles by K / 128 = 4 cluster slabs).
//
// A 512-thread workgroup owns HALF the clusters (256 = 8 cluster tiles) x a THIRD of a clip's columns (11 / 11 / 10 column tiles of 32 at
// D = 1024): 6 workgroups per clip, consecutive ids (one XCD, the same moment), 768 at cfg-5 = three rounds of the chip.  Per clip the frame
// tiles come in twice (1.2 MB) and the assignment tiles three times (0.9 MB) against 2.5 + 2.5 MB before.
// Plain bf16 makes the FRAGMENT READS the scarce resource: one 1 KB ds_read_b128 feeds one 32-cycle MFMA where the split form feeds
// three, and the LDS delivers 128 bytes per cycle to the whole CU.  So a wave owns a 2 x 6 register tile -- cluster tiles (2 cp, 2 cp + 1)
// x six column tiles of the slab's first or second half: 8 fragment reads for 12 MFMAs (192 accumulator registers); one cluster tile x
// 11 column tiles, vlad_clip.hip's shape, would read 12 fragments for 11 MFMAs -- the LDS as busy as the matrix pipe.
// Operand roles as in vlad_tiles3.hip: assignment tile = MFMA A operand (rows = clusters), frame tile = B operand (columns = d):
// acc[m][j][r] = sum over frames for column d = l31 of the tile, cluster 32 (2 cp + m) + 8 (r >> 2) + 4 half + (r & 3): four consecutive
// clusters in consecutive registers, so the d-major result [B, D, K] (the reference's own layout, frame_level_models.py:2817-2821) goes
// through a wave-private fp32 LDS tile [32 d][64 clusters] written with ds_write_b128 and leaves as 16-byte bf16 stores -- 8 lanes =
// one 128-byte piece of a row; residual (asum x centres, fp32) and the slab's square norms are formed on the way, in fp32.
// 24 KB per 16-frame step (8 assignment + up to 12 frame pieces + 4 idle slots so that every wave issues three pieces and the in-order
// vmcnt is one constant), 5-stage ring = 120 KB of LDS, LDS-DMA through buffer loads (resource + step offset in SGPRs).
#include "lpm_common.h"

namespace lpm {

typedef __bf16 vb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned vb_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 vb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float vb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 vb_mfma(vb_u32x4 a, vb_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vb_bf16x8, a), __builtin_bit_cast(vb_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float vb_bf(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned vb_pack2(float a, float b) {          // round to nearest even, both (v_cvt_pk_bf16_f32)
    const vb_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, vb_bf16x2));
}

constexpr int VB_NCT = 12;                         // column tiles per workgroup at most (two groups of six)
constexpr int VB_NJ = 6;                           // column tiles per wave
constexpr int VB_SLOTS = 24;                       // 1 KB slots per stage: 8 assignment pieces, VB_NCT frame pieces, 4 idle
constexpr int VB_PW = VB_SLOTS / 8;                // pieces per wave and step
constexpr int VB_STAGE = VB_SLOTS * 1024;
constexpr int VB_WS = 68;                          // epilogue tile row stride in floats (272 B: 16-byte aligned rows, odd multiple of 16 B)
constexpr int VB_EPI = 8 * 32 * VB_WS * 4;         // 8 wave-private tiles [32 d][64 clusters] fp32: overlays the ring
constexpr int VB_TAIL = (256 + 512) * 4;           // assignment sums [256] + square-norm partials [2][256], behind the tiles

struct VBArgs {
    const uint4* at;            // assignment tiles [b][K/32][S][lane]   (lpm_assign_tiles_bf16: plain bf16, 1 KB per (tile, step))
    const uint4* xt;            // frame tiles      [b][S][D/32][lane]   (lpm_frame_apply_tiles_bf16 / lpm_split_frames_bf16)
    const float* centres;       // [D, K] (cluster_weights2) or null
    int D, K, S, P, KH, residual;
    unsigned short* out;        // [B, D, K] bf16 un-normalised residual sums, d-major
    float* asum;                // [B, K]
    float* colsq_part;          // [B, P, K]
};

template <int NS>
__global__ __launch_bounds__(512) void vlad_clip16_kernel(const VBArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // the ONLY LDS object (guide 5, trap (a))
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int D = g.D, K = g.K, S = g.S, P = g.P, KH = g.KH;
    const int DT = D >> 5, KT = K >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);        // the KH * P workgroups of a clip: consecutive ids, one XCD, the same moment
    const int b = lid / (KH * P), rem_ = lid % (KH * P);
    const int kh = rem_ / P, p = rem_ % P;
    const int base = DT / P, rem = DT % P;
    const int ncol = base + (p < rem ? 1 : 0);               // column tiles of this slab (<= VB_NCT)
    const int ct0 = p * base + min(p, rem);
    const int cp = wave & 3, cg = wave >> 2;                  // cluster pair, column group
    const int n0 = (ncol + 1) >> 1;                           // group 0: tiles [0, n0), group 1: [n0, ncol)
    const int cnt = cg == 0 ? n0 : ncol - n0;                 // this wave's column tiles (<= VB_NJ)
    const int jt0 = cg * n0;

    // this wave's three pieces of a stage: slot = wave + 8 j.  slots 0..7: assignment tile kh * 8 + slot; slots 8..19: frame tile
    // ct0 + min(slot - 8, ncol - 1); slots 20..23: idle (they re-read assignment piece 0: every wave has the same number in flight)
    __amdgpu_buffer_rsrc_t rsrc[VB_PW];
    unsigned sstep[VB_PW];                         // bytes per step
#pragma unroll
    for (int j = 0; j < VB_PW; ++j) {
        const int sl = wave + 8 * j;
        const uint4* src;
        if (sl < 8 || sl >= 8 + VB_NCT) {
            const int q = sl < 8 ? sl : 0;
            src = g.at + (((int64_t)b * KT + kh * 8 + q) * S) * 64;
            sstep[j] = 1024u;
        } else {
            src = g.xt + (((int64_t)b * S) * DT + ct0 + min(sl - 8, ncol - 1)) * 64;
            sstep[j] = (unsigned)DT * 1024u;
        }
        rsrc[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, 0xffffffff, 0x00020000);
    }
    const unsigned lane_off = (unsigned)lane * 16u;
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % NS) * VB_STAGE;
#pragma unroll
        for (int j = 0; j < VB_PW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc[j], (__attribute__((address_space(3))) void*)(st + (wave + 8 * j) * 1024), 16, lane_off,
                                                     (unsigned)s * sstep[j], 0, 2);      // (non-temporal: streamed once per workgroup)
    };

    f32x16 acc[2][VB_NJ];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < VB_NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
    float asum_l[2] = {0.f, 0.f};      // assignment sums of clusters 32 (2 cp + m) + l31 over this lane's 8 frames of every step (group-0 waves)

    // byte offsets of this wave's fragments inside a stage (wave-uniform)
    const int aoff = (2 * cp) * 1024;
    int boff[VB_NJ];
#pragma unroll
    for (int j = 0; j < VB_NJ; ++j) boff[j] = (8 + jt0 + min(j, cnt - 1)) * 1024;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < S) issue(s);
    for (int s = 0; s < S; ++s) {
        // this wave's pieces of step s have landed when at most VB_PW * (younger steps in flight) remain
        const int behind = min(NS - 2, S - 1 - s);
        if (behind >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * VB_PW) : "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VB_PW) : "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VB_PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // everyone's pieces of step s are in LDS; stage (s - 1) % NS is free
        asm volatile("" ::: "memory");
        if (s + NS - 1 < S) issue(s + NS - 1);
        const vb_u32x4* f = reinterpret_cast<const vb_u32x4*>(smem + (s % NS) * VB_STAGE) + lane;
        const vb_u32x4 a0 = f[aoff / 16], a1 = f[aoff / 16 + 64];
#pragma unroll
        for (int j = 0; j < VB_NJ; ++j) {
            const vb_u32x4 bj = f[boff[j] / 16];
            acc[0][j] = vb_mfma(a0, bj, acc[0][j]);
            acc[1][j] = vb_mfma(a1, bj, acc[1][j]);
        }
        if (cg == 0) {                             // wave-uniform: the cluster pair's first column group keeps the assignment sums
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                asum_l[0] += vb_bf(a0[w2] & 0xffffu) + vb_bf(a0[w2] >> 16);
                asum_l[1] += vb_bf(a1[w2] & 0xffffu) + vb_bf(a1[w2] >> 16);
            }
        }
    }
    __syncthreads();         // no DMA in flight (the last step waited for vmcnt(0)), all fragment reads done: the ring is scratch now

    // ---- epilogue
    float* sas = reinterpret_cast<float*>(smem + VB_EPI);     // [256] assignment sums of this cluster half
    float* red = sas + 256;                                   // [2][256] square-norm partials of the two column groups
    if (cg == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float a = asum_l[m] + __shfl_xor(asum_l[m], 32, 64);
            if (half == 0) sas[(2 * cp + m) * 32 + l31] = a;
        }
    }
    __syncthreads();
    const int kbase = kh * 256;
    if (p == 0 && tid < 256) g.asum[(int64_t)b * K + kbase + tid] = sas[tid];
    float* wl = reinterpret_cast<float*>(smem) + wave * (32 * VB_WS);
    const int srow = lane >> 3, k8 = (lane & 7) * 8;          // store pass: d row it * 8 + srow, clusters k8 .. k8 + 7 of the wave's 64
    const bool residual = g.residual != 0;
    float s8[8], nsq[8];
    {
        const float4 sa = *reinterpret_cast<const float4*>(sas + cp * 64 + k8), sb = *reinterpret_cast<const float4*>(sas + cp * 64 + k8 + 4);
        s8[0] = sa.x; s8[1] = sa.y; s8[2] = sa.z; s8[3] = sa.w; s8[4] = sb.x; s8[5] = sb.y; s8[6] = sb.z; s8[7] = sb.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) nsq[e] = 0.f;
    const int kcol = kbase + cp * 64 + k8;                    // this lane's first cluster in the store pass
#pragma unroll
    for (int j = 0; j < VB_NJ; ++j) {
        if (j < cnt) {                                        // wave-uniform
            const int d0 = (ct0 + jt0 + j) * 32;
            float4 c0[4], c1[4];
            if (residual) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float* cr = g.centres + (int64_t)(d0 + it * 8 + srow) * K + kcol;
                    c0[it] = *reinterpret_cast<const float4*>(cr);
                    c1[it] = *reinterpret_cast<const float4*>(cr + 4);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(wl + l31 * VB_WS + m * 32 + 8 * q + 4 * half) =
                        make_float4(acc[m][j][4 * q], acc[m][j][4 * q + 1], acc[m][j][4 * q + 2], acc[m][j][4 * q + 3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave-private tile: program order within the wave is enough
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + srow;
                const float4 u0 = *reinterpret_cast<const float4*>(wl + row * VB_WS + k8), u1 = *reinterpret_cast<const float4*>(wl + row * VB_WS + k8 + 4);
                float u[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
                if (residual) {
                    const float cv[8] = {c0[it].x, c0[it].y, c0[it].z, c0[it].w, c1[it].x, c1[it].y, c1[it].z, c1[it].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] -= s8[e] * cv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) nsq[e] = fmaf(u[e], u[e], nsq[e]);      // (the norms come from the fp32 values)
                const uint4 w = make_uint4(vb_pack2(u[0], u[1]), vb_pack2(u[2], u[3]), vb_pack2(u[4], u[5]), vb_pack2(u[6], u[7]));
                *reinterpret_cast<uint4*>(g.out + ((int64_t)b * D + d0 + row) * K + kcol) = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                  // all reads of the tile done before the next column tile overwrites it
        }
    }
    // the slab's square norm per cluster: the eight d-row classes of a lane group meet by shuffles, the two column groups through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = nsq[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        nsq[e] = v;
    }
    if (lane < 8) {
        *reinterpret_cast<float4*>(red + cg * 256 + cp * 64 + k8) = make_float4(nsq[0], nsq[1], nsq[2], nsq[3]);
        *reinterpret_cast<float4*>(red + cg * 256 + cp * 64 + k8 + 4) = make_float4(nsq[4], nsq[5], nsq[6], nsq[7]);
    }
    __syncthreads();
    if (tid < 256) g.colsq_part[((int64_t)b * P + p) * K + kbase + tid] = red[tid] + red[256 + tid];
}

}  // namespace lpm

// Column slabs per clip of the clip-wide bf16 form (0: shape not supported -- K a multiple of 256, D a multiple of 32 with at least 12
// column tiles so that both column groups of every slab have work)
extern "C" int lpm_vlad_clip16_slabs(int D, int K) {
    if (K <= 0 || K % 256 != 0 || D <= 0 || D % 32 != 0 || D / 32 < 12) return 0;
    const int DT = D / 32;
    return (DT + lpm::VB_NCT - 1) / lpm::VB_NCT;
}

// K2 for bf16 storage on clip-wide items: the contract of lpm_vlad_aggregate_tiles3_fwd_bf16 (at = lpm_assign_tiles_bf16, xt = the plain
// bf16 frame tiles, 4 ceil(T / 64) steps per clip; nrm [B, D, K] bf16 un-normalised sums d-major, asum [B, K]) except that colsq_part is
// [B, P, K] with P = lpm_vlad_clip16_slabs(D, K) (lpm_vlad_row_scales / lpm_vlad_finalize2_fwd take P as an argument).
extern "C" int lpm_vlad_aggregate_clip_fwd_bf16(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                                void* nrm_bf16, float* asum, float* colsq_part, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && nrm_bf16 && asum && colsq_part, LPM_ERR_BADARG, "lpm_vlad_aggregate_clip_fwd_bf16: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_clip_fwd_bf16: RESIDUAL needs centres");
    const int P = lpm_vlad_clip16_slabs(D, K);
    LPM_REQUIRE(B > 0 && T > 0 && P > 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_clip_fwd_bf16: need K %% 256 == 0 and D %% 32 == 0, D >= 384 (D=%d K=%d)", D, K);
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)nrm_bf16) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_clip_fwd_bf16: pointers must be 16-byte aligned");
    VBArgs g{};
    g.at = (const uint4*)at; g.xt = (const uint4*)xt; g.centres = centres;
    g.D = D; g.K = K; g.S = 4 * ((T + 63) / 64); g.P = P; g.KH = K / 256; g.residual = residual;
    g.out = (unsigned short*)nrm_bf16; g.asum = asum; g.colsq_part = colsq_part;
    static const int ns = [] { const char* e = getenv("LPM_VB_NS"); return (e && e[0] == '4') ? 4 : 5; }();
    const size_t ring = (size_t)ns * VB_STAGE, epi = (size_t)VB_EPI + VB_TAIL;
    const size_t lds = ring > epi ? ring : epi;
    void (*kern)(const VBArgs) = ns == 4 ? vlad_clip16_kernel<4> : vlad_clip16_kernel<5>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_clip_fwd_bf16: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid((unsigned)(B * g.KH * P));
    hipEvent_t e0, e1;
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))      // (the video stream's launches only: lpm_common.h)
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, g);
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, g);
    return check_launch("lpm_vlad_aggregate_clip_fwd_bf16");
}
