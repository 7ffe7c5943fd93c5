// K2, fifth form (round 4): clip-wide items -- the residual aggregation of frame_level_models.py:2803-2817 with the L2 -> LDS
// traffic cut to 1.35 x the algorithmic input instead of 3.2 x.
//
// The third form (vlad_tiles3.hip: 128 clusters x 128 columns per workgroup) moves 389 MB through the LDS-DMA path per launch at
// cfg-2 for 123 MB of algorithmic input -- every clip's assignment tiles are re-read by D/128 = 8 column slabs, its frame tiles by
// K/128 = 2 cluster slabs -- and that delivery path (7-8 TB/s chip-wide, measured in rounds 2 and 3) is what bounds it at 57-62 us.
// Here, for K = 256, a 512-thread workgroup owns ALL 256 clusters x a THIRD of a clip's columns (11 / 11 / 10 column tiles of 32 at
// D = 1024): 3 workgroups per clip, 240 at cfg-2 -- one round on 256 CUs, one workgroup per CU.  Per clip the frame tiles come in
// once (1.2 MB) and the assignment tiles three times (0.9 MB): 168 MB per launch.  Wave w owns cluster tile w against every column
// tile of the slab: 11 accumulator tiles (176 registers), per 16-frame step 2 + 22 fragment reads for 33 MFMAs.
//
// Operand roles are SWAPPED with respect to the other forms -- the frame tile is the MFMA's A operand, the assignment tile its B
// operand (the fragment format is the same for both, tile_gemm.h) -- so that an accumulator register holds FOUR CONSECUTIVE COLUMNS d
// of ONE cluster (acc[e][r]: cluster = lane & 31, d = 32 e + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)):
//   * the per-cluster quantities -- assignment sum, residual factor, square norm over the slab -- are per-LANE: no LDS reductions, no
//     workgroup barrier in the epilogue; a half-wave shuffle at the very end;
//   * the k-major result [B, K, D] (the lazily normalised descriptor of NetVladV1, App. C5) leaves through a wave-private LDS tile as
//     float4 along d, 8 lanes per 128-byte row piece; the tile is written with ds_write_b128 (a lane's four d are contiguous).
// 40 KB per step (16 assignment pieces + 22 frame pieces + 2 idle slots so that every wave issues five pieces and the in-order vmcnt
// is one constant), 3-stage ring = 120 KB of LDS.
#include "lpm_common.h"

namespace lpm {

typedef __bf16 vc_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned vc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 vc_mfma(vc_u32x4 a, vc_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vc_bf16x8, a), __builtin_bit_cast(vc_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float vc_bf(unsigned h) { return __uint_as_float(h << 16); }

constexpr int VC_NCT = 11;                         // column tiles per workgroup (accumulator tiles per wave)
constexpr int VC_SLOTS = 40;                       // 1 KB slots per stage: 16 assignment pieces, 2 * VC_NCT frame pieces, 2 idle
constexpr int VC_PW = VC_SLOTS / 8;                // pieces per wave and step
constexpr int VC_STAGE = VC_SLOTS * 1024;
constexpr int VC_TS = 36;                          // epilogue tile row stride in floats (144 B: 16-byte aligned rows)
constexpr int VC_EPI = 8 * 2 * 32 * VC_TS * 4;     // 8 waves x 2 tiles: overlays the ring
static_assert(VC_EPI <= 3 * VC_STAGE, "the epilogue tiles overlay the ring");
static_assert(16 + 2 * VC_NCT <= VC_SLOTS, "slots");

struct VCArgs {
    const uint4* at;            // assignment tiles [b][K/32][S][plane][lane]   (lpm_assign_tiles)
    const uint4* xt;            // frame tiles      [b][S][D/32][plane][lane]   (lpm_split_frames / lpm_frame_apply_tiles)
    const float* centres;       // [D, K] (cluster_weights2) or null
    int T, D, S, P, residual;   // K = 256; P column slabs per clip
    float* out;                 // [B, K, D] un-normalised residual sums (dmajor: [B, D, K])
    int dmajor;                 // the result leaves d-major (NetVladV2's lazily normalised descriptor, round 4): straight from the accumulators
    float* asum;                // [B, K]
    float* colsq_part;          // [B, P, K]
    int dbg;                    // measurement only (LPM_VC_DBG): 1 no main loop, 2 no stores, 4 no DMA, 8 no MFMAs, 16 no residual loads
};

template <int VC_NS, int AUX, int ABL>
__global__ __launch_bounds__(512, 2) void vlad_clip_kernel(const VCArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // the ONLY LDS object (guide 5, trap (a))
    constexpr int K = 256, KT = 8;
    const int dbg = ABL ? g.dbg : 0;               // (the production instantiation carries no measurement branches)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int D = g.D, S = g.S, P = g.P;
    const int DT = D >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);        // the P slabs of a clip: consecutive ids, one XCD, the same moment
    const int b = lid / P, p = lid % P;
    const int base = DT / P, rem = DT % P;
    const int ncol = base + (p < rem ? 1 : 0);               // column tiles of this slab (<= VC_NCT)
    const int ct0 = p * base + min(p, rem);

    // this wave's five pieces of a stage: slot = wave + 8 j.  slots 0..15: assignment tile slot >> 1, plane slot & 1; slots 16..37:
    // frame tile ct0 + min((slot - 16) >> 1, ncol - 1) (a narrower slab re-reads its last tile: its accumulator is never stored);
    // slots 38, 39: idle (they re-read assignment piece 0 so that every wave has the same number of operations in flight)
    // LDS-DMA through BUFFER loads (buffer_load_dwordx4 ... lds): the resource (base address) and the step offset are wave-uniform and
    // live in SGPRs, the only VGPR is the lane's 16-byte offset -- five 64-bit per-lane pointers advanced every step (global_load_lds)
    // cost 10 VGPRs and 10 VALU operations per step that the accumulators need.
    __amdgpu_buffer_rsrc_t rsrc[VC_PW];
    unsigned sstep[VC_PW];                         // bytes per step
#pragma unroll
    for (int j = 0; j < VC_PW; ++j) {
        const int sl = wave + 8 * j;
        const uint4* base;
        if (sl < 16 || sl >= 16 + 2 * VC_NCT) {
            const int q = sl < 16 ? sl : 0;
            base = g.at + ((((int64_t)b * KT + (q >> 1)) * S) * 2 + (q & 1)) * 64;
            sstep[j] = 128u * 16u;
        } else {
            const int q = sl - 16;
            base = g.xt + ((((int64_t)b * S) * DT + ct0 + min(q >> 1, ncol - 1)) * 2 + (q & 1)) * 64;
            sstep[j] = (unsigned)DT * 128u * 16u;
        }
        rsrc[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(base), 0, 0xffffffff, 0x00020000);
    }
    const unsigned lane_off = (unsigned)lane * 16u;
    const bool no_dma = (dbg & 4) != 0;
    auto issue = [&](int s) {
        if (no_dma) return;
        unsigned char* st = smem + (s % VC_NS) * VC_STAGE;
#pragma unroll
        for (int j = 0; j < VC_PW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc[j], (__attribute__((address_space(3))) void*)(st + (wave + 8 * j) * 1024), 16, lane_off,
                                                     (unsigned)s * sstep[j], 0, AUX);
    };

    f32x16 acc[VC_NCT];
#pragma unroll
    for (int e = 0; e < VC_NCT; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    float asum_l = 0.f;      // assignment sum of cluster 32 wave + l31 over this lane's 8 frames of every step

    // ---- main loop.  The ring: step s lives in stage s % NS, NS - 1 steps are requested ahead.
    // ONE raw s_barrier per step, placed in the MIDDLE of the step's MFMA stream: barrier B(s + 1), taken half-way through step s,
    // says "step s + 1 has landed for everyone (every wave waited for its own pieces first) and everyone is past step s - 1", so stage
    // (s - 1) % NS takes step s + NS - 1 right behind it -- and the fragments of step s + 1 are requested while the last MFMAs of step
    // s are still being issued: the matrix pipe does not drain at a step boundary (with the barrier AT the boundary every wave of the
    // workgroup started a step with an empty pipe and a cold LDS queue: 300-400 idle cycles per 2100-cycle step).
    // Fragment reads are inline assembly off one base address per stage with immediate offsets and HAND-COUNTED lgkmcnt (LDS returns in
    // order): a fragment set is requested one MFMA group ahead and waited for only in front of its own MFMAs (left to hipcc every
    // wait was lgkmcnt(0), i.e. also for the reads just issued).
    // Pairs of column tiles, term-major inside a pair: two independent accumulators between dependent MFMAs; every accumulator sees
    // xh.ah, xl.ah, xh.al in this order (the other forms' order of the three terms).
    static_assert(VC_NCT == 11, "the hand-counted waits below are written for 11 column tiles: pairs 0, 2, 4, 6, 8 and the single tile 10");
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address
    struct Pair { vc_u32x4 h0, l0, h1, l1; };
    struct AFrag { vc_u32x4 h, l; };
// ("+v": the new fragment is tied to the register quad of the one it replaces -- with plain outputs hipcc gave every read a fresh quad,
// fragmented the register file and spilled two accumulator tiles per step)
#define VC_RD(dst, base, slot) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(base), "n"((slot) * 1024))
#define VC_RD_PAIR(f, base, e0)                                \
    do {                                                       \
        VC_RD(f.h0, base, 16 + (e0) * 2 + 0);                  \
        VC_RD(f.l0, base, 16 + (e0) * 2 + 1);                  \
        if ((e0) + 1 < VC_NCT) {                               \
            VC_RD(f.h1, base, 16 + (e0) * 2 + 2);              \
            VC_RD(f.l1, base, 16 + (e0) * 2 + 3);              \
        }                                                      \
    } while (0)
#define VC_RD_A(f, base)                                       \
    do {                                                       \
        const unsigned sa_ = (base) + (unsigned)wave * 2048u;  \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "+v"(f.h), "+v"(f.l) : "v"(sa_)); \
    } while (0)
#define VC_WAIT(n)                                             \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)
    const bool no_mfma = (dbg & 8) != 0;
    auto mfma_pair = [&](int e0, const Pair& f, const AFrag& a) {
        if (ABL && no_mfma) {
            asm volatile("" ::"v"(f.h0), "v"(f.l0), "v"(f.h1), "v"(f.l1));
            return;
        }
        if (e0 + 1 < VC_NCT) {
            acc[e0] = vc_mfma(f.h0, a.h, acc[e0]);
            acc[e0 + 1] = vc_mfma(f.h1, a.h, acc[e0 + 1]);
            acc[e0] = vc_mfma(f.l0, a.h, acc[e0]);
            acc[e0 + 1] = vc_mfma(f.l1, a.h, acc[e0 + 1]);
            acc[e0] = vc_mfma(f.h0, a.l, acc[e0]);
            acc[e0 + 1] = vc_mfma(f.h1, a.l, acc[e0 + 1]);
        } else {
            acc[e0] = vc_mfma(f.h0, a.h, acc[e0]);
            acc[e0] = vc_mfma(f.l0, a.h, acc[e0]);
            acc[e0] = vc_mfma(f.h0, a.l, acc[e0]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const vc_u32x4 zero4 = {0u, 0u, 0u, 0u};
    Pair fa = {zero4, zero4, zero4, zero4}, fb = fa;
    AFrag a0 = {zero4, zero4};
    const int nloop = (dbg & 1) ? 0 : S;
#pragma unroll
    for (int s = 0; s < VC_NS - 1; ++s)
        if (s < nloop) issue(s);
    if (nloop > 0) {
        // step 0 has landed (this wave's pieces; min(NS - 2, S - 1) younger steps stay in flight), then for everyone
        if (VC_NS >= 4 && 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VC_PW) : "memory");
        else if (1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VC_PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned sb0 = smem_lds + lane_off;
        VC_RD_A(a0, sb0);
        VC_RD_PAIR(fa, sb0, 0);
        VC_RD_PAIR(fb, sb0, 2);
    }
    // in flight at the top of a step, oldest first: [this step's assignment fragments (2), pair 0 (4)] in either order, pair 2 (4)
    auto body = [&](int s, AFrag& a) {
        const unsigned sb = smem_lds + (unsigned)((s % VC_NS) * VC_STAGE) + lane_off;
        const unsigned sbn = smem_lds + (unsigned)(((s + 1) % VC_NS) * VC_STAGE) + lane_off;
        const bool more = s + 1 < S;               // workgroup-uniform
        VC_WAIT(4);
        mfma_pair(0, fa, a);
        VC_RD_PAIR(fa, sb, 4);
        // assignment sum of this lane's cluster: its 8 frames of the step, hi + lo (VALU work under the MFMAs just issued)
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const unsigned h = a.h[w2], l = a.l[w2];
            asum_l += (vc_bf(h & 0xffffu) + vc_bf(l & 0xffffu)) + (vc_bf(h >> 16) + vc_bf(l >> 16));
        }
        // Pinned HERE (a volatile statement keeps its place among the volatile fragment reads): left free, hipcc sank these additions
        // behind the request for the NEXT step's assignment fragments, kept both generations of `a` alive and copied the new one --
        // v_mov of a register whose ds_read had just been issued -- into the loop-carried registers: stale bits whenever the LDS was
        // slower than the copy (found by tests/test_gpu_determinism.py with a second process on the GPU).  Fragments that cross the
        // loop's back edge must be the SAME registers at both ends: tests/test_build_flags.py checks the loop for register copies.
        asm volatile("" : "+v"(asum_l));
        __builtin_amdgcn_sched_barrier(0);
        VC_WAIT(4);
        mfma_pair(2, fb, a);
        VC_RD_PAIR(fb, sb, 6);
        if (more) {
            // B(s + 1): this wave's pieces of step s + 1 have landed (min(NS - 3, S - 2 - s) younger steps stay in flight) ...
            if (VC_NS >= 4 && s + 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VC_PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // ... everyone's have, and everyone is past step s - 1
            asm volatile("" ::: "memory");
            if (s + VC_NS - 1 < S) issue(s + VC_NS - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        VC_WAIT(4);
        mfma_pair(4, fa, a);
        VC_RD_PAIR(fa, sb, 8);
        VC_WAIT(4);
        mfma_pair(6, fb, a);
        VC_RD_PAIR(fb, sb, 10);                    // (one tile: 2 reads)
        // (no MFMA inside a conditional block: two copies of an MFMA with different accumulator registers at the join cost hipcc 64
        // spilled registers per step)
        VC_WAIT(2);                                // oldest first: pair 8 (4), tile 10 (2)
        mfma_pair(8, fa, a);
        if (more) {
            VC_RD_PAIR(fa, sbn, 0);                // tile 10 (2), next pair 0 (4)
            VC_WAIT(4);
        } else {
            VC_WAIT(0);
        }
        mfma_pair(10, fb, a);
        if (more) {
            // the assignment fragments have ONE register set: the next step's are requested behind the last MFMA that reads this step's
            VC_RD_A(a, sbn);
            VC_RD_PAIR(fb, sbn, 2);                // next pair 0 (4), next a (2), next pair 2 (4): the state the next step expects
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s = 0; s < nloop; ++s) body(s, a0);
#undef VC_RD
#undef VC_RD_PAIR
#undef VC_RD_A
#undef VC_WAIT
    __syncthreads();         // no DMA in flight (the last step waited for vmcnt(0)), all fragment reads done: the ring is scratch now

    // ---- epilogue, wave by wave, no workgroup barrier: residual + square norm in the accumulators' layout (everything per cluster is per
    // lane), then the tile through a wave-private LDS tile [32 clusters][32 d] into float4 stores along d.
    asum_l += __shfl_xor(asum_l, 32, 64);                     // both half-waves hold the cluster's total
    const int kcl = wave * 32 + l31;                          // this lane's cluster
    if (p == 0 && half == 0) g.asum[(int64_t)b * K + kcl] = asum_l;
    float* wl = reinterpret_cast<float*>(smem) + wave * (2 * 32 * VC_TS);
    const int srow = lane >> 3, c4 = (lane & 7) * 4;          // store pass: cluster row it * 8 + srow, columns c4 .. c4 + 3 of the tile
    const bool residual = g.residual != 0 && !(dbg & 16);
    // centres of tile e for this lane: [d = 32 (ct0 + e) + 8 q + 4 half + j][k = kcl], 16 values; consecutive lanes = consecutive k
    const float* cbase = g.centres + (int64_t)(ct0 * 32 + 4 * half) * K + kcl;
    float cen[2][16];                                          // two tiles in flight
    auto load_centres = [&](int e, float (&dst)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[4 * q + j] = cbase[(int64_t)(32 * e + 8 * q + j) * K];
    };
    if (residual) {
        load_centres(0, cen[0]);
        if (1 < ncol) load_centres(1, cen[1]);
    }
    float nsq = 0.f;
    float* ob = g.out + ((int64_t)b * K + wave * 32) * D + ct0 * 32;
    const bool do_store = !(dbg & 2);
#pragma unroll
    for (int e = 0; e < VC_NCT; ++e) {
        if (e < ncol) {                                       // workgroup-uniform
            float* tl = wl + (e & 1) * (32 * VC_TS);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[j] = acc[e][4 * q + j];
                    if (residual) u[j] -= asum_l * cen[e & 1][4 * q + j];
                    nsq = fmaf(u[j], u[j], nsq);
                }
                    if (g.dmajor) {
                    // d-major [B, D, K]: an accumulator register is one d of 32 consecutive clusters -- a half-wave's store is one 128-byte
                    // line of a row, no transposing tile
                    if (do_store) {
                        float* od = g.out + ((int64_t)b * D + ct0 * 32 + 32 * e + 8 * q + 4 * half) * K + kcl;
#pragma unroll
                        for (int j = 0; j < 4; ++j) od[(int64_t)j * K] = u[j];
                    }
                } else {
                    *reinterpret_cast<float4*>(tl + l31 * VC_TS + 8 * q + 4 * half) = make_float4(u[0], u[1], u[2], u[3]);
                }
            }
            if (residual && e + 2 < ncol) load_centres(e + 2, cen[e & 1]);     // in flight under the next tiles' transposes and stores
            if (g.dmajor) continue;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // wave-private tile: program order within the wave is enough
            __builtin_amdgcn_wave_barrier();
            if (do_store) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int kr = it * 8 + srow;
                    *reinterpret_cast<float4*>(ob + (int64_t)kr * D + e * 32 + c4) = *reinterpret_cast<const float4*>(tl + kr * VC_TS + c4);
                }
            }
            // (two tiles alternate: tile e + 2 is written two iterations later, behind this tile's reads in program order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    nsq += __shfl_xor(nsq, 32, 64);
    if (half == 0) g.colsq_part[((int64_t)b * P + p) * K + kcl] = nsq;
}

}  // namespace lpm

// Column slabs per clip of the clip-wide form (0: shape not supported -- K must be 256, D a multiple of 32 with at least 3 tiles)
extern "C" int lpm_vlad_clip_slabs(int D, int K) {
    if (K != 256 || D <= 0 || D % 32 != 0 || D / 32 < 3) return 0;
    const int DT = D / 32;
    return (DT + lpm::VC_NCT - 1) / lpm::VC_NCT;
}

// K2 for the lazily normalised k-major descriptor (as lpm_vlad_aggregate_raw_kmajor_fwd), clip-wide items: raw_kmajor [B, K, D]
// un-normalised residual sums, asum [B, K], colsq_part [B, P, K] with P = lpm_vlad_clip_slabs(D, K); lpm_vlad_row_scales(colsq_part, P, ...)
// follows.  at / xt: lpm_assign_tiles / frame tiles (split-bf16).
static int vlad_clip_impl(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags, float* raw_kmajor,
                          float* asum, float* colsq_part, int dmajor, lpm_stream_t stream);
extern "C" int lpm_vlad_aggregate_clip_kmajor_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                                  int flags, float* raw_kmajor, float* asum, float* colsq_part, lpm_stream_t stream) {
    return vlad_clip_impl(at, xt, centres, B, T, D, K, flags, raw_kmajor, asum, colsq_part, 0, stream);
}
// ... the same kernel leaving the un-normalised sums d-major [B, D, K] (the reference's own layout, frame_level_models.py:2817-2821:
// NetVladV2's lazily normalised descriptor; what lpm_vlad_aggregate_tiles3_fwd writes with LPM_VLAD_RESIDUAL only)
extern "C" int lpm_vlad_aggregate_clip_dmajor_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                                  int flags, float* raw_dmajor, float* asum, float* colsq_part, lpm_stream_t stream) {
    return vlad_clip_impl(at, xt, centres, B, T, D, K, flags, raw_dmajor, asum, colsq_part, 1, stream);
}
static int vlad_clip_impl(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags, float* raw_kmajor,
                          float* asum, float* colsq_part, int dmajor, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && raw_kmajor && asum && colsq_part, LPM_ERR_BADARG, "lpm_vlad_aggregate_clip_kmajor_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_clip_kmajor_fwd: RESIDUAL needs centres");
    const int P = lpm_vlad_clip_slabs(D, K);
    LPM_REQUIRE(B > 0 && T > 0 && P > 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_clip_kmajor_fwd: need K == 256 and D %% 32 == 0, D >= 96 (D=%d K=%d)", D, K);
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)raw_kmajor) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_clip_kmajor_fwd: pointers must be 16-byte aligned");
    VCArgs g{};
    g.at = (const uint4*)at; g.xt = (const uint4*)xt; g.centres = centres;
    g.T = T; g.D = D; g.S = (T + 15) / 16; g.P = P; g.residual = residual;
    g.out = raw_kmajor; g.asum = asum; g.colsq_part = colsq_part; g.dmajor = dmajor;
    static const int dbg = [] { const char* e = getenv("LPM_VC_DBG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    static const int ns = [] { const char* e = getenv("LPM_VC_NS"); return (e && e[0] == '3') ? 3 : 4; }();
    static const int nt = [] { const char* e = getenv("LPM_VC_NT"); return (e && e[0] == '0') ? 0 : 1; }();
    const size_t lds = (size_t)ns * VC_STAGE;
    void (*kern)(const VCArgs);
    if (dbg) kern = ns == 4 ? (nt ? vlad_clip_kernel<4, 2, 1> : vlad_clip_kernel<4, 0, 1>) : (nt ? vlad_clip_kernel<3, 2, 1> : vlad_clip_kernel<3, 0, 1>);
    else kern = ns == 4 ? (nt ? vlad_clip_kernel<4, 2, 0> : vlad_clip_kernel<4, 0, 0>) : (nt ? vlad_clip_kernel<3, 2, 0> : vlad_clip_kernel<3, 0, 0>);
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_clip_kmajor_fwd: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid((unsigned)(B * P));
    hipEvent_t e0, e1;
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, g);
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, g);
    return check_launch("lpm_vlad_aggregate_clip_kmajor_fwd");
}
