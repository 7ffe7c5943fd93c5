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
// 18-20 KB per 16-frame step (8 assignment + 10-12 frame pieces: two or three per wave), 5-stage ring of 24 KB slots = 120 KB of LDS, LDS-DMA
// through buffer loads (resource + step offset in SGPRs).
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
constexpr int VB_SLOTS = 20;                       // 1 KB slots per stage: 8 assignment pieces, VB_NCT frame pieces
constexpr int VB_PW = 3;                            // pieces per wave and step at most (slot = wave + 8 j)
constexpr int VB_STAGE = VB_SLOTS * 1024;
constexpr int VB_WS = 68;                          // epilogue tile row stride in floats (272 B: 16-byte aligned rows, odd multiple of 16 B)
constexpr int VB_EPI = 8 * 32 * VB_WS * 4;         // 8 wave-private tiles [32 d][64 clusters] fp32: overlays the ring
constexpr int VB_TAIL = (256 + 512) * 4;           // assignment sums [256] + square-norm partials [2][256], behind the tiles

struct VBArgs {
    const uint4* at;            // assignment tiles [b][K/32][S][lane]   (lpm_assign_tiles_bf16: plain bf16, 1 KB per (tile, step))
    const uint4* xt;            // frame tiles      [b][S][D/32][lane]   (lpm_frame_apply_tiles_bf16 / lpm_split_frames_bf16)
    const float* centres;       // [D, K] (cluster_weights2) or null
    int D, K, S, P, KH, residual;
    int dbg, delay_us;          // measurement only (LPM_VB_DBG, LPM_VB_DELAY): see the kernel
    unsigned short* out;        // [B, D, K] bf16 un-normalised residual sums, d-major
    float* asum;                // [B, K]
    float* colsq_part;          // [B, P, K]
};

// DBG (measurement builds of the SAME kernel, LPM_VB_DBG=n, results are garbage): 1 no main loop, 2 no result stores, 4 no DMA, 8 no MFMAs,
// 16 no centre loads, 32 no epilogue at all, 64 every second workgroup of the first round starts LPM_VB_DELAY us late (results correct).  The production instantiation (DBG = 0) carries none of these branches.
template <int NS, int DBG, int AUX>
__global__ __launch_bounds__(512) void vlad_clip16_kernel(const VBArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // the ONLY LDS object (guide 5, trap (a))
    const int dbg = DBG ? g.dbg : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int D = g.D, K = g.K, S = g.S, P = g.P, KH = g.KH;
    const int DT = D >> 5, KT = K >> 5;
    // An ITEM = (clip b, cluster half kh, column slab p); the KH * P items of a clip have consecutive ids (one XCD, the same moment).  One item
    // per workgroup.  (Measured and not kept, round 6: a PERSISTENT form -- one workgroup per CU walking items lid, lid + 256, ..., requesting
    // the next item's first stages before it runs the current epilogue, the epilogue's tiles in an LDS region of their own -- 112 us against
    // 85: the item loop around the 192 accumulators cost the epilogue 75 spilled registers, and with one item per workgroup of the same code
    // (LPM_VB_GRID = 768) it was 116 us: what the prefetch hides is worth ~4 us, what the spills cost ~30.)
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    if (DBG && (dbg & 64) && blockIdx.x < 256 && (blockIdx.x & 1)) {
        // measurement: put every second workgroup of the FIRST round g.delay_us behind (do the rounds' load and store phases overlap
        // across CUs once they are out of step?)
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)g.delay_us * 100ull) __builtin_amdgcn_s_sleep(16);      // (100 MHz counter)
    }
    const int b = lid / (KH * P), rem_ = lid % (KH * P);
    const int kh = rem_ / P, p = rem_ % P;
    const int base = DT / P, rem = DT % P;
    const int ncol = base + (p < rem ? 1 : 0);               // column tiles of this slab (<= VB_NCT)
    const int ct0 = p * base + min(p, rem);
    const int cp = wave & 3, cg = wave >> 2;                  // cluster pair, column group
    const int n0 = (ncol + 1) >> 1;                           // group 0: tiles [0, n0), group 1: [n0, ncol)
    const int cnt = cg == 0 ? n0 : ncol - n0;                 // this wave's column tiles (<= VB_NJ)
    const int jt0 = cg * n0;

    // this wave's pieces of a stage: slot = wave + 8 j.  slots 0..7: assignment tile kh * 8 + slot; slots 8..8 + ncol - 1: frame tile
    // ct0 + slot - 8.  Every wave has pieces j = 0, 1; the third exists for wave + 16 < 8 + ncol only (wave-uniform `has3`): no dummy pieces
    // (a first version padded every wave to three so that one vmcnt constant served all -- a fifth of the DMA traffic -- and ran no slower
    // or faster: the DMA volume is not what bounds this kernel).  The waits come in two flavours, selected by a scalar branch.
    __amdgpu_buffer_rsrc_t rsrc[VB_PW];
    unsigned sstep[VB_PW];                         // bytes per step
    const bool has3 = wave + 16 < 8 + ncol;
#pragma unroll
    for (int j = 0; j < VB_PW; ++j) {
        const int sl = wave + 8 * j;
        const uint4* src;
        if (sl < 8) {
            src = g.at + (((int64_t)b * KT + kh * 8 + sl) * S) * 64;
            sstep[j] = 1024u;
        } else {
            src = g.xt + (((int64_t)b * S) * DT + ct0 + min(sl - 8, ncol - 1)) * 64;
            sstep[j] = (unsigned)DT * 1024u;
        }
        rsrc[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, 0xffffffff, 0x00020000);
    }
    const unsigned lane_off = (unsigned)lane * 16u;
    const bool no_dma = (dbg & 4) != 0;
    auto issue = [&](int s) {
        if (DBG && no_dma) return;
        unsigned char* st = smem + (s % NS) * VB_STAGE;
#pragma unroll
        for (int j = 0; j < VB_PW; ++j)
            if (j < 2 || has3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc[j], (__attribute__((address_space(3))) void*)(st + (wave + 8 * j) * 1024), 16, lane_off,
                                                         (unsigned)s * sstep[j], 0, AUX);    // (AUX = 2: the non-temporal policy, LPM_VB_NT)
    };
    // this wave's pieces of a step have landed when at most `steps` younger steps' pieces are outstanding (two or three pieces per step)
#define VB_VMWAIT(steps)                                                                         \
    do {                                                                                         \
        if (has3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (steps)) : "memory");             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (steps)) : "memory");                  \
    } while (0)

    f32x16 acc[2][VB_NJ];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < VB_NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
    float asum_l[2] = {0.f, 0.f};      // assignment sums of clusters 32 (2 cp + m) + l31 over this lane's 8 frames of every step (group-0 waves)

    // ---- main loop.  The ring: step s lives in stage s % NS, NS - 1 steps are requested ahead.  ONE raw s_barrier per step, placed
    // BEHIND the step's second MFMA group (vlad_clip.hip's arrangement): barrier B(s + 1) says "step s + 1 has landed for everyone (every
    // wave waited for its own pieces first) and everyone is past the fragment reads of step s - 1", so stage (s - 1) % NS takes step
    // s + NS - 1 right behind it and the first fragments of step s + 1 are requested while the last MFMA group of step s is still being
    // issued: the matrix pipe does not drain at a step boundary.
    // Fragment reads are inline assembly off two base addresses per stage (A tiles, this wave's B tiles) with immediate offsets and
    // HAND-COUNTED lgkmcnt (LDS returns in order).  A step is three MFMA groups of four -- (a0, a1) x (b0, b1), x (b2, b3), x (b4, b5) --
    // over TWO B-pair register sets P, Q that swap roles every step and TWO A sets that alternate (the loop is unrolled by two):
    //   top of a step, in flight, oldest first:  a (2), first pair (2), second pair (2)
    //   WAIT(2)  group 0 on the first pair;   first pair  <- (b4, b5)
    //   WAIT(2)  group 1 on the second pair;  [vmcnt, barrier, DMA issue]  a' <- next A,  second pair <- next (b0, b1)
    //   WAIT(4)  group 2 on the first pair;   first pair  <- next (b2, b3)          -> the next step's (a, second, first)
    // A wave with fewer than six column tiles reads the slot behind its own -- the other group's first tile, or a slot no DMA writes
    // (whatever bits the LDS holds) -- into accumulators it never stores.
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address
    const unsigned rd_a = smem_lds + lane_off + (unsigned)(2 * cp) * 1024u;
    const unsigned rd_b = smem_lds + lane_off + (unsigned)(8 + jt0) * 1024u;
    struct Pair { vb_u32x4 x, y; };
// ("+v": the new fragment is tied to the register quad of the one it replaces -- vlad_clip.hip: with plain outputs hipcc gives every read a
// fresh quad, fragments the register file and spills accumulator tiles)
#define VB_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(base), "n"(off))
#define VB_RD_PAIR(f, base, j0)                  \
    do {                                         \
        VB_RD(f.x, base, (j0) * 1024);           \
        VB_RD(f.y, base, (j0) * 1024 + 1024);    \
    } while (0)
#define VB_WAIT(n)                                                  \
    do {                                                            \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");  \
        __builtin_amdgcn_sched_barrier(0);                          \
    } while (0)
    const bool no_mfma = (dbg & 8) != 0;
    auto group = [&](int j0, const Pair& f, const Pair& a) {
        if (DBG && no_mfma) {
            asm volatile("" ::"v"(f.x), "v"(f.y), "v"(a.x), "v"(a.y));
            return;
        }
        acc[0][j0] = vb_mfma(a.x, f.x, acc[0][j0]);
        acc[1][j0] = vb_mfma(a.y, f.x, acc[1][j0]);
        acc[0][j0 + 1] = vb_mfma(a.x, f.y, acc[0][j0 + 1]);
        acc[1][j0 + 1] = vb_mfma(a.y, f.y, acc[1][j0 + 1]);
        __builtin_amdgcn_sched_barrier(0);
    };
    const vb_u32x4 zero4 = {0u, 0u, 0u, 0u};
    Pair pa = {zero4, zero4}, pb = pa, a0 = pa, a1 = pa;
    const int nloop = (DBG && (dbg & 1)) ? 0 : S;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nloop) issue(s);
    if (nloop > 0) {
        // step 0 has landed (this wave's pieces; min(NS - 2, S - 1) younger steps stay in flight), then for everyone
        const int behind = min(NS - 2, S - 1);
        if (behind >= 3) VB_VMWAIT(3);
        else if (behind == 2) VB_VMWAIT(2);
        else if (behind == 1) VB_VMWAIT(1);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        VB_RD_PAIR(a0, rd_a, 0);
        VB_RD_PAIR(pa, rd_b, 0);
        VB_RD_PAIR(pb, rd_b, 2);
    }
    // one step: a = this step's A fragments, an = the next step's; first / second = the B-pair sets in this step's roles
    auto body = [&](int s, Pair& a, Pair& an, Pair& first, Pair& second) {
        const unsigned so = (unsigned)((s % NS) * VB_STAGE), son = (unsigned)(((s + 1) % NS) * VB_STAGE);
        const bool more = s + 1 < S;               // workgroup-uniform
        VB_WAIT(2);
        group(0, first, a);
        VB_RD_PAIR(first, rd_b + so, 4);
        if (cg == 0) {                             // wave-uniform: the cluster pair's first column group keeps the assignment sums
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                asum_l[0] += vb_bf(a.x[w2] & 0xffffu) + vb_bf(a.x[w2] >> 16);
                asum_l[1] += vb_bf(a.y[w2] & 0xffffu) + vb_bf(a.y[w2] >> 16);
            }
            // pinned here (vlad_clip.hip: left free, hipcc sinks these additions behind the request for the NEXT step's A fragments, keeps
            // both generations alive and copies a register whose ds_read has just been issued)
            asm volatile("" : "+v"(asum_l[0]), "+v"(asum_l[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        VB_WAIT(2);
        group(2, second, a);
        if (more) {
            // B(s + 1): this wave's pieces of step s + 1 have landed (min(NS - 3, S - 2 - s) younger steps stay in flight) ...
            const int behind = min(NS - 3, S - 2 - s);
            if (behind >= 2) VB_VMWAIT(2);
            else if (behind == 1) VB_VMWAIT(1);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // ... everyone's have, and everyone is past the reads of step s - 1
            asm volatile("" ::: "memory");
            if (s + NS - 1 < S) issue(s + NS - 1);
            __builtin_amdgcn_sched_barrier(0);
            VB_RD_PAIR(an, rd_a + son, 0);
            VB_RD_PAIR(second, rd_b + son, 0);
            VB_WAIT(4);                            // oldest first: first = (b4, b5) (2), an (2), second (2)
        } else {
            VB_WAIT(0);
        }
        group(4, first, a);
        if (more) {
            VB_RD_PAIR(first, rd_b + son, 2);      // the state the next step expects: an (2), second = its first pair (2), first = its second (2)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s = 0; s < nloop; s += 2) {
        body(s, a0, a1, pa, pb);
        if (s + 1 < S) body(s + 1, a1, a0, pb, pa);
    }
#undef VB_RD
#undef VB_RD_PAIR
#undef VB_WAIT
#undef VB_VMWAIT
    __syncthreads();         // no DMA in flight (the last step waited for vmcnt(0)), all fragment reads done: the ring is scratch now
    if (DBG && (dbg & 32)) return;

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
    const bool residual = g.residual != 0 && !(DBG && (dbg & 16));
    const bool do_store = !(DBG && (dbg & 2));
    float s8[8], nsq[8];
    {
        const float4 sa = *reinterpret_cast<const float4*>(sas + cp * 64 + k8), sb = *reinterpret_cast<const float4*>(sas + cp * 64 + k8 + 4);
        s8[0] = sa.x; s8[1] = sa.y; s8[2] = sa.z; s8[3] = sa.w; s8[4] = sb.x; s8[5] = sb.y; s8[6] = sb.z; s8[7] = sb.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) nsq[e] = 0.f;
    const int kcol = kbase + cp * 64 + k8;                    // this lane's first cluster in the store pass
    // (Two centre tiles in flight -- tile j + 1 requested as soon as tile j's accumulators have gone to LDS, vlad_clip.hip's arrangement --
    // was tried: 64 registers of centres beside 192 accumulators spilled 92 registers.  One tile's 32, requested at the top of its
    // iteration, under the LDS round trip of the accumulators.)
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
                if (do_store) *reinterpret_cast<uint4*>(g.out + ((int64_t)b * D + d0 + row) * K + kcol) = w;
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
    static const int dbg = [] { const char* e = getenv("LPM_VB_DBG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    static const int delay = [] { const char* e = getenv("LPM_VB_DELAY"); return e ? atoi(e) : 8; }();
    g.delay_us = delay;
    constexpr int NS = 4;
    const size_t ring = (size_t)NS * VB_STAGE, epi = (size_t)VB_EPI + VB_TAIL;
    const size_t lds = ring > epi ? ring : epi;
    static const int nt = [] { const char* e = getenv("LPM_VB_NT"); return (e && e[0] == '0') ? 0 : 1; }();
    void (*kern)(const VBArgs) = dbg ? (nt ? vlad_clip16_kernel<NS, 1, 2> : vlad_clip16_kernel<NS, 1, 0>)
                                     : (nt ? vlad_clip16_kernel<NS, 0, 2> : vlad_clip16_kernel<NS, 0, 0>);
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
