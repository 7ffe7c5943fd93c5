// K2, third form: LDS-shared split-bf16 tiles fed by LDS-DMA (global_load_lds), 128 clusters x 128 columns per
// workgroup.
//
// vlad_tiles.hip streams every fragment from L2 straight into registers with (clip, 32-cluster slab) workgroups:
// HBM traffic is algorithmic, but each clip's frames are re-read by its K/32 = 8 slab workgroups, 1.09 GB of
// L2 -> CU traffic per launch, and that delivery path (not HBM, not the matrix pipe) bounds it at ~100 us.
// Here a 512-thread workgroup owns 128 clusters x 128 columns of one clip: per 16-frame step it brings 4 cluster
// tiles + 4 column tiles (hi and lo planes, 16 KB) into LDS ONCE with LDS-DMA -- the tile format is lane-linear,
// so a 1 KB fragment is exactly one wave-wide global_load_lds_dwordx4 -- and its 8 waves (2 cluster pairs x 4
// column tiles, 64 x 32 accumulator tile each) read their fragments from LDS with conflict-free ds_read_b128.
// L2 -> CU traffic drops to A x D/128 + x x K/128 = 394 MB at cfg-2.  A 3-stage ring keeps two steps of DMA in
// flight across ONE raw s_barrier per step with hand-counted vmcnt (cdna guide: never __syncthreads with glds in
// flight); 48 KB of ring + 2.5 KB of reduction scratch and <= 80 VGPRs put three workgroups on a CU.  Because a
// workgroup no longer sees all D columns of a cluster, the intra-normalisation moves to the finalize pass: this
// kernel writes the un-normalised residual sums U and per-column-slab partial square norms.
// Measured at cfg-2 (80 x 300 x 1024 x 256, tools/time_k2.py): 63 us.  Ablations of this form: DMA + barriers only
// 36 us, LDS fragment reads only 23 us, reads + MFMAs without DMA 40 us, loop without epilogue 48 us, epilogue only
// 19 us -- the loop is bound by the delivery path and by LDS read bandwidth (48 KB of fragment reads per 16 KB
// staged), not by HBM or the matrix pipe (15 us busy).
#include "lpm_common.h"

namespace lpm {

typedef __bf16 t3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned t3_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 t3_mfma(t3_u32x4 a, t3_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(t3_bf16x8, a), __builtin_bit_cast(t3_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float t3_bf(unsigned h) { return __uint_as_float(h << 16); }

constexpr int T3_NS = 3;                  // ring stages (48 KB: three workgroups per CU)
constexpr int T3_NT_DEFAULT = 0;          // see t3_nt()
constexpr int T3_STAGE = 16 * 1024;       // bytes per stage: 4 A tiles + 4 x tiles, 2 planes, 1 KB each
constexpr int T3_WS = 36;                 // epilogue per-wave tile row stride (floats): 144 B, 16-B aligned
constexpr int T3_EPI = 8 * 32 * T3_WS * 4;  // epilogue bytes (8 waves x [32 d][32 k + pad]), overlays the ring
constexpr int T3_SMX_RAW = T3_NS * T3_STAGE + (4 * 128 + 128 + 16) * 4;     // SMX: raw-logit ring [NS][8 KB]
constexpr int T3_SMX_LDS = T3_SMX_RAW + T3_NS * 8192;

// FUSED (lpm_vlad_aggregate_fused_fwd): the finalize pass moves into this kernel.  The K/128 x D/128 workgroups of a clip
// publish their partial column square norms (write-through stores), count themselves in on a per-clip arrival counter, wait --
// bounded -- until the clip is complete, read every partial norm of the clip back (L1-bypassing loads), form 1/n_k and the
// clip's 1/sqrt(g) themselves and store their tile of the NORMALISED descriptor straight from the accumulators, d-major or
// k-major.  The un-normalised sums never make a round trip through HBM (they are still written, once, when the backward will
// read them: store_u).  The workgroups of a clip are consecutive in dispatch order on one XCD, so they are co-resident in
// practice and the wait is a few microseconds of skew; if a workgroup ever times out it leaves its U tile in `nrm`, raises
// fail[clip], and the caller's follow-up launch (vlad_finalize2 restricted to flagged clips) finishes those clips: the result
// never depends on dispatch order, only the speed does (cdna guide 6, Guideline 16).
struct T3Fused {
    float* out;                 // [B, D*K] or [B, K, D]
    int kmajor, store_u, debug_fallback;     // debug_fallback (tests): the first column slab of every clip acts as if its wait had timed out
    int raw_kmajor;             // lpm_vlad_aggregate_raw_kmajor_fwd: no wait at all -- `out` receives the UN-normalised sums k-major [B, K, D]
                                // (the normalisation becomes a [B, K] row scale the descriptor's consumers apply: lpm_vlad_row_scales)
    float* colsq;               // [B, K] (outputs for the backward, written by the first workgroup of a clip)
    float* csq;
    float* gsq;                 // [B]
    unsigned* arrive;           // [B], zero at launch
    unsigned* fail;             // [B * K/128 * D/128] one flag per workgroup tile, zero at launch
};

// SMX (lpm_vlad_aggregate_raw_kmajor_smx_fwd): the softmax moves INTO this kernel -- frame_level_models.py:2798-2822 as one
// launch (+ the row statistics below and the [B, K] row scales).  No assignment tiles exist: per 16-frame step the workgroup's eight
// waves read the 16 x 128 logits of its cluster slab (the same 8 KB an A-tile stage was), apply the cluster_bn affine, exp(z - m_t)
// / sum_t from the per-frame statistics, split into bf16 hi / lo and write the four A fragment tiles of the NEXT step straight into
// its ring stage; the frame tiles keep arriving by LDS-DMA.
struct T3Softmax {
    const float* logits;        // [B*T, K] fp32 (K1's output)
    const float* scale;         // [K] cluster_bn folded (null: 1)
    const float* shift;         // [K] (null: 0; the bias when there is no batch norm)
    const float* stats;         // [B][Tpad][2]: row maximum of z = logits * scale + shift, 1 / sum exp(z - max); (0, 0) for t >= T
    int Tpad;                   // 16 * steps
};

// row statistics of the softmax, thread layout and order of operations of assign_tiles2_kernel (vlad_tiles.hip): a lane owns VPL
// consecutive clusters, a wave four frames, a workgroup one 16-frame step -- so that exp(z - m) * inv here and there are the same bits
template <int VPL>
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ logits, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int T, int S, float* __restrict__ stats) {
    constexpr int K = 64 * VPL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / S, s = blockIdx.x % S;
    const int c0 = lane * VPL;
    float sc[VPL], sh[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        sc[j] = scale ? scale[c0 + j] : 1.f;
        sh[j] = shift ? shift[c0 + j] : 0.f;
    }
    float v[4][VPL];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int t = 16 * s + wave * 4 + rr;
        if (t < T) {                                         // wave-uniform
            const float* p = logits + ((int64_t)b * T + t) * K + c0;
            if (VPL % 4 == 0) {
#pragma unroll
                for (int j = 0; j < VPL; j += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(p + j);
                    v[rr][j] = q.x; v[rr][j + 1] = q.y; v[rr][j + 2] = q.z; v[rr][j + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < VPL; ++j) v[rr][j] = p[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < VPL; ++j) v[rr][j] = 0.f;
        }
    }
    float m[4], sum[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        m[rr] = -INFINITY;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            v[rr][j] = fmaf(v[rr][j], sc[j], sh[j]);
            m[rr] = fmaxf(m[rr], v[rr][j]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) m[rr] = wave_max_dpp(m[rr]);          // (the reductions of assign_tiles2_kernel, in its order)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        sum[rr] = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) sum[rr] += __expf(v[rr][j] - m[rr]);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sum[rr] = wave_sum_dpp(sum[rr]);
    if (lane < 4) {
        const int t = 16 * s + wave * 4 + lane;
        float mm = 0.f, inv = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            if (rr == lane && t < T) { mm = m[rr]; inv = 1.f / sum[rr]; }
        *reinterpret_cast<float2*>(stats + ((int64_t)b * S * 16 + t) * 2) = make_float2(mm, inv);
    }
}

typedef __bf16 t3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float t3_f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed (hi, lo) bf16 pairs, round-to-nearest-even both (v_cvt_pk_bf16_f32; the bits of vlad_tiles.hip's split8)
__device__ __forceinline__ void t3_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const t3_f32x2 v = {a, b};
    const t3_bf16x2 h = __builtin_convertvector(v, t3_bf16x2);
    const t3_f32x2 hf = __builtin_convertvector(h, t3_f32x2);
    const t3_bf16x2 l = __builtin_convertvector(v - hf, t3_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// PL = 1 (bf16 storage): AT / XT are plain bf16 tiles, ONE plane per (tile, step), S (even) steps per clip; a ring stage then
// carries two consecutive frame steps where the split form carries the two planes of one step, and a product is one MFMA.
// (Round 5: this form runs the video stream of BASELINE configs[4] in 98 us = 0.32 of the HBM peak -- the 52 us quoted in rounds 3
// and 4 averaged the audio stream's 10 us launches in; the timing sites below now take video launches only.  A 256 x 256 form --
// one workgroup per CU, assignment tiles 4 x and frame tiles 2 x instead of 8 x and 4 x, six fragment reads per eight MFMAs -- was
// built twice and passed every bf16-storage test both times: as 1 024 workgroups (four rounds, every prologue and epilogue exposed)
// 124 us; as 256 persistent workgroups walking four tiles each with the DMA ring running across the tile boundaries 103 us (after
// two compiler traps: per-stage 64-bit lane pointers get s_waitcnt vmcnt(0) in front of the reuse of their registers -- buffer loads
// with scalar offsets instead --, and 64 hoisted epilogue pointers spill, a scratch reload behind every DMA issue).  It moves half the
// bytes into LDS but with one workgroup per CU nothing shares them in TIME: a clip's re-reads are 25 us apart, miss the XCD's L2 and
// come back from the fabric at ~3 TB/s, where this form's neighbours (same clip, same cluster slab, consecutive on one XCD, three
// workgroups per CU) ask for the same tile within microseconds.  Removed again; what helps is the clip-wide form of vlad_clip.hip.)
template <bool FUSED, int PL, bool SMX = false>
__global__ __launch_bounds__(512, SMX ? 4 : 6) void vlad_aggregate_tiles3_kernel(
    const uint4* __restrict__ at, const uint4* __restrict__ xt, const float* __restrict__ centres, int T, int D, int K,
    int S, int KT, int resflags, float* __restrict__ nrm, float* __restrict__ asum, float* __restrict__ colsq_part, const T3Fused fz,
    const T3Softmax sm) {
    static_assert(!SMX || PL == 2, "the in-kernel softmax writes split-bf16 fragments");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object (guide 5, trap (a))
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int kw = wave >> 2, dw = wave & 3;
    const int DT = D >> 5, P = D >> 7, KB = K >> 7;      // column tiles, column slabs, cluster slabs
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / (KB * P), rem = lid % (KB * P);
    const int kb = rem / P, ds = rem % P;                 // consecutive workgroups: same clip, same cluster slab

    // DMA role of this wave: piece (tile = wave >> 1, plane = wave & 1) of the A block and of the x block
    const int ptile = wave >> 1, pplane = wave & 1;
    // split form: + s * 128 (A), + s * DT * 128 (x);  plain form: the piece is step 2 s + pplane: + s * 128, + s * DT * 128 as well
    const uint4* asrc = PL == 2 ? at + ((((int64_t)b * KT + kb * 4 + ptile) * S) * 2 + pplane) * 64 + lane
                                : at + (((int64_t)b * KT + kb * 4 + ptile) * S + pplane) * 64 + lane;
    const uint4* xsrc = PL == 2 ? xt + ((((int64_t)b * S) * DT + ds * 4 + ptile) * 2 + pplane) * 64 + lane
                                : xt + (((int64_t)b * S + pplane) * DT + ds * 4 + ptile) * 64 + lane;
    const int adst = (ptile * 2 + pplane) * 1024, xdst = 8192 + (ptile * 2 + pplane) * 1024;
    const int NST = PL == 2 ? S : S / 2;           // ring stages
    const int residual = resflags & 1;             // bit 0: subtract (sum_t a) * centres; bits 1, 2: LDS-DMA cache policy, see issue()
    const int k0s = kb * 128;                      // first cluster of this workgroup's slab

    auto issue = [&](int s) {                 // the pieces of step min(s, NST - 1) into stage s % NS
        unsigned char* st = smem + (s % T3_NS) * T3_STAGE;
        const int sc = SMX ? min(s, NST - 1) : s;
        // resflags bit 1 / bit 2 (LPM_T3_NT = 1 / 2 / 3): non-temporal policy for the assignment / frame pieces
        if (!SMX) {
            if (resflags & 2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + (int64_t)sc * 128),
                                                 (__attribute__((address_space(3))) void*)(st + adst), 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + (int64_t)sc * 128),
                                                 (__attribute__((address_space(3))) void*)(st + adst), 16, 0, 0);
        }
        if (resflags & 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + (int64_t)sc * DT * 128),
                                             (__attribute__((address_space(3))) void*)(st + xdst), 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + (int64_t)sc * DT * 128),
                                             (__attribute__((address_space(3))) void*)(st + xdst), 16, 0, 0);
    };

    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float asum_w = 0.f;     // partial assignment sum of cluster row l31 of tile (kw, dw >> 1)

    // one step's MFMAs + assignment sums from the fragments in stage s % NS
    auto compute = [&](int s) {
        const unsigned char* st = smem + (s % T3_NS) * T3_STAGE;
        const t3_u32x4* af = reinterpret_cast<const t3_u32x4*>(st) + lane;                 // A tile c, plane p: + (c*2+p)*64
        const t3_u32x4* xf = reinterpret_cast<const t3_u32x4*>(st + 8192) + lane;
        const t3_u32x4 xh = xf[(dw * 2 + 0) * 64], xl = xf[(dw * 2 + 1) * 64];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const t3_u32x4 ah = af[((kw * 2 + c) * 2 + 0) * 64], al = af[((kw * 2 + c) * 2 + 1) * 64];
            if (PL == 2) {
                acc[c] = t3_mfma(ah, xh, acc[c]);
                acc[c] = t3_mfma(ah, xl, acc[c]);
                acc[c] = t3_mfma(al, xh, acc[c]);
            } else {                // (ah, xh): first frame step of the stage, (al, xl): second
                acc[c] = t3_mfma(ah, xh, acc[c]);
                acc[c] = t3_mfma(al, xl, acc[c]);
            }
            // assignment sums: the four column-tile waves of a cluster pair split the work (tile c = dw >> 1, frame pairs
            // 2 * (dw & 1) .. + 1 of the fragment); the epilogue adds their partial sums.  All of it on every wave was the
            // largest VALU consumer of the loop (64 ops per wave-step next to six MFMAs).
            if (c == (dw >> 1)) {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const unsigned h = (dw & 1) ? ah[2 + q2] : ah[q2], l = (dw & 1) ? al[2 + q2] : al[q2];
                    asum_w += (t3_bf(h & 0xffffu) + t3_bf(l & 0xffffu)) + (t3_bf(h >> 16) + t3_bf(l >> 16));
                }
            }
        }
    };

    if constexpr (!SMX) {
#pragma unroll
        for (int s = 0; s < T3_NS - 1; ++s)
            if (s < NST) issue(s);

        for (int s = 0; s < NST; ++s) {
            // this wave's two pieces of step s have landed when at most 2 * (younger steps in flight) remain
            const int behind = min(T3_NS - 2, NST - 1 - s);
            if (behind >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // everyone's pieces of step s are in LDS; stage (s-1) % NS is free
            asm volatile("" ::: "memory");
            if (s + T3_NS - 1 < NST) issue(s + T3_NS - 1);
            compute(s);
        }
    } else {
        // ---- in-kernel softmax.  Two more LDS-DMA streams ride next to the frame tiles (behind the reduction scratch: T3_SMX_RAW):
        //   Lg(j): the 16 x 128 raw logits of step j, eight 1 KB pieces -- wave w brings frames w and w + 8 (one 512-byte row each).
        // The clip's row statistics (max, 1 / sum) sit in LDS from the start.
        // During step s every wave turns its quarter of A tile tc = w >> 1 of step s + 1 -- lane (half, l31): cluster 32 tc + l31,
        // frames 8 half + 4 fq + {0..3}, fq = w & 1 -- from raw logits into split-bf16 fragment bytes: affine, exp(z - m) / sum,
        // hi / lo, two 8-byte LDS stores into stage (s + 1) % NS.  Every step issues the SAME operations -- X(s + 2), Lg(s + 3)
        // -- sources clamped to the last step past the end -- so the in-order vmcnt at the top of a step is a constant: the youngest
        // step's two operations stay in flight.
        const int tc = wave >> 1, fq = wave & 1;
        unsigned char* rawb = smem + T3_SMX_RAW;                       // [NS][8 pieces][1 KB]
        const float* lsrc = sm.logits + (int64_t)b * T * K + k0s + (lane & 31) * 4;      // + frame * K
        const int lfr = wave + 8 * (lane >> 5);                        // the frame of the step this lane's 16 bytes belong to
        // the clip's row statistics (max, 1 / sum per frame: 8 bytes x Tpad) are copied into LDS once, BEFORE the first DMA is issued:
        // ordinary loads inside the loop would share the in-order vector-memory counter with the ring
        float* stl = reinterpret_cast<float*>(rawb + T3_NS * 8192);
        for (int i = tid; i < sm.Tpad * 2; i += 512) stl[i] = sm.stats[(int64_t)b * sm.Tpad * 2 + i];
        const unsigned stl_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)stl;      // LDS byte addresses
        const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const float csc = sm.scale ? sm.scale[k0s + tc * 32 + l31] : 1.f;
        const float csh = sm.shift ? sm.shift[k0s + tc * 32 + l31] : 0.f;
        auto issue_l = [&](int j) {
            const int jc = min(j, NST - 1);
            const int t = min(16 * jc + lfr, T - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + (int64_t)t * K),
                                             (__attribute__((address_space(3))) void*)(rawb + (j % T3_NS) * 8192 + wave * 1024), 16, 0, 0);
        };
        // raw logits + statistics of step j -> this wave's quarter of A tile tc of stage j % NS, in two halves: the LDS reads go out
        // BEFORE the step's MFMAs (their latency runs under the matrix pipe), arithmetic and stores come after.  All LDS traffic of
        // this path is inline assembly: hipcc puts an s_waitcnt vmcnt(0) in front of an ordinary LDS read or store of these regions
        // while LDS-DMA is in flight -- draining the two ring steps -- although none of it touches what the DMA writes.  (LDS
        // returns in order, so these extra reads only make the compiler's own lgkmcnt waits stricter, never wrong.)
        struct CReg { f32x4 s0, s1; float l[4]; };
        const unsigned raw_lds = smem_lds + T3_SMX_RAW + (half * 128 + tc * 32 + l31) * 4 + fq * 4096;      // + stage * 8192; frame q: + q * 1024
        auto convert_read = [&](int j, CReg& r) {
            const unsigned sa = stl_lds + (16 * j + 8 * half + 4 * fq) * 8;
            const unsigned ra = raw_lds + (j % T3_NS) * 8192;
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\t"
                         "ds_read_b32 %2, %7\n\tds_read_b32 %3, %7 offset:1024\n\tds_read_b32 %4, %7 offset:2048\n\tds_read_b32 %5, %7 offset:3072"
                         : "=&v"(r.s0), "=&v"(r.s1), "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3])
                         : "v"(sa), "v"(ra)
                         : "memory");
        };
        auto convert_write = [&](int j, CReg& r) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.s0), "+v"(r.s1), "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3])::"memory");
            const float mx[4] = {r.s0[0], r.s0[2], r.s1[0], r.s1[2]}, iv[4] = {r.s0[1], r.s0[3], r.s1[1], r.s1[3]};
            float av[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                av[q] = __expf(fmaf(r.l[q], csc, csh) - mx[q]) * iv[q];
                // the product is ROUNDED before the hi / lo split, as in lpm_assign_tiles: left to itself (fp-contract=fast) hipcc turns
                // a - hi into fma(e, inv, -hi), i.e. splits the unrounded product, and the two forms stop agreeing bit for bit
                asm volatile("" : "+v"(av[q]));
            }
            unsigned h01, l01, h23, l23;
            t3_split2(av[0], av[1], h01, l01);
            t3_split2(av[2], av[3], h23, l23);
            typedef unsigned t3_u32x2 __attribute__((ext_vector_type(2)));
            const t3_u32x2 hv = {h01, h23}, lv2 = {l01, l23};
            const unsigned da = smem_lds + (j % T3_NS) * T3_STAGE + tc * 2048 + lane * 16 + fq * 8;
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:1024" ::"v"(da), "v"(hv), "v"(lv2) : "memory");
        };
        CReg cr;
        issue_l(0);
        issue(0);
        issue_l(1);
        issue(1);
        issue_l(2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // Lg(0) has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        convert_read(0, cr);
        convert_write(0, cr);
        for (int s = 0; s < NST; ++s) {
            // oldest first: X(s), Lg(s+1) | X(s+1), Lg(s+2)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's quarter of A(s) (written during step s - 1) is in LDS
            __builtin_amdgcn_s_barrier();              // X(s), A(s), Lg(s+1) are in LDS for everyone; step s - 1 is done with
            asm volatile("" ::: "memory");
            issue(s + 2);
            issue_l(s + 3);
            if (s + 1 < NST) convert_read(s + 1, cr);
            compute(s);
            if (s + 1 < NST) convert_write(s + 1, cr);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the redundant tail operations
    }
    __syncthreads();     // no DMA in flight any more: the ring is reused by the epilogue

    // ---- epilogue: transpose through LDS, then residual + partial column square norms ride on the coalesced d-major store.
    // Two passes (one per 32-cluster tile of the wave) through a wave-private [32 d][32 k] LDS tile that overlays the ring,
    // so the workgroup's LDS stays at 3 ring stages and THREE workgroups share a CU.  Each lane of the store pass owns 4
    // fixed cluster columns over 4 rows: the centres are read from global in exactly the store's pattern (issued before the
    // transposes, no LDS staging), the residual and the squares are applied to the values as they leave LDS, and the column
    // norms need three cross-lane steps instead of a 32-lane butterfly per accumulator register.
    float* wl = reinterpret_cast<float*>(smem) + wave * (32 * T3_WS);
    float* red = reinterpret_cast<float*>(smem + T3_NS * T3_STAGE);   // [4 dw][128 k]
    float* ssum = red + 4 * 128;                                      // [128 k]
    const int k0 = kb * 128, d0 = ds * 128 + dw * 32;
    const int srow = lane >> 3, c4 = (lane & 7) * 4;            // store pass: row it*8 + srow, columns c4..c4+3 of the tile's 32
    float4 cw[2][4];
    if (residual && !FUSED) {          // (the fused form holds more live state: it fetches the centres tile by tile below)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                cw[c][it] = *reinterpret_cast<const float4*>(centres + (int64_t)(d0 + it * 8 + srow) * K + k0 + kw * 64 + c * 32 + c4);
    }
    asum_w += __shfl_xor(asum_w, 32, 64);
    if (half == 0) red[(dw & 1) * 128 + (kw * 2 + (dw >> 1)) * 32 + l31] = asum_w;      // two partial sums per cluster
    __syncthreads();
    if (tid < 128) ssum[tid] = red[tid] + red[128 + tid];
    __syncthreads();
    float* ob = nrm + ((int64_t)b * D + d0) * K + k0 + kw * 64;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (FUSED && residual) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                cw[c][it] = *reinterpret_cast<const float4*>(centres + (int64_t)(d0 + it * 8 + srow) * K + k0 + kw * 64 + c * 32 + c4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(wl + l31 * T3_WS + 8 * q + 4 * half) =
                make_float4(acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave-private tile: program order within the wave is enough
        __builtin_amdgcn_wave_barrier();
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (residual) s4 = *reinterpret_cast<const float4*>(ssum + (kw * 2 + c) * 32 + c4);
        float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + srow;
            float4 u = *reinterpret_cast<const float4*>(wl + row * T3_WS + c4);
            if (residual) {
                u.x -= s4.x * cw[c][it].x; u.y -= s4.y * cw[c][it].y; u.z -= s4.z * cw[c][it].z; u.w -= s4.w * cw[c][it].w;
            }
            sq.x = fmaf(u.x, u.x, sq.x); sq.y = fmaf(u.y, u.y, sq.y); sq.z = fmaf(u.z, u.z, sq.z); sq.w = fmaf(u.w, u.w, sq.w);
            if (PL == 1) {        // bf16 storage: the un-normalised sums are kept as bf16 as well (the norms below come from the fp32 values)
                auto rne = [](float f) { unsigned w = __float_as_uint(f); w += 0x7fffu + ((w >> 16) & 1u); return w >> 16; };
                unsigned short* obh = reinterpret_cast<unsigned short*>(nrm) + ((int64_t)b * D + d0) * K + k0 + kw * 64;
                *reinterpret_cast<uint2*>(obh + (int64_t)row * K + c * 32 + c4) = make_uint2(rne(u.x) | (rne(u.y) << 16), rne(u.z) | (rne(u.w) << 16));
            } else if (!FUSED || fz.store_u) {
                *reinterpret_cast<float4*>(ob + (int64_t)row * K + c * 32 + c4) = u;
            }
            if (FUSED) *reinterpret_cast<float4*>(wl + row * T3_WS + c4) = u;      // the residual goes back into the accumulators
        }
        if (FUSED) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(wl + l31 * T3_WS + 8 * q + 4 * half);
                acc[c][4 * q] = t.x; acc[c][4 * q + 1] = t.y; acc[c][4 * q + 2] = t.z; acc[c][4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            sq.x += __shfl_xor(sq.x, m, 64); sq.y += __shfl_xor(sq.y, m, 64); sq.z += __shfl_xor(sq.z, m, 64); sq.w += __shfl_xor(sq.w, m, 64);
        }
        if (lane < 8) *reinterpret_cast<float4*>(red + dw * 128 + (kw * 2 + c) * 32 + c4) = sq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                            // all reads of the tile done before pass 1 overwrites it
    }
    __syncthreads();
    if (!FUSED) {
        if (tid < 128) {
            colsq_part[((int64_t)b * P + ds) * K + k0 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            if (ds == 0) asum[(int64_t)b * K + k0 + tid] = ssum[tid];
        }
        return;
    }
    if (fz.raw_kmajor) {
        // the residual sums as they are, k-major, straight from the accumulators (acc[c][r]: column d = l31, cluster 8 (r >> 2) + 4 half
        // + (r & 3)): for a fixed register the 32 lanes of a half-wave hold 32 consecutive d of one cluster row -- 128-byte segments
        if (tid < 128) {
            colsq_part[((int64_t)b * P + ds) * K + k0 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            if (ds == 0) asum[(int64_t)b * K + k0 + tid] = ssum[tid];
        }
        // through the wave-private LDS tile once more, this time as [32 k][32 d]: the store pass then writes float4 along d
        // (8 lanes = one 128-byte row piece; 8 stores per lane instead of 32 dword stores)
        float* okb = fz.out + ((int64_t)b * K + k0 + kw * 64) * D + d0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) wl[(8 * q + 4 * half + j) * T3_WS + l31] = acc[c][4 * q + j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int kr = it * 8 + srow;                       // cluster row of the tile; c4 = 4 consecutive d
                *reinterpret_cast<float4*>(okb + (int64_t)(c * 32 + kr) * D + c4) = *reinterpret_cast<const float4*>(wl + kr * T3_WS + c4);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // ---- publish this workgroup's 128 partial norms write-through, arrive, wait for the clip (Guideline 16, form R1 with sc1 loads)
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) unsigned gu32;
    if (tid < 128) {
        const float pn = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
        __hip_atomic_store((gfloat*)(colsq_part + ((int64_t)b * P + ds) * K + k0 + tid), pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ds == 0) asum[(int64_t)b * K + k0 + tid] = ssum[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
    __syncthreads();
    int* okp = reinterpret_cast<int*>(ssum + 128);
    if (tid == 0) {
        gu32* arr = (gu32*)(fz.arrive + b);
        __hip_atomic_fetch_add(arr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned need = (unsigned)(KB * P);
        const long long t0 = wall_clock64();                   // 100 MHz
        int ok = 1;
        while (__hip_atomic_load(arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > 200000) { ok = 0; break; }     // 2 ms: dispatch did not keep the clip together
        }
        if (fz.debug_fallback && ds == 0) ok = 0;
        *okp = ok;
    }
    __syncthreads();
    if (*okp == 0) {
        // fallback: leave the un-normalised tile for the caller's follow-up finalize of this clip
        if (!fz.store_u) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(wl + l31 * T3_WS + 8 * q + 4 * half) =
                        make_float4(acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + srow;
                    *reinterpret_cast<float4*>(ob + (int64_t)row * K + c * 32 + c4) = *reinterpret_cast<const float4*>(wl + row * T3_WS + c4);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (tid == 0) __hip_atomic_store((gu32*)(fz.fail + lid), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- every partial norm of the clip -> 1/n_k for all K clusters and the clip's 1/sqrt(g)   (fixed summation order)
    float* gl = reinterpret_cast<float*>(smem);                // [P][K], overlays the wave tiles (their content lives in acc again)
    for (int i = tid; i < P * K; i += 512)
        gl[i] = __hip_atomic_load((gfloat*)(colsq_part + (int64_t)b * P * K + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    float* invn = red;                                         // [K <= 512]
    float cg = 0.f;
    if (tid < K) {
        float n = 0.f;
        for (int p = 0; p < P; ++p) n += gl[p * K + tid];
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        cg = n * iv * iv;
        invn[tid] = iv;
        if (kb == 0 && ds == 0) {
            fz.colsq[(int64_t)b * K + tid] = n;
            fz.csq[(int64_t)b * K + tid] = cg;
        }
    }
    cg = wave_sum(cg);
    float* wg8 = ssum;                                         // [8] per-wave partial sums of g (ssum is dead)
    if (lane == 0) wg8[wave] = cg;
    __syncthreads();
    const float tot = ((wg8[0] + wg8[1]) + (wg8[2] + wg8[3])) + ((wg8[4] + wg8[5]) + (wg8[6] + wg8[7]));
    const float ig = rsqrtf(fmaxf(tot, kL2Eps));
    if (kb == 0 && ds == 0 && tid == 0) fz.gsq[b] = tot;
    // ---- the normalised tile, straight from the accumulators (acc[c][r]: column d = l31, cluster 8 (r >> 2) + 4 half + (r & 3))
    if (fz.kmajor) {
        // [B, K, D]: for a fixed register the 32 lanes of a half-wave hold 32 consecutive d of one cluster row: 128-byte segments
        float* okb = fz.out + ((int64_t)b * K + k0 + kw * 64) * D + d0 + l31;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kl = c * 32 + 8 * q + 4 * half;
                const float4 iv = *reinterpret_cast<const float4*>(invn + k0 + kw * 64 + kl);
                okb[(int64_t)(kl + 0) * D] = acc[c][4 * q + 0] * (iv.x * ig);
                okb[(int64_t)(kl + 1) * D] = acc[c][4 * q + 1] * (iv.y * ig);
                okb[(int64_t)(kl + 2) * D] = acc[c][4 * q + 2] * (iv.z * ig);
                okb[(int64_t)(kl + 3) * D] = acc[c][4 * q + 3] * (iv.w * ig);
            }
    } else {
        // [B, D*K] d-major: through the wave-private LDS tile like the U store, scaled on the way in
        float* od = fz.out + ((int64_t)b * D + d0) * K + k0 + kw * 64;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 iv = *reinterpret_cast<const float4*>(invn + k0 + kw * 64 + c * 32 + 8 * q + 4 * half);
                *reinterpret_cast<float4*>(wl + l31 * T3_WS + 8 * q + 4 * half) =
                    make_float4(acc[c][4 * q] * (iv.x * ig), acc[c][4 * q + 1] * (iv.y * ig), acc[c][4 * q + 2] * (iv.z * ig),
                                acc[c][4 * q + 3] * (iv.w * ig));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + srow;
                *reinterpret_cast<float4*>(od + (int64_t)row * K + c * 32 + c4) = *reinterpret_cast<const float4*>(wl + row * T3_WS + c4);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Follow-up of the fused kernel: one workgroup per (clip, 128 clusters, 128 columns) tile; tiles whose workgroup completed (flag
// zero: all of them in practice) return at once, a tile whose workgroup gave up waiting is finished here from the U it left in nrm.
__global__ __launch_bounds__(256) void vlad_fused_fixup_kernel(const float* __restrict__ nrm, const float* __restrict__ colsq_part,
                                                               const unsigned* __restrict__ fail, int D, int K, T3Fused fz) {
    const int lid = blockIdx.x;
    if (fail[lid] == 0) return;
    const int P = D >> 7, KB = K >> 7;
    const int b = lid / (KB * P), rem = lid % (KB * P), kb = rem / P, ds = rem % P;
    __shared__ float invn[512];
    __shared__ float wg[4];
    const int tid = threadIdx.x;
    float g = 0.f;
    for (int k = tid; k < K; k += 256) {
        // (the P partial norms are requested together, eight at a time, and added in slab order: the kernel is one dependent chain of
        // load -> reduce -> rsqrt -> store per clip, 9 us at P = 8 when every load waited for the previous add)
        float n = 0.f;
        const float* src = colsq_part + (int64_t)b * P * K + k;
        for (int p0 = 0; p0 < P; p0 += 8) {
            float pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = (p0 + u < P) ? src[(int64_t)(p0 + u) * K] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) n += pv[u];
        }
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        const float c = n * iv * iv;
        invn[k] = iv;
        g += c;
        if (kb == 0 && ds == 0) {
            fz.colsq[(int64_t)b * K + k] = n;
            fz.csq[(int64_t)b * K + k] = c;
        }
    }
    g = wave_sum(g);
    if ((tid & 63) == 0) wg[tid >> 6] = g;
    __syncthreads();
    const float tot = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    const float ig = rsqrtf(fmaxf(tot, kL2Eps));
    if (kb == 0 && ds == 0 && tid == 0) fz.gsq[b] = tot;
    for (int i = tid; i < 128 * 128; i += 256) {
        const int dl = i >> 7, kl = i & 127;
        const int d = ds * 128 + dl, k = kb * 128 + kl;
        const float v = nrm[((int64_t)b * D + d) * K + k] * (invn[k] * ig);
        if (fz.kmajor) fz.out[((int64_t)b * K + k) * D + d] = v;
        else fz.out[((int64_t)b * D + d) * K + k] = v;
    }
}

// finalize for the un-normalised form: per clip n_k = sum_p colsq_part, inv_n = rsqrt(max(n,eps)), c_k = n inv_n^2,
// g = sum_k c_k;  nrm <- U * inv_n (in place, d-major: what the backward reads);  out = nrm * rsqrt(max(g,eps)) laid out
// d-major [B, D*K] or k-major [B,K,D].   grid (D/32, B).
template <bool KMAJOR>
__global__ __launch_bounds__(256) void vlad_finalize2_kernel(float* __restrict__ nrm, const float* __restrict__ colsq_part,
                                                             int P, int D, int K, float* __restrict__ out,
                                                             float* __restrict__ colsq, float* __restrict__ csq,
                                                             float* __restrict__ gsq, int keep_u, int out_bf16, int nrm_bf16,
                                                             int64_t out_bs) {
    // keep_u (LPM_VLAD_NRM_RAW): nrm is left as the un-normalised sums U (the tile backward rebuilds N = U * inv_n itself)
    // out_bf16 (d-major only): `out` is bf16 storage;  out_bs: distance between the clips' descriptors in `out`, in elements
    extern __shared__ float fs[];            // [K] inv_n, then [32][33] transpose tile, [4] partial sums
    float* invn = fs;
    float* tile = fs + K;
    float* wg = tile + 32 * 33;
    const int b = blockIdx.y, d0 = blockIdx.x * 32, tid = threadIdx.x;   // consecutive workgroups: consecutive 32-row chunks of one clip
    float g = 0.f;
    for (int k = tid; k < K; k += 256) {
        float n = 0.f;
        for (int p = 0; p < P; ++p) n += colsq_part[((int64_t)b * P + p) * K + k];
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        const float c = n * iv * iv;
        invn[k] = iv;
        g += c;
        if (blockIdx.x == 0) {
            colsq[(int64_t)b * K + k] = n;
            csq[(int64_t)b * K + k] = c;
        }
    }
    g = wave_sum(g);
    if ((tid & 63) == 0) wg[tid >> 6] = g;
    __syncthreads();
    const float tot = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    const float ig = rsqrtf(fmaxf(tot, kL2Eps));
    if (blockIdx.x == 0 && tid == 0) gsq[b] = tot;
    float* src = nrm + ((int64_t)b * D + d0) * K;
    if (!KMAJOR) {
        float* dst = out + (int64_t)b * out_bs + (int64_t)d0 * K;
        const int n4 = 32 * K / 4, K4 = K / 4;
        for (int i = tid; i < n4; i += 256) {
            const int k = (i % K4) * 4;
            float4 v;
            if (nrm_bf16) {               // (bf16 storage: U as bf16, never written back)
                const uint2 q = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(nrm) + ((int64_t)b * D + d0) * K)[i];
                v = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                                __uint_as_float(q.y & 0xffff0000u));
            } else {
                v = reinterpret_cast<const float4*>(src)[i];
            }
            v.x *= invn[k]; v.y *= invn[k + 1]; v.z *= invn[k + 2]; v.w *= invn[k + 3];
            if (!keep_u && !nrm_bf16) reinterpret_cast<float4*>(src)[i] = v;
            v.x *= ig; v.y *= ig; v.z *= ig; v.w *= ig;
            if (out_bf16) {
                auto rne = [](float f) { unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; };
                reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + (int64_t)b * out_bs + (int64_t)d0 * K)[i] =
                    make_uint2(rne(v.x) | (rne(v.y) << 16), rne(v.z) | (rne(v.w) << 16));
            } else {
                reinterpret_cast<float4*>(dst)[i] = v;
            }
        }
    } else {
        float* dst = out + (int64_t)b * out_bs;
        const int tx = tid & 31, ty = tid >> 5;
        for (int kb = 0; kb < K; kb += 32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dl = ty + 8 * i, k = kb + tx;
                float v = 0.f;
                if (k < K) {
                    v = src[(int64_t)dl * K + k] * invn[k];
                    if (!keep_u) src[(int64_t)dl * K + k] = v;
                }
                tile[dl * 33 + tx] = v * ig;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kl = ty + 8 * i, k = kb + kl;
                if (k < K) dst[(int64_t)k * D + d0 + tx] = tile[tx * 33 + kl];
            }
            __syncthreads();
        }
    }
}

// k-major finalize for K <= 512 (App. C5: the [B, K, D] layout NetVladV1's cluster encoder reads): the same arithmetic as
// vlad_finalize2_kernel<true>, but every global access is 16 bytes wide.  R d-rows x K values are scaled by 1/n_k, written
// back in place (d-major, what the backward reads) and staged, times 1/sqrt(g), in an LDS tile whose row stride K + 2 makes
// the transposed read conflict-free (bank = 8 dq + 2 j + k over the 64 lanes of a wave); the tile leaves as float4 pieces
// along d: 4 R contiguous bytes per cluster row.   grid (D/R, B).
template <int R>
__global__ __launch_bounds__(256) void vlad_finalize2_kmajor4_kernel(float* __restrict__ nrm, const float* __restrict__ colsq_part,
                                                                     int P, int D, int K, float* __restrict__ out,
                                                                     float* __restrict__ colsq, float* __restrict__ csq,
                                                                     float* __restrict__ gsq, int keep_u) {
    extern __shared__ float fs[];            // [K] inv_n, [4] partial sums, [R][K + 2] tile
    float* invn = fs;
    float* wg = fs + K;
    float* tile = fs + K + 4;                // (K % 4 == 0: 16-byte aligned)
    const int TS = K + 2;
    const int b = blockIdx.y, d0 = blockIdx.x * R, tid = threadIdx.x;    // consecutive workgroups: consecutive R-row chunks of one clip
    float g = 0.f;
    for (int k = tid; k < K; k += 256) {
        float n = 0.f;
        for (int p = 0; p < P; ++p) n += colsq_part[((int64_t)b * P + p) * K + k];
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        const float c = n * iv * iv;
        invn[k] = iv;
        g += c;
        if (blockIdx.x == 0) {
            colsq[(int64_t)b * K + k] = n;
            csq[(int64_t)b * K + k] = c;
        }
    }
    g = wave_sum(g);
    if ((tid & 63) == 0) wg[tid >> 6] = g;
    __syncthreads();
    const float tot = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    const float ig = rsqrtf(fmaxf(tot, kL2Eps));
    if (blockIdx.x == 0 && tid == 0) gsq[b] = tot;
    float4* src4 = reinterpret_cast<float4*>(nrm + ((int64_t)b * D + d0) * K);
    const int K4 = K / 4;
    for (int i = tid; i < R * K4; i += 256) {
        const int row = i / K4, k = (i - row * K4) * 4;
        float4 v = src4[i];
        const float4 iv = *reinterpret_cast<const float4*>(invn + k);
        v.x *= iv.x; v.y *= iv.y; v.z *= iv.z; v.w *= iv.w;
        if (!keep_u) src4[i] = v;
        float2* t2 = reinterpret_cast<float2*>(tile + row * TS + k);        // 8-byte aligned: TS and k are even
        t2[0] = make_float2(v.x * ig, v.y * ig);
        t2[1] = make_float2(v.z * ig, v.w * ig);
    }
    __syncthreads();
    float* dst = out + (int64_t)b * K * D + d0;
    constexpr int Q = R / 4;
    for (int i = tid; i < K * Q; i += 256) {
        const int k = i / Q, dq = i - k * Q;
        const float* t = tile + (4 * dq) * TS + k;
        *reinterpret_cast<float4*>(dst + (int64_t)k * D + 4 * dq) = make_float4(t[0], t[TS], t[2 * TS], t[3 * TS]);
    }
}

}  // namespace lpm

namespace lpm {
// LDS-DMA cache policy of the aggregation kernel's two streams (bit 0: assignment tiles, bit 1: frame tiles non-temporal)
static int t3_nt() {
    static const int v = [] { const char* e = getenv("LPM_T3_NT"); return e ? (atoi(e) & 3) : T3_NT_DEFAULT; }();
    return v << 1;
}
}  // namespace lpm
extern "C" int lpm_vlad_tiles3_supported(int D, int K) { return (D % 128 == 0 && K % 128 == 0 && D >= 128 && K >= 128) ? 1 : 0; }

static int vlad_aggregate_tiles3_impl(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                      float* nrm, float* asum, float* colsq_part, int planes, lpm_stream_t stream);
extern "C" int lpm_vlad_aggregate_tiles3_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                             int flags, float* nrm, float* asum, float* colsq_part, lpm_stream_t stream) {
    return vlad_aggregate_tiles3_impl(at, xt, centres, B, T, D, K, flags, nrm, asum, colsq_part, 2, stream);
}
// bf16 storage (BASELINE cfg-5): at = lpm_assign_tiles_bf16, xt = lpm_frame_apply_tiles_bf16 / lpm_split_frames_bf16 (plain bf16
// tiles, 4 ceil(T / 64) steps per clip); one MFMA per product, fp32 accumulation; outputs as lpm_vlad_aggregate_tiles3_fwd.
extern "C" int lpm_vlad_aggregate_tiles3_fwd_bf16(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                                  int flags, float* nrm, float* asum, float* colsq_part, lpm_stream_t stream) {
    return vlad_aggregate_tiles3_impl(at, xt, centres, B, T, D, K, flags, nrm, asum, colsq_part, 1, stream);
}
static int vlad_aggregate_tiles3_impl(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                      float* nrm, float* asum, float* colsq_part, int planes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && nrm && asum && colsq_part, LPM_ERR_BADARG, "lpm_vlad_aggregate_tiles3_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_tiles3_fwd: RESIDUAL needs centres");
    LPM_REQUIRE(B > 0 && T > 0 && lpm_vlad_tiles3_supported(D, K), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_tiles3_fwd: need D %% 128 == 0 and K %% 128 == 0 (D=%d K=%d)", D, K);
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)nrm) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_tiles3_fwd: pointers must be 16-byte aligned");
    const int S = planes == 1 ? 4 * ((T + 63) / 64) : (T + 15) / 16, KT = K / 32;
    const size_t lds = (size_t)T3_NS * T3_STAGE + (4 * 128 + 128) * sizeof(float);
    static_assert(T3_EPI <= T3_NS * T3_STAGE, "the epilogue tiles overlay the DMA ring");
    auto kern = planes == 1 ? vlad_aggregate_tiles3_kernel<false, 1> : vlad_aggregate_tiles3_kernel<false, 2>;
    const T3Fused fz{};
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_tiles3_fwd: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid(B * (K / 128) * (D / 128));
    hipEvent_t e0, e1;
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))      // (the video stream's launches only: lpm_common.h)
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, (const uint4*)at, (const uint4*)xt, centres, T,
                              D, K, S, KT, residual | t3_nt(), nrm, asum, colsq_part, fz, T3Softmax{});
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, (const uint4*)at, (const uint4*)xt, centres, T, D, K, S, KT,
                           residual | t3_nt(), nrm, asum, colsq_part, fz, T3Softmax{});
    return check_launch("lpm_vlad_aggregate_tiles3_fwd");
}

// ---- K2 writing the un-normalised sums k-major + the row scales that normalise them -----------------------------------------------
namespace lpm {
// per clip: n_k = sum_p colsq_part, 1/n_k, c_k, g = sum_k c_k;  scale[b, k] = rsqrt(max(n_k, eps)) * rsqrt(max(g, eps))   (K <= 1024)
__global__ __launch_bounds__(256) void vlad_row_scales_kernel(const float* __restrict__ colsq_part, int P, int K, float* __restrict__ scale,
                                                              float* __restrict__ colsq, float* __restrict__ csq, float* __restrict__ gsq) {
    __shared__ float wg[4];
    __shared__ float invn[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    float g = 0.f;
    for (int k = tid; k < K; k += 256) {
        float n = 0.f;
        for (int p = 0; p < P; ++p) n += colsq_part[((int64_t)b * P + p) * K + k];
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        const float c = n * iv * iv;
        invn[k] = iv;
        g += c;
        colsq[(int64_t)b * K + k] = n;
        csq[(int64_t)b * K + k] = c;
    }
    g = wave_sum(g);
    if ((tid & 63) == 0) wg[tid >> 6] = g;
    __syncthreads();
    const float tot = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    const float ig = rsqrtf(fmaxf(tot, kL2Eps));
    if (tid == 0) gsq[b] = tot;
    for (int k = tid; k < K; k += 256) scale[(int64_t)b * K + k] = invn[k] * ig;
}
}  // namespace lpm

// K2 for a consumer that applies the normalisation itself (the NetVladV1 cluster encoders, App. C5: tokens = clusters): the
// aggregation kernel stores the UN-normalised residual sums k-major [B, K, D] -- once, straight from the accumulators -- and
// lpm_vlad_row_scales turns the partial norms into scale [B, K] = 1 / (n_k sqrt(g)) (+ colsq, csq, gsq for the backward), so that
// descriptor[b, k, :] = raw[b, k, :] * scale[b, k] (frame_level_models.py:2819-2822 as a per-row factor).  No finalize pass: the
// [B, D, K]-sized tensor is written once and never re-read by the pooling.  colsq_part: [B, D/128, K] floats.
extern "C" int lpm_vlad_aggregate_raw_kmajor_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                                 int flags, float* raw_kmajor, float* asum, float* colsq_part, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && raw_kmajor && asum && colsq_part, LPM_ERR_BADARG, "lpm_vlad_aggregate_raw_kmajor_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_raw_kmajor_fwd: RESIDUAL needs centres");
    LPM_REQUIRE(B > 0 && T > 0 && lpm_vlad_tiles3_supported(D, K), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_raw_kmajor_fwd: need D %% 128 == 0 and K %% 128 == 0 (D=%d K=%d)", D, K);
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)raw_kmajor) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_raw_kmajor_fwd: pointers must be 16-byte aligned");
    const int S = (T + 15) / 16, KT = K / 32;
    T3Fused fz{};
    fz.out = raw_kmajor; fz.raw_kmajor = 1;
    // LPM_K2_EXTRA_LDS (measurement): bytes of unused LDS added to the launch -- 27000 leaves two workgroups per CU instead of three
    static const int extra_lds = [] { const char* e = getenv("LPM_K2_EXTRA_LDS"); return e ? atoi(e) : 0; }();
    const size_t lds = (size_t)T3_NS * T3_STAGE + (4 * 128 + 128 + 16) * sizeof(float) + (size_t)extra_lds;
    auto kern = vlad_aggregate_tiles3_kernel<true, 2>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_raw_kmajor_fwd: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid(B * (K / 128) * (D / 128));
    hipEvent_t e0, e1;
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))      // (the video stream's launches only: lpm_common.h)
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, (const uint4*)at, (const uint4*)xt, centres, T, D, K,
                              S, KT, residual | t3_nt(), (float*)nullptr, asum, colsq_part, fz, T3Softmax{});
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, (const uint4*)at, (const uint4*)xt, centres, T, D, K, S, KT,
                           residual | t3_nt(), (float*)nullptr, asum, colsq_part, fz, T3Softmax{});
    return check_launch("lpm_vlad_aggregate_raw_kmajor_fwd");
}

// ... with the softmax inside the aggregation kernel (T3Softmax): logits [B*T, K] fp32 + the folded cluster_bn affine (scale / shift,
// either may be NULL) in, no assignment tiles.  stats: lpm_vlad_smx_stats_bytes(B, T) bytes of scratch (per-frame row maximum and
// 1 / row sum, written by a first small launch).  K in {128, 256, 512}, T >= 33.
extern "C" size_t lpm_vlad_smx_stats_bytes(int B, int T) { return (size_t)B * 16 * ((T + 15) / 16) * 2 * sizeof(float); }
extern "C" int lpm_vlad_smx_supported(int T, int D, int K) {
    return (lpm_vlad_tiles3_supported(D, K) && (K == 128 || K == 256 || K == 512) && (T + 15) / 16 >= 3 && T <= 4096) ? 1 : 0;
}
extern "C" int lpm_vlad_aggregate_raw_kmajor_smx_fwd(const float* logits, const float* scale, const float* shift, const void* xt,
                                                     const float* centres, int B, int T, int D, int K, int flags, float* raw_kmajor,
                                                     float* asum, float* colsq_part, float* stats, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(logits && xt && raw_kmajor && asum && colsq_part && stats, LPM_ERR_BADARG, "lpm_vlad_aggregate_raw_kmajor_smx_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_raw_kmajor_smx_fwd: RESIDUAL needs centres");
    LPM_REQUIRE((flags & LPM_VLAD_SOFTMAX) != 0, LPM_ERR_BADARG, "lpm_vlad_aggregate_raw_kmajor_smx_fwd: this entry IS the softmax form");
    LPM_REQUIRE(B > 0 && lpm_vlad_smx_supported(T, D, K), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_raw_kmajor_smx_fwd: need D %% 128 == 0, K in {128, 256, 512}, T >= 33 (T=%d D=%d K=%d)", T, D, K);
    LPM_REQUIRE((((uintptr_t)logits | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)raw_kmajor | (uintptr_t)stats) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_raw_kmajor_smx_fwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int S = (T + 15) / 16, KT = K / 32;
    hipEvent_t e0, e1;
    const bool timed_st = D >= 1024 && timing_request(LPM_TIMING_ASSIGN_TILES, &e0, &e1);
#define LPM_SMX_STATS(VPL)                                                                                                          \
    do {                                                                                                                            \
        if (timed_st) hipExtLaunchKernelGGL((softmax_stats_kernel<VPL>), dim3(B * S), dim3(256), 0, s, e0, e1, 0, logits, scale, shift, T, S, stats); \
        else hipLaunchKernelGGL((softmax_stats_kernel<VPL>), dim3(B * S), dim3(256), 0, s, logits, scale, shift, T, S, stats);       \
    } while (0)
    if (K == 128) LPM_SMX_STATS(2); else if (K == 256) LPM_SMX_STATS(4); else LPM_SMX_STATS(8);
#undef LPM_SMX_STATS
    T3Fused fz{};
    fz.out = raw_kmajor; fz.raw_kmajor = 1;
    T3Softmax sm{logits, scale, shift, stats, 16 * S};
    const size_t lds = (size_t)T3_SMX_LDS + (size_t)16 * S * 2 * sizeof(float);
    LPM_REQUIRE(lds <= 160 * 1024, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_vlad_aggregate_raw_kmajor_smx_fwd: T = %d needs %zu bytes of LDS", T, lds);
    auto kern = vlad_aggregate_tiles3_kernel<true, 2, true>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_raw_kmajor_smx_fwd: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid(B * (K / 128) * (D / 128));
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))      // (the video stream's launches only: lpm_common.h)
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, s, e0, e1, 0, (const uint4*)nullptr, (const uint4*)xt, centres, T, D, K, S, KT,
                              residual | t3_nt(), (float*)nullptr, asum, colsq_part, fz, sm);
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const uint4*)nullptr, (const uint4*)xt, centres, T, D, K, S, KT, residual | t3_nt(),
                           (float*)nullptr, asum, colsq_part, fz, sm);
    return check_launch("lpm_vlad_aggregate_raw_kmajor_smx_fwd");
}

extern "C" int lpm_vlad_row_scales(const float* colsq_part, int P, int B, int K, float* scale, float* colsq, float* csq, float* gsq,
                                   lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(colsq_part && scale && colsq && csq && gsq, LPM_ERR_BADARG, "lpm_vlad_row_scales: null pointer");
    LPM_REQUIRE(B > 0 && P > 0 && K > 0 && K <= 1024, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_vlad_row_scales: need 0 < K <= 1024 (K=%d)", K);
    hipEvent_t e0, e1;
    if (K >= 256 && timing_request(LPM_TIMING_FINALIZE, &e0, &e1))
        hipExtLaunchKernelGGL(vlad_row_scales_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, colsq_part, P, K, scale, colsq, csq,
                              gsq);
    else
        hipLaunchKernelGGL(vlad_row_scales_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, colsq_part, P, K, scale, colsq, csq, gsq);
    return check_launch("lpm_vlad_row_scales");
}

// ---- K2 with the finalize pass fused in (see T3Fused): workspace = [colsq_part B*P*K floats | arrive B | fail B] --------------------
extern "C" int lpm_vlad_fused_supported(int D, int K) {
    return (lpm_vlad_tiles3_supported(D, K) && K <= 512 && (int64_t)(D / 128) * K * 4 <= lpm::T3_NS * lpm::T3_STAGE) ? 1 : 0;
}
extern "C" size_t lpm_vlad_fused_workspace_bytes(int B, int D, int K) {
    return ((size_t)B * (D / 128) * K + (size_t)B + (size_t)B * (K / 128) * (D / 128)) * sizeof(float);
}

extern "C" int lpm_vlad_aggregate_fused_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                            float* nrm, float* out, float* asum, float* colsq, float* csq, float* gsq, void* workspace,
                                            size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && nrm && out && asum && colsq && csq && gsq && workspace, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_fused_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_fused_fwd: RESIDUAL needs centres");
    LPM_REQUIRE(B > 0 && T > 0 && lpm_vlad_fused_supported(D, K), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_fused_fwd: need D %% 128 == 0, K %% 128 == 0, K <= 512, D/128 * K <= 12288 (D=%d K=%d)", D, K);
    LPM_REQUIRE(workspace_bytes >= lpm_vlad_fused_workspace_bytes(B, D, K), LPM_ERR_WORKSPACE,
                "lpm_vlad_aggregate_fused_fwd: workspace too small");
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)nrm | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0,
                LPM_ERR_BADARG, "lpm_vlad_aggregate_fused_fwd: pointers must be 16-byte aligned");
    const int S = (T + 15) / 16, KT = K / 32, P = D / 128;
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    unsigned* arrive = (unsigned*)(part + (size_t)B * P * K);
    unsigned* fail = arrive + B;
    const size_t ntile = (size_t)B * (K / 128) * (D / 128);
    if (hipMemsetAsync(arrive, 0, ((size_t)B + ntile) * sizeof(unsigned), s) != hipSuccess) {      // counters and flags: zero at every launch
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_fused_fwd: cannot clear the arrival counters");
        return LPM_ERR_LAUNCH;
    }
    T3Fused fz{};
    fz.out = out; fz.kmajor = (flags & LPM_VLAD_OUT_KMAJOR) ? 1 : 0; fz.store_u = (flags & LPM_VLAD_NRM_RAW) ? 1 : 0;
    fz.colsq = colsq; fz.csq = csq; fz.gsq = gsq; fz.arrive = arrive; fz.fail = fail;
    fz.debug_fallback = (flags & LPM_VLAD_DEBUG_FALLBACK) ? 1 : 0;
    const size_t lds = (size_t)T3_NS * T3_STAGE + (4 * 128 + 128 + 16) * sizeof(float);
    auto kern = vlad_aggregate_tiles3_kernel<true, 2>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("lpm_vlad_aggregate_fused_fwd: cannot reserve %zu bytes of LDS", lds);
        return LPM_ERR_LAUNCH;
    }
    dim3 grid(B * (K / 128) * (D / 128));
    hipEvent_t e0, e1;
    if (D >= 1024 && timing_request(LPM_TIMING_K2, &e0, &e1))      // (the video stream's launches only: lpm_common.h)
        hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, s, e0, e1, 0, (const uint4*)at, (const uint4*)xt, centres, T, D, K, S, KT,
                              residual | t3_nt(), nrm, asum, part, fz, T3Softmax{});
    else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const uint4*)at, (const uint4*)xt, centres, T, D, K, S, KT, residual | t3_nt(), nrm, asum,
                           part, fz, T3Softmax{});
    // follow-up for tiles whose workgroup gave up waiting for its clip (fail flag set; none in practice)
    hipLaunchKernelGGL(vlad_fused_fixup_kernel, grid, dim3(256), 0, s, nrm, part, fail, D, K, fz);
    return check_launch("lpm_vlad_aggregate_fused_fwd");
}

static int vlad_finalize2_impl(float* nrm, const float* colsq_part, int P, int B, int D, int K, int flags, float* out, int64_t out_bs,
                               float* colsq, float* csq, float* gsq, lpm_stream_t stream);
extern "C" int lpm_vlad_finalize2_fwd(float* nrm, const float* colsq_part, int P, int B, int D, int K, int flags, float* out,
                                      float* colsq, float* csq, float* gsq, lpm_stream_t stream) {
    return vlad_finalize2_impl(nrm, colsq_part, P, B, D, K, flags, out, (int64_t)D * K, colsq, csq, gsq, stream);
}
// ... with the clips' descriptors out_batch_stride elements apart in `out` (>= D * K, a multiple of 4): a column slot of a wider
// [B, total] buffer (ops.DescriptorSlots) -- the concatenation of the streams' descriptors (frame_level_models.py:2309) without a copy
extern "C" int lpm_vlad_finalize2_fwd_ld(float* nrm, const float* colsq_part, int P, int B, int D, int K, int flags, float* out,
                                         int64_t out_batch_stride, float* colsq, float* csq, float* gsq, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(out_batch_stride >= (int64_t)D * K && out_batch_stride % 4 == 0, LPM_ERR_BADARG,
                "lpm_vlad_finalize2_fwd_ld: the batch stride must be >= D * K and a multiple of 4");
    LPM_REQUIRE(out_batch_stride == (int64_t)D * K || !(flags & LPM_VLAD_OUT_KMAJOR) || K > 512, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_finalize2_fwd_ld: the 16-byte k-major form writes contiguous descriptors only");
    return vlad_finalize2_impl(nrm, colsq_part, P, B, D, K, flags, out, out_batch_stride, colsq, csq, gsq, stream);
}
static int vlad_finalize2_impl(float* nrm, const float* colsq_part, int P, int B, int D, int K, int flags, float* out, int64_t out_bs,
                               float* colsq, float* csq, float* gsq, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(nrm && colsq_part && out && colsq && csq && gsq, LPM_ERR_BADARG, "lpm_vlad_finalize2_fwd: null pointer");
    LPM_REQUIRE(B > 0 && P > 0 && D % 32 == 0 && K % 4 == 0 && K <= 4096, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_finalize2_fwd: need D %% 32 == 0, K %% 4 == 0 (D=%d K=%d)", D, K);
    dim3 grid(D / 32, B);
    const int keep_u = (flags & LPM_VLAD_NRM_RAW) ? 1 : 0;
    const int out_bf16 = (flags & LPM_VLAD_OUT_BF16) ? 1 : 0;
    const int nrm_bf16 = (flags & LPM_VLAD_NRM_BF16) ? 1 : 0;
    LPM_REQUIRE(!(out_bf16 || nrm_bf16) || !(flags & LPM_VLAD_OUT_KMAJOR), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_finalize2_fwd: bf16 storage is implemented for the reference's d-major layout only");
    LPM_REQUIRE(!nrm_bf16 || keep_u, LPM_ERR_BADARG, "lpm_vlad_finalize2_fwd: LPM_VLAD_NRM_BF16 goes with LPM_VLAD_NRM_RAW");
    const size_t lds = (size_t)(K + 32 * 33 + 4) * sizeof(float);
    static const int wide = [] { const char* e = getenv("LPM_FINALIZE_KMAJOR4"); return e ? atoi(e) : 32; }();   // 0: scalar form (A/B)
    if ((flags & LPM_VLAD_OUT_KMAJOR) && wide && K <= 512 && (((uintptr_t)nrm | (uintptr_t)out) & 15) == 0) {
        const int R = (wide == 64 && D % 64 == 0) ? 64 : 32;
        const size_t lds4 = (size_t)(K + 4 + R * (K + 2)) * sizeof(float);
        dim3 grid4(D / R, B);
        auto launch = [&](auto kern) -> int {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4) != hipSuccess) {
                (void)hipGetLastError();
                set_error("lpm_vlad_finalize2_fwd: cannot reserve %zu bytes of LDS", lds4);
                return LPM_ERR_LAUNCH;
            }
            hipEvent_t e0, e1;
            if (D >= 1024 && timing_request(LPM_TIMING_FINALIZE, &e0, &e1))
                hipExtLaunchKernelGGL(kern, grid4, dim3(256), lds4, (hipStream_t)stream, e0, e1, 0, nrm, colsq_part, P, D, K, out, colsq, csq,
                                      gsq, keep_u);
            else
                hipLaunchKernelGGL(kern, grid4, dim3(256), lds4, (hipStream_t)stream, nrm, colsq_part, P, D, K, out, colsq, csq, gsq, keep_u);
            return check_launch("lpm_vlad_finalize2_fwd");
        };
        return R == 64 ? launch(vlad_finalize2_kmajor4_kernel<64>) : launch(vlad_finalize2_kmajor4_kernel<32>);
    }
    if (flags & LPM_VLAD_OUT_KMAJOR)
        hipLaunchKernelGGL(vlad_finalize2_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, nrm, colsq_part, P, D, K, out, colsq,
                           csq, gsq, keep_u, 0, 0, out_bs);
    else {
        hipEvent_t e0, e1;
        if (D >= 1024 && timing_request(LPM_TIMING_FINALIZE, &e0, &e1))
            hipExtLaunchKernelGGL(vlad_finalize2_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, e0, e1, 0, nrm, colsq_part, P, D, K,
                                  out, colsq, csq, gsq, keep_u, out_bf16, nrm_bf16, out_bs);
        else
            hipLaunchKernelGGL(vlad_finalize2_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, nrm, colsq_part, P, D, K, out,
                               colsq, csq, gsq, keep_u, out_bf16, nrm_bf16, out_bs);
    }
    return check_launch("lpm_vlad_finalize2_fwd");
}
