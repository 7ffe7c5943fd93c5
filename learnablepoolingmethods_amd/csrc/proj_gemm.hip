// a9, the VLAD -> hidden projection (frame_level_models.py:2314-2319: tf.matmul(vlad, hidden1_weights)) as weight-stream kernels.
// M (clips, <= 128) rows against a [Kd, N] fp32 weight of 0.55 GB (cfg-2: Kd = 270 336, N = 512) to 2.2 GB (cfg-5): the weight is read
// ONCE per pass straight from its fp32 master copy -- no bf16 image of it exists anywhere -- and split into bf16 hi / lo planes in
// registers, so the pass is bound by the 4 bytes per weight it has to read (a library fp32 GEMM streams it at 2.7 TB/s).
//
//   forward  y[M, N]  = x[M, Kd] . W[Kd, N]        split-K: every workgroup owns a range of 16-row slabs of W for one 512-column block,
//                                                   partial sums [split][M][N] + a reduce pass (tile_gemm's tg_reduce_splits layout)
//   backward dx[M, Kd] = dy[M, N] . W^T            (lpm_proj_dx below)
//
// Forward: 8 computing waves x 64 columns + a loader wave for x; per slab the workgroup brings W[16][512] (32 KB, contiguous when N = 512)
// and, per pair of slabs, x[M][32] into LDS rings by LDS-DMA -- both images are plain row-major, which is exactly what a lane-linear DMA
// writes.  The B fragment of v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane while W is n-contiguous: a lane gathers its 8 values
// with eight ds_read_b32 down a column (bank = n mod 32: conflict-free), splits them (hi = bf16(w), lo = bf16(w - hi)) and keeps the two
// planes in registers; x fragments are two ds_read_b128 per tile.  3 MFMAs per product (x_h W_h + x_h W_l + x_l W_h), fp32
// accumulation: the error of the split-bf16 tile GEMMs (~5e-6).
#include "tile_gemm.h"

namespace lpm {

// Ring depth: the pass is a pure stream, so what matters is bytes in flight per CU (HBM latency under load ~2-3 us x 25 GB/s per CU
// = 50-75 KB): four stages of 32 KB, three of them in flight behind the one being consumed.
constexpr int PJ_NS = 4;                           // ring stages
constexpr int PJ_WBYTES = 16 * 512 * 4;            // W slab
#ifndef LPM_PJ_AUX
#define LPM_PJ_AUX 2                                // (a variant build with 0 = the default policy: tools/build_proj_policy_variant.sh, A/B)
#endif
constexpr int PJ_AUX = LPM_PJ_AUX;
#ifndef LPM_PJ_SC_ONE_LOADER_MAX
#define LPM_PJ_SC_ONE_LOADER_MAX 4                  // pieces per pair up to which ONE wave loads the scaled x (beyond: two waves, half the pieces each)
#endif
#ifndef LPM_PJ_SC_SPLIT
#define LPM_PJ_SC_SPLIT 0
#endif
constexpr bool PJ_SC_SPLIT = LPM_PJ_SC_SPLIT != 0;                 // LDS-DMA cache policy of the weight stream: nt (every byte is read once, by one CU)

// MT = row tiles (ceil(M / 32)).  Eight computing waves (64 columns each) bring the weight slabs in; a NINTH wave brings ALL of x in and
// does nothing else, in PAIRS of slabs: 128 bytes per row = a whole cache line (round 3; before, 2 MT of the computing waves asked for
// every line of x twice, 64 bytes at a time, and their in-order vmcnt queue held the contiguous weight pieces back behind the scattered
// x pieces).  x has its own ring of XD pair-buffers of NP pieces (8 rows x 128 bytes each; NP = 4 MT, or 10 for M <= 80 so that three
// buffers fit) behind the four weight stages; a row's eight 16-byte parts sit XOR-swizzled by (row & 7) -- through the SOURCE address,
// the LDS image stays lane-linear -- so the 32 rows a fragment read touches, 128 bytes apart, spread over all banks.  Split ranges
// start on even slabs so that a pair is one aligned line.
//
// Round 4 -- THE RING IS READ WITH ds_reads THE COMPILER CANNOT SEE.  To LLVM an LDS-DMA load is a store to LDS that any later LDS read may
// alias: the plain C++ reads of rounds 1-3 each got an `s_waitcnt vmcnt(0)` in front of them, the loop waited for EVERY stage in flight --
// also the one it had requested a moment before -- and the four-stage ring ran as "request, wait a full HBM round trip, compute"
// (cfg-2: 144 us; counted waits as written here: 107 us.  tools/scan_lds_dma_waits.py lists such loops, tests/test_build_flags.py keeps
// these kernels off that list).  All W reads of a stage are requested at once, the x fragments one row tile ahead; each wait hands its
// registers over through "+v" operands (the compiler may not touch them earlier, and no scalar load sits in the loop: LDS returns in order,
// so the lgkmcnt values below count ds_reads only).
// SC (round 4, the lazily normalised d-major descriptor of NetVladV2 / the model without encoders): x is TWO column blocks -- columns
// < pp.n1a from x (row stride ldx), each multiplied by pp.scale[row][column % pp.ks] as it is brought in (the un-normalised residual sums
// of the video stream and their per-(clip, cluster) scale 1 / (n_k sqrt g): frame_level_models.py:2819-2822 applied where the operand is
// read), the rest from pp.x2 as they are (the audio stream's descriptor) -- tf.concat at :2309 / :2445 never materialises.  Only the
// loader wave differs: its pieces go through registers (load, multiply, ds_write into the same image the LDS-DMA leaves), one pair ahead.
struct ProjParts {
    const float* x2;
    int64_t ldx2, n1a;
    const float* scale;
    int ks;
    int x1_bf16;               // the first block is stored as bf16 (bf16 storage of the descriptor: BASELINE configs[4]); ldx in elements
                               // (selects the SC = 2 instantiation)
};
// (SC with more than 10 pieces per pair: TWO loader waves, half the pieces each -- sixteen pieces' data and scales do not fit one wave's
// registers beside what the computing path sets the kernel's allocation to)
__host__ __device__ constexpr int pj_loaders(int NP, int SC) { return (SC != 0 && NP > LPM_PJ_SC_ONE_LOADER_MAX) ? 2 : 1; }
// W16 (round 5, BASELINE configs[4]: "master fp32 + bf16 compute copy"): W points at the bf16 COPY of the weight
// (written by the Adam epilogue, csrc/tile_gemm.hip) -- half the bytes of the stream.  A slab is 16 rows x 512 columns x 2 bytes = 16 KB,
// the ring EIGHT stages (seven in flight: 112 KB per CU), a wave brings two 1 KB rows per stage; a lane gathers its 8 values down the
// column with eight ds_read_u16 and packs pairs into the B fragment (no split: the copy IS the operand), x is rounded once to
// bf16 as it is read (the descriptor of this configuration is a bf16 tensor): ONE MFMA per product.
constexpr int PJ16_NS = 8;
constexpr int PJ16_WBYTES = 16 * 512 * 2;
static_assert(PJ16_NS * PJ16_WBYTES == PJ_NS * PJ_WBYTES, "the x ring sits behind the weight ring at the same offset in both forms");
template <int MT, int NP, int XD, int SC = 0, bool W16 = false>
__global__ __launch_bounds__(512 + 64 * pj_loaders(NP, SC), 1) void proj_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int M, int64_t Kd,
                                                          int N, int nslab, int splits, float* __restrict__ part, const ProjParts pp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int nb = blockIdx.y, sp = blockIdx.x;                  // 512-column block, split
    const int npair = nslab >> 1;
    const int s0 = 2 * (int)((int64_t)npair * sp / splits);
    const int s1 = sp == splits - 1 ? nslab : 2 * (int)((int64_t)npair * (sp + 1) / splits);
    const int ns = s1 - s0;
    constexpr int PJ_XPAIR = NP * 1024;            // one pair-buffer of x

    if (wave >= 8) {
        // the x loader: piece p = 0 .. NP - 1 of a pair = rows 8 p .. 8 p + 7 (clamped to M - 1: rows >= M are never stored), lane ->
        // (row r = 8 p + lane / 8, LDS slot q = lane % 8) holding source part q ^ (r & 7); parts 0-3 = slab 2 j, parts 4-7 = slab 2 j + 1
        // (the last pair of an odd range: its second slab is re-read from the first -- nobody consumes it)
        const int q = lane & 7;
        const int part = q ^ ((lane >> 3) & 7);                    // (8 p + lane / 8) & 7 == lane / 8
        if constexpr (SC != 0) {
            const int nss = (ns + 1) >> 1;
            // SC 1 (fp32 sums): xv = the pair's x pieces, sv = their scales, riding along where the registers allow (NP <= 10; otherwise the
            // scales are fetched when the pair is written -- from L2, one dependent round trip per piece: slow, and not a shape any
            // configuration runs).  SC 2 (bf16 sums): xh = the pieces as they are stored (two registers each), sv = their scales -- or, for a
            // pair of the second block (fp32, no scales), the pieces themselves.
            constexpr int NPW = NP / pj_loaders(NP, SC);       // pieces of this loader wave: p0 .. p0 + NPW - 1
            const int p0 = (wave - 8) * NPW;
            constexpr bool HOLD = SC == 2 || NPW <= 10;
            f32x4 xv[SC == 1 ? NPW : 1], sv[HOLD ? NPW : 1];
            uint2 xh[SC == 2 ? NPW : 1];
            bool firstv = true;
            int kcv = 0;
            auto load = [&](int j) {
                int64_t col = (int64_t)s0 * 16 + (int64_t)j * 32 + part * 4 - ((part >= 4 && 2 * j + 1 >= ns) ? 16 : 0);
                const bool first = col < pp.n1a;               // (a pair never straddles the blocks: n1a is a multiple of 32, s0 is even)
                const int kc = first ? (int)(col % pp.ks) : 0;
                firstv = first; kcv = kc;
#pragma unroll
                for (int p = 0; p < NPW; ++p) {
                    const int row = min((p0 + p) * 8 + (lane >> 3), M - 1);
                    if constexpr (SC == 2) {
                        if (first) {
                            xh[p] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + (int64_t)row * ldx + col);
                            sv[p] = *reinterpret_cast<const f32x4*>(pp.scale + (int64_t)row * pp.ks + kc);
                        } else {
                            sv[p] = *reinterpret_cast<const f32x4*>(pp.x2 + (int64_t)row * pp.ldx2 + (col - pp.n1a));
                        }
                    } else {
                        const float* src = first ? x + (int64_t)row * ldx + col : pp.x2 + (int64_t)row * pp.ldx2 + (col - pp.n1a);
                        xv[p] = *reinterpret_cast<const f32x4*>(src);
                        if constexpr (HOLD)
                            sv[p] = first ? *reinterpret_cast<const f32x4*>(pp.scale + (int64_t)row * pp.ks + kc) : f32x4{1.f, 1.f, 1.f, 1.f};
                    }
                }
            };
            auto store = [&](int j) {
                unsigned char* st = smem + PJ_NS * PJ_WBYTES + (j % XD) * PJ_XPAIR;
#pragma unroll
                for (int p = 0; p < NPW; ++p) {
                    f32x4 v;
                    if constexpr (SC == 2) {
                        const uint2 u = xh[p];
                        const f32x4 xf = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                          __uint_as_float(u.y & 0xffff0000u)};
                        v = firstv ? xf * sv[p] : sv[p];
                    } else if constexpr (HOLD) {
                        v = xv[p] * sv[p];
                    } else {
                        const f32x4 sc = firstv ? *reinterpret_cast<const f32x4*>(pp.scale + (int64_t)min((p0 + p) * 8 + (lane >> 3), M - 1) * pp.ks + kcv)
                                                : f32x4{1.f, 1.f, 1.f, 1.f};
                        v = xv[p] * sc;
                    }
                    *reinterpret_cast<f32x4*>(st + (p0 + p) * 1024 + lane * 16) = v;
                }
            };
            load(0);
            store(0);
            if (1 < nss) load(1);
            for (int s = 0; s < ns; ++s) {
                __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): this wave's LDS writes are done
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // after the barrier of stage 2 j every wave is past pair j - 1: the buffer pair j + 1 goes to (pair j + 1 - XD's) is free.
                // (LPM_PJ_SC_SPLIT=1 -- the requests for pair j + 2 one stage later than the write of pair j + 1, so that the load issues
                // do not delay this wave's arrival at the next barrier -- measured at cfg-3: 318 us against 161: one stage is not enough
                // for the pieces to arrive, eighty 128-byte lines exactly 1 MB apart)
                const int j = s >> 1;
                if (!(s & 1)) {
                    if (j + 1 < nss) store(j + 1);
                    if (!PJ_SC_SPLIT && j + 2 < nss) load(j + 2);
                } else if (PJ_SC_SPLIT && j + 2 < nss) {
                    load(j + 2);
                }
            }
            return;
        }
        const float* xs[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xs[p] = x + (int64_t)min(p * 8 + (lane >> 3), M - 1) * ldx + (int64_t)s0 * 16 + part * 4;
        const int nss = (ns + 1) >> 1;
        auto issue_x = [&](int j) {
            unsigned char* st = smem + PJ_NS * PJ_WBYTES + (j % XD) * PJ_XPAIR;
            const int64_t off = (int64_t)j * 32 - ((part >= 4 && 2 * j + 1 >= ns) ? 16 : 0);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs[p] + off),
                                                 (__attribute__((address_space(3))) void*)(st + p * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int j = 0; j < XD - 1; ++j)
            if (j < nss) issue_x(j);
        for (int s = 0; s < ns; ++s) {
            // pair j = s / 2 is consumed by stages 2 j and 2 j + 1; at the barrier of stage 2 j the buffer of pair j - 1 is free again and
            // pairs j .. j + XD - 2 are on their way (LDS-DMA loads of a wave retire in order: NP per pair)
            if (!(s & 1)) {
                if (XD > 2 && (s >> 1) + XD - 2 < nss) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 2) * NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!(s & 1) && (s >> 1) + XD - 1 < nss) issue_x((s >> 1) + XD - 1);
        }
        return;
    }
    // W DMA: piece p = wave * 4 + j (j < 4): slab row p >> 1, 1 KB half (p & 1) of the row's 512-column segment
    // (W16: piece p = wave * 2 + j (j < 2) = slab row p, its whole 512-column segment)
    constexpr int WNS = W16 ? PJ16_NS : PJ_NS, WSTAGE = W16 ? PJ16_WBYTES : PJ_WBYTES, WPC = W16 ? 2 : 4;
    const float* wsrc[WPC];
#pragma unroll
    for (int j = 0; j < WPC; ++j) {
        const int p = wave * WPC + j;
        if (W16) wsrc[j] = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(W) + ((int64_t)s0 * 16 + p) * N + nb * 512 + lane * 8);
        else wsrc[j] = W + ((int64_t)s0 * 16 + (p >> 1)) * N + nb * 512 + (p & 1) * 256 + lane * 4;
    }
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % WNS) * WSTAGE;
#pragma unroll
        for (int j = 0; j < WPC; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + (int64_t)s * (W16 ? 8 : 16) * N),
                                             (__attribute__((address_space(3))) void*)(st + (wave * WPC + j) * 1024), 16, 0, PJ_AUX);
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;
    // LDS byte addresses of this lane's reads: the weight column (row 8 half of the slab, column wave * 64 + l31; a fragment = 8 rows down
    // that column, bank = column mod 32: conflict-free) and, per row tile, part 2 half of its row of an x pair-buffer (rows past the
    // buffer, >= M, are never stored: clamped)
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned w_lane = (unsigned)((8 * half) * 512 + wave * 64 + l31) * (W16 ? 2u : 4u);
    unsigned x_lane[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int xrow = (NP < 4 * MT) ? min(m * 32 + l31, NP * 8 - 1) : m * 32 + l31;
        x_lane[m] = (unsigned)(xrow * 128 + (((2 * half) ^ (xrow & 7)) << 4));
    }

#pragma unroll
    for (int s = 0; s < WNS - 1; ++s)
        if (s < ns) issue(s);
    for (int s = 0; s < ns; ++s) {
        // stage s has landed when at most the pieces of the younger stages in flight (WNS - 2, fewer at the end) remain: WPC per stage
        const int young = min(WNS - 2, ns - 1 - s);
        if (W16) {
            switch (young) {
                case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            if (young == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (young == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();              // stage s is in LDS for everyone; stage (s - 1) % NS is free again
        asm volatile("" ::: "memory");
        if (s + WNS - 1 < ns) issue(s + WNS - 1);
        const unsigned wa = smem_lds + (unsigned)(s % WNS) * WSTAGE + w_lane;
        const unsigned xa = smem_lds + PJ_NS * PJ_WBYTES + (unsigned)((s >> 1) % XD) * PJ_XPAIR;
        if constexpr (W16) {
            // B fragments: register i of column c = (row 2 i, row 2 i + 1) of the lane's eight rows, low half first
            unsigned bw[2][8];
            tg_u32x4 bq[2];
            f32x4 xq[2][2];
            constexpr bool PRE = MT <= 3;
            auto read_xq = [&](int m) {
                const unsigned a0 = xa + (x_lane[m] ^ ((unsigned)(s & 1) << 6));
                asm volatile("ds_read_b128 %0, %1" : "=v"(xq[PRE ? (m & 1) : 0][0]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1" : "=v"(xq[PRE ? (m & 1) : 0][1]) : "v"(a0 ^ 16u));
            };
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e)      // (zero-extending reads + one v_lshl_or per pair: with SRAM ECC a d16 load clears the other half)
                    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(bw[c][e]) : "v"(wa), "n"(e * 1024 + c * 64));
            if (PRE) read_xq(0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                asm volatile("s_waitcnt lgkmcnt(%8)"
                             : "+v"(bw[c][0]), "+v"(bw[c][1]), "+v"(bw[c][2]), "+v"(bw[c][3]), "+v"(bw[c][4]), "+v"(bw[c][5]), "+v"(bw[c][6]), "+v"(bw[c][7])
                             : "n"((1 - c) * 8 + (PRE ? 2 : 0)));
                bq[c] = tg_u32x4{bw[c][0] | (bw[c][1] << 16), bw[c][2] | (bw[c][3] << 16), bw[c][4] | (bw[c][5] << 16), bw[c][6] | (bw[c][7] << 16)};
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4* xc = xq[PRE ? (m & 1) : 0];
                if (!PRE) read_xq(m);
                else if (m + 1 < MT) read_xq(m + 1);
                if (PRE && m + 1 < MT) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xc[0]), "+v"(xc[1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0]), "+v"(xc[1]));
                const float v[8] = {xc[0][0], xc[0][1], xc[0][2], xc[0][3], xc[1][0], xc[1][1], xc[1][2], xc[1][3]};
                uint4 hi, lo;
                tg_split8(v, hi, lo);
                const tg_u32x4 ah = tg_u32x4{hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[m][c] = tg_mfma(ah, bq[c], acc[m][c]);
            }
            continue;
        }
        float wv[2][8];
        constexpr bool PRE = MT <= 3;              // x fragments requested one row tile ahead (MT = 4: no registers left for that)
        f32x4 xv[2][2];
        auto read_x = [&](int m) {
            const unsigned a0 = xa + (x_lane[m] ^ ((unsigned)(s & 1) << 6));               // parts p0, p0 + 1 of the row: 16 bytes apart after the swizzle
            asm volatile("ds_read_b128 %0, %1" : "=v"(xv[PRE ? (m & 1) : 0][0]) : "v"(a0));
            asm volatile("ds_read_b128 %0, %1" : "=v"(xv[PRE ? (m & 1) : 0][1]) : "v"(a0 ^ 16u));
        };
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(wv[c][e]) : "v"(wa), "n"(e * 2048 + c * 128));
        if (PRE) read_x(0);
        tg_u32x4 bh[2], bl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            asm volatile("s_waitcnt lgkmcnt(%8)"
                         : "+v"(wv[c][0]), "+v"(wv[c][1]), "+v"(wv[c][2]), "+v"(wv[c][3]), "+v"(wv[c][4]), "+v"(wv[c][5]), "+v"(wv[c][6]), "+v"(wv[c][7])
                         : "n"((1 - c) * 8 + (PRE ? 2 : 0)));
            uint4 hi, lo;
            tg_split8(wv[c], hi, lo);
            bh[c] = tg_u32x4{hi.x, hi.y, hi.z, hi.w};
            bl[c] = tg_u32x4{lo.x, lo.y, lo.z, lo.w};
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4* xc = xv[PRE ? (m & 1) : 0];
            if (!PRE) read_x(m);
            else if (m + 1 < MT) read_x(m + 1);
            if (PRE && m + 1 < MT) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xc[0]), "+v"(xc[1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0]), "+v"(xc[1]));
            const float v[8] = {xc[0][0], xc[0][1], xc[0][2], xc[0][3], xc[1][0], xc[1][1], xc[1][2], xc[1][3]};
            uint4 hi, lo;
            tg_split8(v, hi, lo);
            const tg_u32x4 ah = tg_u32x4{hi.x, hi.y, hi.z, hi.w}, al = tg_u32x4{lo.x, lo.y, lo.z, lo.w};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[m][c] = tg_mfma(ah, bh[c], acc[m][c]);
                acc[m][c] = tg_mfma(ah, bl[c], acc[m][c]);
                acc[m][c] = tg_mfma(al, bh[c], acc[m][c]);
            }
        }
    }
    // partial sums of this split: acc[m][c][r] = y[row m * 32 + mfma32_row(r, lane)][column nb * 512 + wave * 64 + c * 32 + l31]
    float* pb = part + ((int64_t)sp * M) * N + nb * 512 + wave * 64 + l31;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + mfma32_row(r, lane);
            if (row < M) {
#pragma unroll
                for (int c = 0; c < 2; ++c) pb[(int64_t)row * N + c * 32] = acc[m][c][r];
            }
        }
}

// out[i] = sum_z part[z][i] (float4 granularity): 64 outputs x 4 quarters of the splits per workgroup, the quarters added in a fixed
// order -- the result does not depend on scheduling
__global__ __launch_bounds__(256) void proj_reduce_kernel(const float4* __restrict__ part, int Z, int64_t n4, float4* __restrict__ out) {
    __shared__ float4 sh[3][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    const int z0 = Z * q / 4, z1 = Z * (q + 1) / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4)
        for (int z = z0; z < z1; ++z) {
            const float4 p = part[(int64_t)z * n4 + i];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
    if (q) sh[q - 1][o] = s;
    __syncthreads();
    if (q == 0 && i < n4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 p = sh[k][o];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        out[i] = s;
    }
}

static int proj_splits(int64_t Kd, int N) {
    const int nb = N / 512, nslab = (int)(Kd / 16);
    int sp = 256 / nb;                             // one workgroup per CU
    if (sp > nslab / 8) sp = nslab / 8;            // >= 8 slabs per workgroup
    return sp < 1 ? 1 : sp;
}

// ---- backward: dx[M, Kd] = dy[M, N] . W^T ------------------------------------------------------------------------------------------------
// Workgroup = 256 rows of W (= 256 columns of dx), 8 waves x 32 rows, the whole reduction over N inside.  Here the B fragment IS
// W's layout: lane (k-row l31, half) wants 8 consecutive n of its row.  Per MFMA step (16 n) a wave brings its 32 rows x 64 B in by
// two LDS-DMAs (16 rows x 64 B each), XOR-swizzled at 16-byte granularity through the SOURCE address -- the LDS image stays
// lane-linear -- so that the 32 lanes of a fragment read, 64 bytes apart, spread over the banks.  dy arrives as split-bf16 row tiles
// (lpm_split_rows_tiles of the [M, N] matrix: 0.2-1 MB, L2-resident) and goes through the same ring.  Small stages (22-24 KB), SIX of
// them: five in flight per CU keep the weight stream fed (three stages of 44 KB, first form: 2.7 TB/s).
constexpr int PD_NS = 6;
constexpr int PD_WBYTES = 8 * 32 * 64;             // 8 waves x 32 rows x 16 floats = 16 KB
template <int MT>
__host__ __device__ constexpr int pd_stage() { return PD_WBYTES + MT * 2048; }    // + MT row tiles x (hi, lo) of one n-step

template <int MT>
__global__ __launch_bounds__(512, 2) void proj_dx_kernel(const uint4* __restrict__ dyt, const float* __restrict__ W, int M, int64_t Kd, int N,
                                                         float* __restrict__ dx, int64_t lddx) {
    constexpr int STAGE = pd_stage<MT>();
    constexpr int NAP = MT * 2;                    // dy pieces (1 KB) per stage: one per wave for waves < NAP
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int64_t k0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + wave * 32;     // this wave's 32 rows of W
    const int NB = N / 16, CS = N / 16;            // n-steps = stages; row-tile stride of dyt
    // W DMA: piece j (0..1) = rows 16 j .. 16 j + 15 of the wave's 32, lane -> (row r = 16 j + lane / 4, 16-byte slot q = lane % 4);
    // LDS slot q of row r holds source part q ^ (r & 3)
    const float* wsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * j + (lane >> 2), q = lane & 3;
        const int64_t row = min(k0 + r, Kd - 1);
        wsrc[j] = W + row * N + ((q ^ (r & 3)) * 4);
    }
    const bool awave = wave < NAP;                 // dy piece of this wave: (row tile wave >> 1, plane wave & 1)
    const uint4* asrc = dyt + ((int64_t)(wave >> 1) * CS) * 128 + (wave & 1) * 64 + lane;
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % PD_NS) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + s * 16),
                                             (__attribute__((address_space(3))) void*)(st + wave * 2048 + j * 1024), 16, 0, PJ_AUX);
        if (awave)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + (int64_t)s * 128),
                                             (__attribute__((address_space(3))) void*)(st + PD_WBYTES + wave * 1024), 16, 0, 0);
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned w_lane = (unsigned)(wave * 2048 + l31 * 64 + (((2 * half) ^ (l31 & 3)) << 4));      // this lane's row (64 B), slot of part 2 half
    const unsigned a_lane = (unsigned)(PD_WBYTES + lane * 16);
#pragma unroll
    for (int s = 0; s < PD_NS - 1; ++s)
        if (s < NB) issue(s);
    for (int s = 0; s < NB; ++s) {
        // younger stages in flight behind stage s: up to PD_NS - 2 = 4, of 3 (2) pieces of this wave each
        const int young = min(PD_NS - 2, NB - 1 - s);
        if (awave) {
            if (young == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (young == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (young == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (young == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (young == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (young == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (young == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (young == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + PD_NS - 1 < NB) issue(s + PD_NS - 1);
        // (the ring is read with ds_reads the compiler cannot see -- proj_fwd_kernel's comment)
        const unsigned sb = smem_lds + (unsigned)(s % PD_NS) * STAGE;
        f32x4 b0, b1;
        tg_u32x4 ah[MT], al[MT];
        // 8 consecutive n of row l31: 16-byte slots 2 half and 2 half + 1 (swizzled: slot q holds part q ^ (row & 3))
        asm volatile("ds_read_b128 %0, %1" : "=v"(b0) : "v"(sb + w_lane));
        asm volatile("ds_read_b128 %0, %1" : "=v"(b1) : "v"(sb + (w_lane ^ 16u)));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[m]) : "v"(sb + a_lane), "n"((m * 2 + 0) * 1024));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[m]) : "v"(sb + a_lane), "n"((m * 2 + 1) * 1024));
        }
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(2 * MT));
        const float v[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        uint4 hi, lo;
        tg_split8(v, hi, lo);
        const tg_u32x4 bh = tg_u32x4{hi.x, hi.y, hi.z, hi.w}, bl = tg_u32x4{lo.x, lo.y, lo.z, lo.w};
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ah[m]), "+v"(al[m]) : "n"(2 * (MT - 1 - m)));
            acc[m] = tg_mfma(ah[m], bh, acc[m]);
            acc[m] = tg_mfma(ah[m], bl, acc[m]);
            acc[m] = tg_mfma(al[m], bh, acc[m]);
        }
    }
    // dx[row][k0 + l31]: the 32 lanes of a half-wave write 128 contiguous bytes per row
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + mfma32_row(r, lane);
            if (row < M && k0 + l31 < Kd) dx[(int64_t)row * lddx + k0 + l31] = acc[m][r];
        }
}

// ---- backward, second form (round 3): every 128-byte line of W is requested ONCE, whole ------------------------------------------------------
// proj_dx_kernel above walks a row of W in 64-byte pieces (one 16-deep MFMA step per ring stage): each 128-byte line is requested as
// two halves by two different stages, and with the stream's non-temporal policy the second half comes from HBM again -- the pass ran at
// 2.2 TB/s (cfg-5) and lost to the library at cfg-2.  Here a stage is 32 columns of the reduction = one whole line per row: a wave
// brings its 32 rows x 128 B in by four LDS-DMAs of 8 rows x 128 B, XOR-swizzled at 16-byte granularity through the SOURCE address
// (LDS slot q of row r holds chunk q ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group -- rows r .. r + 15, two rows per 256-byte
// bank row -- hit 16 different 16-byte slots), and two MFMA steps consume it.  dy's row tiles (L2-resident) ride in the same ring, every
// wave issuing the same number of pieces per stage (piece ids past the end are clamped duplicates into spare slots), so the counted
// vmcnt is one constant.  Three stages of 32 KB + 16 KB.
constexpr int PE_NS = 3;
constexpr int PE_WBYTES = 8 * 32 * 128;            // 8 waves x 32 rows x 32 floats
constexpr int PE_APIECES = 16;                     // dy piece slots per stage (2 per wave): MT row tiles x 2 steps x 2 planes <= 16
constexpr int PE_STAGE = PE_WBYTES + PE_APIECES * 1024;

template <int MT>
__global__ __launch_bounds__(512, 2) void proj_dx2_kernel(const uint4* __restrict__ dyt, const float* __restrict__ W, int M, int64_t Kd, int N,
                                                          float* __restrict__ dx, int64_t lddx) {
    static_assert(4 * MT <= PE_APIECES, "dy pieces per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int64_t k0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + wave * 32;     // this wave's 32 rows of W
    const int NB = N / 32, CS = N / 16;            // stages; 16-deep steps per row tile of dyt
    // W DMA: piece j (0..3) = rows 8 j .. 8 j + 7 of the wave's 32; lane -> (row r = 8 j + lane / 8, LDS slot q = lane % 8)
    const float* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3), q = lane & 7;
        const int64_t row = min(k0 + r, Kd - 1);
        wsrc[j] = W + row * N + ((q ^ ((r >> 1) & 7)) * 4);
    }
    // dy DMA: piece id p = wave + 8 j (j < 2) = (row tile p >> 2, step-in-stage (p >> 1) & 1, plane p & 1); ids >= 4 MT: duplicates of piece 0
    const uint4* asrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave + 8 * j, pc = p < 4 * MT ? p : 0;
        asrc[j] = dyt + ((int64_t)(pc >> 2) * CS + ((pc >> 1) & 1)) * 128 + (pc & 1) * 64 + lane;
    }
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % PE_NS) * PE_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + s * 32),
                                             (__attribute__((address_space(3))) void*)(st + wave * 4096 + j * 1024), 16, 0, PJ_AUX);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + (int64_t)s * 256),
                                             (__attribute__((address_space(3))) void*)(st + PE_WBYTES + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned w_lane = (unsigned)(wave * 4096 + l31 * 128 + (((2 * half) ^ ((l31 >> 1) & 7)) << 4));   // this lane's row (128 B), slot of chunk 2 half
    const unsigned a_lane = (unsigned)(PE_WBYTES + lane * 16);
#pragma unroll
    for (int s = 0; s < PE_NS - 1; ++s)
        if (s < NB) issue(s);
    for (int s = 0; s < NB; ++s) {
        const int young = min(PE_NS - 2, NB - 1 - s);      // younger stages in flight behind stage s, 6 pieces of this wave each
        if (young == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + PE_NS - 1 < NB) issue(s + PE_NS - 1);
        // (the ring is read with ds_reads the compiler cannot see -- proj_fwd_kernel's comment)
        const unsigned sb = smem_lds + (unsigned)(s % PE_NS) * PE_STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // 8 consecutive n of row l31, step t: chunks 4 t + 2 half and + 1 (swizzled: slot q holds chunk q ^ ((row >> 1) & 7))
            f32x4 b0, b1;
            tg_u32x4 ah[MT], al[MT];
            const unsigned wa = sb + (w_lane ^ ((unsigned)t << 6));
            asm volatile("ds_read_b128 %0, %1" : "=v"(b0) : "v"(wa));
            asm volatile("ds_read_b128 %0, %1" : "=v"(b1) : "v"(wa ^ 16u));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[m]) : "v"(sb + a_lane), "n"(((m * 2 + t) * 2 + 0) * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[m]) : "v"(sb + a_lane), "n"(((m * 2 + t) * 2 + 1) * 1024));
            }
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(2 * MT));
            const float v[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            uint4 hi, lo;
            tg_split8(v, hi, lo);
            const tg_u32x4 bh = tg_u32x4{hi.x, hi.y, hi.z, hi.w}, bl = tg_u32x4{lo.x, lo.y, lo.z, lo.w};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ah[m]), "+v"(al[m]) : "n"(2 * (MT - 1 - m)));
                acc[m] = tg_mfma(ah[m], bh, acc[m]);
                acc[m] = tg_mfma(ah[m], bl, acc[m]);
                acc[m] = tg_mfma(al[m], bh, acc[m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + mfma32_row(r, lane);
            if (row < M && k0 + l31 < Kd) dx[(int64_t)row * lddx + k0 + l31] = acc[m][r];
        }
}

// ... from the bf16 compute copy of the weight (round 5, BASELINE configs[4]; proj_fwd_kernel's W16 note): a stage is 64 columns of the
// reduction = one whole 128-byte line per row of the copy, FOUR 16-deep MFMA steps; the lane's B fragment of a step is ONE ds_read_b128
// of its row (chunk 2 t + half, swizzled as above) -- the copy is the operand, nothing is split -- and dy enters by its hi plane alone
// (the gradient rounded once to bf16, as every gradient operand of this configuration's backward is): one MFMA per product.
// dy pieces per stage: MT row tiles x 4 steps (<= 16: two per wave, ids past the end are clamped duplicates).
constexpr int PF_NS = 3;
constexpr int PF_WBYTES = 8 * 32 * 128;            // 8 waves x 32 rows x 64 bf16
constexpr int PF_APIECES = 16;
constexpr int PF_STAGE = PF_WBYTES + PF_APIECES * 1024;

template <int MT>
__global__ __launch_bounds__(512, 1) void proj_dx16_kernel(const uint4* __restrict__ dyt, const unsigned short* __restrict__ W16, int M, int64_t Kd, int N,
                                                           float* __restrict__ dx, int64_t lddx) {
    static_assert(4 * MT <= PF_APIECES, "dy pieces per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int64_t k0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + wave * 32;     // this wave's 32 rows of W
    const int NB = N / 64, CS = N / 16;            // stages; 16-deep steps per row tile of dyt
    // W DMA: piece j (0..3) = rows 8 j .. 8 j + 7 of the wave's 32; lane -> (row r = 8 j + lane / 8, LDS slot q = lane % 8 holding chunk
    // q ^ ((r >> 1) & 7) of the row's eight 16-byte chunks)
    const unsigned short* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3), q = lane & 7;
        const int64_t row = min(k0 + r, Kd - 1);
        wsrc[j] = W16 + row * N + ((q ^ ((r >> 1) & 7)) * 8);
    }
    // dy DMA: piece id p = wave + 8 j (j < 2) = (row tile p >> 2, step-in-stage p & 3), hi plane; ids >= 4 MT: duplicates of piece 0
    const uint4* asrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave + 8 * j, pc = p < 4 * MT ? p : 0;
        asrc[j] = dyt + ((int64_t)(pc >> 2) * CS + (pc & 3)) * 128 + lane;
    }
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % PF_NS) * PF_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + s * 64),
                                             (__attribute__((address_space(3))) void*)(st + wave * 4096 + j * 1024), 16, 0, PJ_AUX);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + (int64_t)s * 512),
                                             (__attribute__((address_space(3))) void*)(st + PF_WBYTES + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned w_lane = (unsigned)(wave * 4096 + l31 * 128 + ((half ^ ((l31 >> 1) & 7)) << 4));       // this lane's row, slot of chunk `half`
    const unsigned a_lane = (unsigned)(PF_WBYTES + lane * 16);
#pragma unroll
    for (int s = 0; s < PF_NS - 1; ++s)
        if (s < NB) issue(s);
    for (int s = 0; s < NB; ++s) {
        const int young = min(PF_NS - 2, NB - 1 - s);      // younger stages in flight behind stage s, 6 pieces of this wave each
        if (young == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + PF_NS - 1 < NB) issue(s + PF_NS - 1);
        const unsigned sb = smem_lds + (unsigned)(s % PF_NS) * PF_STAGE;
        tg_u32x4 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(b[t]) : "v"(sb + (w_lane ^ ((unsigned)t << 5))));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            tg_u32x4 ah[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[m]) : "v"(sb + a_lane), "n"((m * 4 + t) * 1024));
            // (LDS returns in order: behind the four B reads of the stage come MT reads per step)
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[t]) : "n"(MT));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ah[m]) : "n"(MT - 1 - m));
                acc[m] = tg_mfma(ah[m], b[t], acc[m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + mfma32_row(r, lane);
            if (row < M && k0 + l31 < Kd) dx[(int64_t)row * lddx + k0 + l31] = acc[m][r];
        }
}

}  // namespace lpm

extern "C" int lpm_proj_supported(int M, int64_t Kd, int N) {
    return (M > 0 && M <= 128 && N > 0 && N % 512 == 0 && Kd > 0 && Kd % 16 == 0 && Kd / 16 >= 8) ? 1 : 0;
}
extern "C" size_t lpm_proj_fwd_workspace_bytes(int M, int64_t Kd, int N) {
    return (size_t)lpm::proj_splits(Kd, N) * M * N * sizeof(float);
}

static int proj_fwd_impl(const float* x, int64_t ldx, const lpm::ProjParts* parts, const float* W, int M, int64_t Kd, int N, float* y,
                         void* workspace, size_t workspace_bytes, lpm_stream_t stream, bool w16 = false);
extern "C" int lpm_proj_fwd(const float* x, int64_t ldx, const float* W, int M, int64_t Kd, int N, float* y, void* workspace,
                            size_t workspace_bytes, lpm_stream_t stream) {
    return proj_fwd_impl(x, ldx, nullptr, W, M, Kd, N, y, workspace, workspace_bytes, stream);
}
// y = [x1 * scale | x2] . W without the concatenation: x1 [M, n1a] (row stride ldx1; n1a a multiple of 32) is multiplied by
// scale[row][column % ks] where it is read (ks a multiple of 4: the lazily normalised d-major descriptor -- the un-normalised residual
// sums [M, D, K] and lpm_vlad_row_scales' [M, K]; x1_bf16: stored as bf16, ldx1 in elements); x2 [M, Kd - n1a] (row stride ldx2; may be
// NULL when n1a == Kd) enters as it is.
extern "C" int lpm_proj_fwd_parts(const void* x1, int64_t ldx1, int64_t n1a, int x1_bf16, const float* scale, int ks, const float* x2,
                                  int64_t ldx2, const float* W, int M, int64_t Kd, int N, float* y, void* workspace, size_t workspace_bytes,
                                  lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x1 && scale && (x2 || n1a == Kd), LPM_ERR_BADARG, "lpm_proj_fwd_parts: null pointer");
    LPM_REQUIRE(n1a > 0 && n1a <= Kd && n1a % 32 == 0 && ks > 0 && ks % 4 == 0 && n1a % ks == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_proj_fwd_parts: the scaled block must be a multiple of 32 columns and of ks (n1a=%lld ks=%d)", (long long)n1a, ks);
    LPM_REQUIRE(ldx1 >= n1a && ldx1 % 4 == 0 && (n1a == Kd || (ldx2 >= Kd - n1a && ldx2 % 4 == 0)), LPM_ERR_BADARG,
                "lpm_proj_fwd_parts: row strides must cover their block and be multiples of 4");
    LPM_REQUIRE((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)scale) & 15) == 0, LPM_ERR_BADARG, "lpm_proj_fwd_parts: pointers must be 16-byte aligned");
    const ProjParts pp{x2, ldx2, n1a, scale, ks, x1_bf16 ? 1 : 0};
    return proj_fwd_impl((const float*)x1, ldx1, &pp, W, M, Kd, N, y, workspace, workspace_bytes, stream);
}
// ... from the bf16 compute copy W16 [Kd, N] of the weight (lpm_factored_clip_adam_copy keeps it): x1 must be the bf16-stored block
extern "C" int lpm_proj_fwd_parts_w16(const void* x1, int64_t ldx1, int64_t n1a, const float* scale, int ks, const float* x2, int64_t ldx2,
                                      const void* W16, int M, int64_t Kd, int N, float* y, void* workspace, size_t workspace_bytes,
                                      lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x1 && scale && (x2 || n1a == Kd), LPM_ERR_BADARG, "lpm_proj_fwd_parts_w16: null pointer");
    LPM_REQUIRE(n1a > 0 && n1a <= Kd && n1a % 32 == 0 && ks > 0 && ks % 4 == 0 && n1a % ks == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_proj_fwd_parts_w16: the scaled block must be a multiple of 32 columns and of ks (n1a=%lld ks=%d)", (long long)n1a, ks);
    LPM_REQUIRE(ldx1 >= n1a && ldx1 % 4 == 0 && (n1a == Kd || (ldx2 >= Kd - n1a && ldx2 % 4 == 0)), LPM_ERR_BADARG,
                "lpm_proj_fwd_parts_w16: row strides must cover their block and be multiples of 4");
    LPM_REQUIRE((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)scale) & 15) == 0, LPM_ERR_BADARG, "lpm_proj_fwd_parts_w16: pointers must be 16-byte aligned");
    const ProjParts pp{x2, ldx2, n1a, scale, ks, 1};
    return proj_fwd_impl((const float*)x1, ldx1, &pp, (const float*)W16, M, Kd, N, y, workspace, workspace_bytes, stream, true);
}
static int proj_fwd_impl(const float* x, int64_t ldx, const lpm::ProjParts* parts, const float* W, int M, int64_t Kd, int N, float* y,
                         void* workspace, size_t workspace_bytes, lpm_stream_t stream, bool w16) {
    using namespace lpm;
    LPM_REQUIRE(x && W && y && workspace, LPM_ERR_BADARG, "lpm_proj_fwd: null pointer");
    LPM_REQUIRE(lpm_proj_supported(M, Kd, N), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_proj_fwd: need M <= 128, N %% 512 == 0, Kd %% 16 == 0 (M=%d Kd=%lld N=%d)", M, (long long)Kd, N);
    LPM_REQUIRE(workspace_bytes >= lpm_proj_fwd_workspace_bytes(M, Kd, N), LPM_ERR_WORKSPACE, "lpm_proj_fwd: workspace too small");
    LPM_REQUIRE(parts || (ldx >= Kd && ldx % 4 == 0), LPM_ERR_BADARG, "lpm_proj_fwd: the row stride of x must be >= Kd and a multiple of 4");
    LPM_REQUIRE((((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)workspace) & 15) == 0, LPM_ERR_BADARG,
                "lpm_proj_fwd: pointers must be 16-byte aligned");
    const int splits = proj_splits(Kd, N), nslab = (int)(Kd / 16), MT = (M + 31) / 32;
    dim3 grid(splits, N / 512);
    hipStream_t s = (hipStream_t)stream;
#define LPM_PJ(MTV, NPV, XDV)                                                                                                \
    do {                                                                                                                     \
        const size_t lds = (size_t)PJ_NS * PJ_WBYTES + XDV * NPV * 1024;                                                     \
        auto kern = !parts ? proj_fwd_kernel<MTV, NPV, XDV, 0>                                                               \
                           : (w16 ? proj_fwd_kernel<MTV, NPV, XDV, 2, true>                                                  \
                                  : (parts->x1_bf16 ? proj_fwd_kernel<MTV, NPV, XDV, 2> : proj_fwd_kernel<MTV, NPV, XDV, 1>)); \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {    \
            (void)hipGetLastError();                                                                                         \
            set_error("lpm_proj_fwd: cannot reserve %zu bytes of LDS", lds);                                                 \
            return LPM_ERR_LAUNCH;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(512 + 64 * pj_loaders(NPV, parts ? 1 : 0)), lds, s, x, ldx, W, M, Kd, N, nslab,  \
                           splits, (float*)workspace, parts ? *parts : ProjParts{});                                         \
    } while (0)
    /* three pair-buffers of x where they fit beside the four weight stages (measured: no gain over two once the ring is read without the */
    /* compiler's waits -- kept where it is free) */
    switch (MT) {
        case 1: LPM_PJ(1, 4, 3); break;
        case 2: LPM_PJ(2, 8, 3); break;
        case 3: if (M <= 80) LPM_PJ(3, 10, 3); else LPM_PJ(3, 12, 2); break;
        default: LPM_PJ(4, 16, 2); break;
    }
#undef LPM_PJ
    const int64_t n4 = (int64_t)M * N / 4;
    hipLaunchKernelGGL(proj_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, s, (const float4*)workspace, splits, n4, (float4*)y);
    return check_launch("lpm_proj_fwd");
}

// dx = dy . W^T from the bf16 compute copy W16 [Kd, N]; dyt as for lpm_proj_dx (only its hi plane is read); N a multiple of 64
extern "C" int lpm_proj_dx_w16(const void* dyt, const void* W16, int M, int64_t Kd, int N, float* dx, int64_t lddx, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dyt && W16 && dx, LPM_ERR_BADARG, "lpm_proj_dx_w16: null pointer");
    LPM_REQUIRE(M > 0 && M <= 128 && N > 0 && N % 64 == 0 && Kd > 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_proj_dx_w16: need M <= 128, N %% 64 == 0 (M=%d N=%d)", M, N);
    LPM_REQUIRE((((uintptr_t)dyt | (uintptr_t)W16 | (uintptr_t)dx) & 15) == 0, LPM_ERR_BADARG, "lpm_proj_dx_w16: pointers must be 16-byte aligned");
    LPM_REQUIRE(lddx >= Kd, LPM_ERR_BADARG, "lpm_proj_dx_w16: the row stride of dx must be >= Kd");
    const int MT = (M + 31) / 32;
    dim3 grid((unsigned)((Kd + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
#define LPM_PF(MTV)                                                                                                          \
    do {                                                                                                                     \
        auto kern = proj_dx16_kernel<MTV>;                                                                                   \
        const size_t lds = (size_t)PF_NS * PF_STAGE;                                                                         \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {    \
            (void)hipGetLastError();                                                                                         \
            set_error("lpm_proj_dx_w16: cannot reserve %zu bytes of LDS", lds);                                              \
            return LPM_ERR_LAUNCH;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const uint4*)dyt, (const unsigned short*)W16, M, Kd, N, dx, lddx); \
    } while (0)
    switch (MT) {
        case 1: LPM_PF(1); break;
        case 2: LPM_PF(2); break;
        case 3: LPM_PF(3); break;
        default: LPM_PF(4); break;
    }
#undef LPM_PF
    return check_launch("lpm_proj_dx_w16");
}

// dyt: lpm_split_rows_tiles(dy, ldx = N, B = 1, T = M, C = N) -- split-bf16 row tiles of the [M, N] gradient
extern "C" int lpm_proj_dx(const void* dyt, const float* W, int M, int64_t Kd, int N, float* dx, int64_t lddx, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dyt && W && dx, LPM_ERR_BADARG, "lpm_proj_dx: null pointer");
    LPM_REQUIRE(M > 0 && M <= 128 && N > 0 && N % 16 == 0 && Kd > 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_proj_dx: need M <= 128, N %% 16 == 0 (M=%d N=%d)", M, N);
    LPM_REQUIRE((((uintptr_t)dyt | (uintptr_t)W | (uintptr_t)dx) & 15) == 0, LPM_ERR_BADARG, "lpm_proj_dx: pointers must be 16-byte aligned");
    LPM_REQUIRE(lddx >= Kd, LPM_ERR_BADARG, "lpm_proj_dx: the row stride of dx must be >= Kd");
    const int MT = (M + 31) / 32;
    dim3 grid((unsigned)((Kd + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    // second form (whole 128-byte lines of W per stage): N a multiple of 32; LPM_PROJ_DX_FORM=1 selects the first form (A/B)
    static const int form = [] { const char* e = getenv("LPM_PROJ_DX_FORM"); return (e && e[0] == '1') ? 1 : 2; }();
    if (form == 2 && N % 32 == 0 && N >= 64) {
#define LPM_PE(MTV)                                                                                                          \
    do {                                                                                                                     \
        auto kern = proj_dx2_kernel<MTV>;                                                                                    \
        const size_t lds = (size_t)PE_NS * PE_STAGE;                                                                         \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {    \
            (void)hipGetLastError();                                                                                         \
            set_error("lpm_proj_dx: cannot reserve %zu bytes of LDS", lds);                                                  \
            return LPM_ERR_LAUNCH;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const uint4*)dyt, W, M, Kd, N, dx, lddx);                         \
    } while (0)
        switch (MT) {
            case 1: LPM_PE(1); break;
            case 2: LPM_PE(2); break;
            case 3: LPM_PE(3); break;
            default: LPM_PE(4); break;
        }
#undef LPM_PE
        return check_launch("lpm_proj_dx");
    }
#define LPM_PD(MTV)                                                                                                          \
    do {                                                                                                                     \
        auto kern = proj_dx_kernel<MTV>;                                                                                     \
        const size_t lds = (size_t)PD_NS * pd_stage<MTV>();                                                                  \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {    \
            (void)hipGetLastError();                                                                                         \
            set_error("lpm_proj_dx: cannot reserve %zu bytes of LDS", lds);                                                  \
            return LPM_ERR_LAUNCH;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const uint4*)dyt, W, M, Kd, N, dx, lddx);                         \
    } while (0)
    switch (MT) {
        case 1: LPM_PD(1); break;
        case 2: LPM_PD(2); break;
        case 3: LPM_PD(3); break;
        default: LPM_PD(4); break;
    }
#undef LPM_PD
    return check_launch("lpm_proj_dx");
}
