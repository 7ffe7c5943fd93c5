// K1 (soft-assignment GEMM) and its backward on the bf16 matrix pipe, as instances of the split-bf16 tile GEMM
// (tile_gemm.h).  Reference: frame_level_models.py:2781-2789 (tf.matmul + the batch statistics of cluster_bn); the
// backward is TF autodiff of that matmul.
//   forward   logits[B*T, K] = x . W           A = row tiles of x (per clip), B = weight tiles of W, + BN-stat epilogue
//   dx       += dlogits . W^T                   A = row tiles of dlogits,      B = weight tiles of W^T (reduction over K)
//   dW        = x^T . dlogits                   A = frame tiles of x,          B = frame tiles of dlogits: the reduction
//                                               runs over every frame of every clip, split over workgroups + a reduce pass
#include "tile_gemm.h"

namespace lpm {

__host__ __device__ constexpr int tg_ring_bytes(int ntw) { return TG_NS * (4 + 8 * ntw) * 1024; }
__host__ __device__ constexpr int tg_epi_stride(int ntw) { return 128 * ntw + 1; }
static size_t tg_lds_bytes(int ntw, int epi) {
    const size_t ring = tg_ring_bytes(ntw);
    const size_t e = epi == TG_EPI_SOFTMAX_BWD ? (size_t)64 * tg_epi_stride(ntw) * sizeof(float) : 0;
    return ring > e ? ring : e;
}

// MW = 1: 256 threads, 64 rows per workgroup, blockIdx.x = (batch, 64-row block).
// MW = 2 (STORE only): 512 threads, 128 rows per workgroup = two row groups of four waves each; the row tiles of all
// batches are ONE flat sequence (a_batch == a_tiles * a_tile, b_batch == 0, no second operand pair, one split) and
// blockIdx.x counts groups of four consecutive tiles, which may straddle two batches.  24 KB staged per 128 x 256 x 16
// products (131 MFMA-flop per byte against 77 for the 64-row form), NS-stage ring with NS - 2 younger steps in flight.
// RTW = 4 (MW = 2, STORE only): 256 rows per workgroup -- every wave owns FOUR row tiles x NTW column tiles (128 x 64 accumulators,
// 128 registers): 32 KB staged per 256 x 256 x 16 products (262 MFMA-flop per byte) and 12 KB of fragment reads per 24 MFMAs and
// wave instead of 8 KB per 12 -- the LDS read volume is what holds the 128-row form at ~1.1 PFLOP/s on the encoder's dense shapes.
// FORM 1 (MW = 2, round 4): the two wave groups of a workgroup run in ANTI-PHASE -- see the branch below.
// DBG 1: the measurement switches of TileGemmArgs::dbg are live (LPM_TG_DBG; a separate instantiation: the production kernels carry no
// such branches -- as run-time tests around the MFMA groups they cost K1 30 %: 42.5 -> 55.5 us, measured in round 4)
template <int NTW, int EPI, int MW, int NS, int PL, int RTW = 2, int FORM = 0, int DBG = 0>
__global__ __launch_bounds__(256 * MW, (MW == 2 && NS == 3 && FORM == 0) ? 4 : 2) void tile_gemm_kernel(const TileGemmArgs g) {
    const int dbg = DBG ? g.dbg : 0;
    static_assert(FORM == 0 || (MW == 2 && EPI == TG_EPI_STORE), "the anti-phase form: 128- / 256-row workgroups, store epilogue");
    static_assert(RTW == 2 || (RTW == 4 && MW == 2 && EPI == TG_EPI_STORE && NS >= 4), "four row tiles per wave: pipelined 128-row form only");
    static_assert(MW == 1 || EPI == TG_EPI_STORE, "the 128-row form has the store epilogue only");
    static_assert(EPI != TG_EPI_ADAM || (NTW == 1 && PL == 2), "the Adam epilogue: 64 x 128 tiles of split-bf16 operands");
    static_assert(PL >= 1 && PL <= 4, "planes");
    static_assert(PL < 3 || (MW == 2 && RTW == 4 && EPI == TG_EPI_STORE && FORM == 0 && NS >= 4), "the fp16 forms: pipelined 256-row workgroups");
    // PL == 1: plain bf16 tiles, a ring stage = two reduction steps ("sub" below is the step within the stage where the split
    // form has the plane); no second operand pair.
    // PL == 3 (round 5): fp16 operands, A = (hi, lo) planes, B = the hi plane only (64 units per (tile, step)): a . b = ah bh + al bh.
    // PL == 4: fp16 operands, both (hi, lo): the three terms of the split form on the fp16 pipe.
    constexpr bool SPLIT = PL == 2 || PL == 4;     // both operands carry (hi, lo) planes, three terms per product
    constexpr int PLB = PL == 3 ? 1 : 2;           // pieces per column tile and stage
    constexpr int NTB = 4 * NTW;                   // column tiles per workgroup
    constexpr int NWV = 4 * MW;                    // waves per workgroup
    constexpr int NRP = 2 * RTW * MW;              // row-tile pieces per stage (RTW MW tiles x 2 planes)
    constexpr int NP = NRP + PLB * NTB;            // 1 KB pieces per stage
    constexpr int PW = NP / NWV;                   // pieces per wave per stage
    static_assert(NP % NWV == 0, "pieces must divide over the waves");
    constexpr int STAGE = NP * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave >> 2, cw = wave & 3;       // row group, column slice
    const int l31 = lane & 31;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    int cb = blockIdx.y;
    if (MW == 1 && g.cols_inner > 0) {             // one-dimensional grid, column block fastest
        cb = lid % g.cols_inner;
        lid /= g.cols_inner;
    }
    const int batch = MW == 1 ? lid / g.rb_per_batch : 0, rb = MW == 1 ? lid % g.rb_per_batch : 0;
    const int split = blockIdx.z;
    const int step0 = split * g.steps_per_split;
    const int nred = min(g.steps_per_split, g.total_steps - step0);     // reduction steps of this split
    const int nstep1 = PL >= 2 ? nred : (nred + 1) / 2;                   // ring stages (the second operand pair: split-bf16 only)
    const int nstep = nstep1 + (PL == 2 ? g.steps2 : 0);

    // this wave's PW pieces of a stage: piece p < 4: row tile p>>1, plane p&1;  else column tile (p-4)>>1, plane (p-4)&1
    const uint4* src[PW];
    const uint4* src2[PW];
    int64_t sstep[PW], sstep2[PW];
    int sub[PW];                                   // PL == 1: which of the stage's two steps this piece is
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int p = wave + NWV * j;
        // split form: the piece's plane sits (p & 1) * 64 units into the (tile, step) pair; plain form: same tile, step + (p & 1)
        const int pl = (SPLIT || (PL == 3 && p < NRP)) ? (p & 1) * 64 : 0;
        sub[j] = p & 1;
        if (p < NRP) {
            if (MW == 1) {
                const int t = min(rb * 2 + (p >> 1), g.a_tiles - 1);
                src[j] = g.a + batch * g.a_batch + t * g.a_tile + step0 * g.a_step + pl + lane;
                src2[j] = PL == 2 ? g.a2 + batch * g.a2_batch + t * g.a2_tile + pl + lane : nullptr;
            } else if (g.a_img_row) {          // A from an operand image: this lane's row, its 16-byte half of the step
                src[j] = g.a + ((int64_t)(lid * RTW * MW + (p >> 1)) * 32 + l31) * g.a_img_row + step0 * g.a_step + (p & 1) * g.a_img_lo + (lane >> 5);
                src2[j] = nullptr;
            } else {
                src[j] = g.a + (int64_t)(lid * RTW * MW + (p >> 1)) * g.a_tile + step0 * g.a_step + pl + lane;
                src2[j] = nullptr;
            }
            sstep[j] = g.a_step;
            sstep2[j] = g.a2_step;
        } else {
            const int ct = cb * NTB + (PL == 3 ? (p - NRP) : ((p - NRP) >> 1));
            src[j] = g.b + batch * g.b_batch + min(ct, g.b_tiles - 1) * g.b_tile + step0 * g.b_step + pl + lane;
            sstep[j] = g.b_step;
            src2[j] = PL == 2 ? g.b2 + batch * g.b2_batch + min(ct, g.b2_tiles - 1) * g.b2_tile + pl + lane : nullptr;
            sstep2[j] = g.b2_step;
        }
    }
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % NS) * STAGE;
        const bool second = PL == 2 && MW == 1 && s >= nstep1;       // wave-uniform
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            // plain form: stage s = steps 2 s and 2 s + 1; past an odd end the piece re-reads the last step (its MFMA is skipped)
            const uint4* q = PL == 1 ? src[j] + (int64_t)min(2 * s + sub[j], nred - 1) * sstep[j]
                                     : ((PL == 2 && second) ? src2[j] + (s - nstep1) * sstep2[j] : src[j] + s * sstep[j]);
            // (default cache policy: the non-temporal policy was measured on both operands, alone and together -- LPM_TG_NT, round 4 --
            // and changed nothing for K1 and the wide dense shapes, +-5 % either way on the N = 1024 shapes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                             (__attribute__((address_space(3))) void*)(st + (wave + NWV * j) * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[RTW][NTW];
#pragma unroll
    for (int m = 0; m < RTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ADAM epilogue: this thread's 8 float4 of param / m / v (the epilogue's (row, column) mapping) are requested BEFORE the
    // reduction, so that their HBM round trip runs under it: the kernel is a 24 B-per-parameter stream with a tile GEMM inside,
    // and a workgroup that asked for its 96 KB only after its last MFMA was bound by that latency (0.75 ms at cfg-2 instead of ~0.5).
    f32x4 pa[EPI == TG_EPI_ADAM ? 8 : 1], ma[EPI == TG_EPI_ADAM ? 8 : 1], va[EPI == TG_EPI_ADAM ? 8 : 1];
    if constexpr (EPI == TG_EPI_ADAM) {
        const int gcol = cb * 128 + (tid & 31) * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int grow = rb * 64 + (j >> 2) * 32 + (tid >> 5) + 8 * (j & 3);
            const bool ok = grow < g.rows_valid && gcol < g.cols_valid;
            const int64_t o = ok ? (int64_t)grow * g.ldo + gcol : 0;
            pa[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.adam_p + o));
            ma[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.adam_m + o));
            va[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.adam_v + o));
        }
    }
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nstep) issue(s);
    if constexpr (FORM == 1) {
        // ANTI-PHASE wave groups (round 4).  Measured on the pipelined form below (LPM_TG_DBG ablations, K1 and the 256-row dense form):
        // its time is the SUM of its matrix time, its fragment-read time and its DMA-issue time -- all eight waves are released by the
        // same barrier, issue their LDS-DMA pieces and fragment reads at the same moment (an LDS burst of 64-96 KB, then silence) and
        // the matrix pipe idles meanwhile: the two waves of a SIMD are in the same state.  Here the workgroup's two row groups (waves
        // 0-3 / 4-7: wave w and wave w + 4 share a SIMD) are half a step apart: a step of a wave is a COMPUTE phase (its MFMAs, nothing
        // else) and a LOAD phase (DMA issue for step t + NS, fragment reads of step t + 1 into the SAME registers -- the MFMAs that read
        // them have been issued --, wait), with a barrier behind each; while one group computes the other loads (cdna guide 5.5
        // T3+T4: the phase interleave is the lever).  One fragment register set (the pipelined form holds two).
        //   phase 2t: group 0 C(t), group 1 L(t)   |   phase 2t + 1: group 0 L(t + 1), group 1 C(t)
        // Step k is read in phases 2k - 1 (group 0) and 2k (group 1): every wave has waited for its own pieces of step k before the
        // barrier that ends phase 2k - 2 -- group 0 at the end of C(k - 1), group 1 at the end of L(k - 1).  L(t + 1) refills stage
        // t % NS (step t + NS): both groups have read step t by then (phases 2t - 1, 2t).  All NS stages are requested up front.
        struct Frag { tg_u32x4 ah[RTW], al[RTW], bh[NTW], bl[NTW]; };
        Frag fr;
        auto read_frags = [&](int st_, Frag& f_) {
            const tg_u32x4* f = reinterpret_cast<const tg_u32x4*>(smem + (st_ % NS) * STAGE) + lane;
#pragma unroll
            for (int m = 0; m < RTW; ++m) {
                f_.ah[m] = f[((rg * RTW + m) * 2 + 0) * 64];
                f_.al[m] = f[((rg * RTW + m) * 2 + 1) * 64];
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                f_.bh[n] = f[(NRP + (cw * NTW + n) * 2 + 0) * 64];
                f_.bl[n] = f[(NRP + (cw * NTW + n) * 2 + 1) * 64];
            }
        };
        auto mfma_term = [&](int t, int sstep_) {
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int m = 0; m < RTW; ++m) {
                    if (PL == 2) acc[m][n] = tg_mfma(t == 2 ? fr.al[m] : fr.ah[m], t == 1 ? fr.bl[n] : fr.bh[n], acc[m][n]);
                    else acc[m][n] = tg_mfma(t ? fr.al[m] : fr.ah[m], t ? fr.bl[n] : fr.bh[n], acc[m][n]);
                }
            (void)sstep_;
        };
        int hi = min(NS - 2, nstep - 1);               // the youngest step requested so far (the common prologue above: 0 .. NS - 2)
        if (NS - 1 < nstep) { issue(NS - 1); hi = NS - 1; }
        auto wait_for = [&](int k) {                   // this wave's pieces of step k have landed (hi - k younger steps stay in flight)
            if (k >= nstep) return;
            const int y = hi - k;
            if (y >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PW) : "memory");
            else if (y == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
            else if (y == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0) (as a builtin: the compiler's counter bookkeeping sees it)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        wait_for(0);
        sync();
        read_frags(0, fr);
        if (rg == 1) {                                 // group 1's idle phase 0 (group 0 computes step 0)
            wait_for(1);
            sync();
        }
        for (int t = 0; t < nstep; ++t) {
            // C(t)
            __builtin_amdgcn_sched_barrier(0);
            mfma_term(0, t);
            if (PL == 2 || 2 * t + 1 < nred) mfma_term(1, t);
            if (PL == 2) mfma_term(2, t);
            __builtin_amdgcn_sched_barrier(0);
            if (rg == 0) wait_for(t + 1);
            if (rg == 0 || t + 1 < nstep) sync();
            if (t + 1 < nstep) {
                // L(t + 1)
                if (t + NS < nstep) { issue(t + NS); hi = t + NS; }
                read_frags(t + 1, fr);
                if (rg == 1) wait_for(t + 2);
                sync();
            }
        }
    } else if constexpr (MW == 2 && NS == 3) {
        // 128-row form, TWO workgroups per CU (72 KB ring, <= 128 VGPRs): no software pipelining inside a workgroup -- the other
        // workgroup's MFMAs cover this one's barrier, DMA issue and fragment reads, and its epilogue runs under this one's main loop
        for (int s = 0; s < nstep; ++s) {
            if (s + 1 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");   // one younger step in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + NS - 1 < nstep) issue(s + NS - 1);
            const tg_u32x4* f = reinterpret_cast<const tg_u32x4*>(smem + (s % NS) * STAGE) + lane;
            tg_u32x4 ah[2], al[2], bh[NTW], bl[NTW];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = f[((rg * 2 + m) * 2 + 0) * 64];
                al[m] = f[((rg * 2 + m) * 2 + 1) * 64];
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                bh[n] = f[(NRP + (cw * NTW + n) * 2 + 0) * 64];
                bl[n] = f[(NRP + (cw * NTW + n) * 2 + 1) * 64];
            }
#pragma unroll
            for (int t = 0; t < (PL == 2 ? 3 : 2); ++t) {
                if (PL == 1 && t == 1 && !(2 * s + 1 < nred)) break;
#pragma unroll
                for (int n = 0; n < NTW; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (PL == 2) acc[m][n] = tg_mfma(t == 2 ? al[m] : ah[m], t == 1 ? bl[n] : bh[n], acc[m][n]);
                        else acc[m][n] = tg_mfma(t ? al[m] : ah[m], t ? bl[n] : bh[n], acc[m][n]);
                    }
            }
        }
    } else if constexpr (MW == 1) {
        for (int s = 0; s < nstep; ++s) {
            if (s + 1 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");   // one younger step in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // step s is in LDS for everyone; stage (s-1) % NS is free again
            asm volatile("" ::: "memory");
            if (s + NS - 1 < nstep) issue(s + NS - 1);
            const tg_u32x4* f = reinterpret_cast<const tg_u32x4*>(smem + (s % NS) * STAGE) + lane;
            tg_u32x4 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = f[(m * 2 + 0) * 64];
                al[m] = f[(m * 2 + 1) * 64];
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int nt = wave * NTW + n;
                const tg_u32x4 bh = f[(4 + nt * 2 + 0) * 64], bl = f[(4 + nt * 2 + 1) * 64];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (PL == 2) {
                        acc[m][n] = tg_mfma(ah[m], bh, acc[m][n]);
                        acc[m][n] = tg_mfma(ah[m], bl, acc[m][n]);
                        acc[m][n] = tg_mfma(al[m], bh, acc[m][n]);
                    } else {          // (ah, bh) = first step of the stage, (al, bl) = second
                        acc[m][n] = tg_mfma(ah[m], bh, acc[m][n]);
                        if (2 * s + 1 < nred) acc[m][n] = tg_mfma(al[m], bl, acc[m][n]);
                    }
                }
            }
        }
    } else {
        // Software-pipelined: the fragments of step s + 1 are read from LDS while the matrix pipe works on step s (two
        // register sets, the loop unrolled by two), so the only wait inside a step is the barrier itself.  The barrier
        // at the top of step s says: step s + 1 has landed for everyone, everyone holds the fragments of step s in
        // registers, everyone is done with step s - 1 -> stage (s - 1) % NS takes step s + NS - 1.  NS - 2 steps are in
        // flight behind the one being read.  (nstep >= NS is the launcher's condition.)
        static_assert(NS >= 4 && NS <= 6, "ring depth of the pipelined 128-row form");
        struct Frag { tg_u32x4 ah[RTW], al[RTW], bh[NTW], bl[PL == 3 ? 1 : NTW]; };
        auto read_frags = [&](int s, Frag& fr) {
            const tg_u32x4* f = reinterpret_cast<const tg_u32x4*>(smem + (s % NS) * STAGE) + lane;
#pragma unroll
            for (int m = 0; m < RTW; ++m) {
                fr.ah[m] = f[((rg * RTW + m) * 2 + 0) * 64];
                fr.al[m] = f[((rg * RTW + m) * 2 + 1) * 64];
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                if constexpr (PL == 3) {
                    fr.bh[n] = f[(NRP + (cw * NTW + n)) * 64];
                } else {
                    fr.bh[n] = f[(NRP + (cw * NTW + n) * 2 + 0) * 64];
                    fr.bl[n] = f[(NRP + (cw * NTW + n) * 2 + 1) * 64];
                }
            }
        };
        auto body = [&](int s, Frag& cur, Frag& nxt) {
            if (s + 1 < nstep) {                   // step s + 1 must have landed; younger steps issued: min(NS - 3, nstep - 2 - s)
                if (NS >= 6 && s + 4 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PW) : "memory");
                else if (NS >= 5 && s + 3 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
                else if (s + 2 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // the fragments of step s, read during step s - 1, are complete (as a builtin, so that the compiler's own
            // counter bookkeeping sees it and does not wait for the reads of step s + 1 in front of the MFMAs)
            __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // The matrix pipe gets work straight after the barrier; the DMA issue (address arithmetic + NS-stage ring
            // writes) and the LDS reads of the next fragments go into the shadow of MFMAs already queued -- with all
            // eight waves released together, a wave that first issued its DMA pieces and reads left the pipe idle for
            // a few hundred cycles per step.  term-major order: four different accumulators between dependent MFMAs.
            auto mfma_term = [&](int t) {
#pragma unroll
                for (int n = 0; n < NTW; ++n)
#pragma unroll
                    for (int m = 0; m < RTW; ++m) {
                        if constexpr (PL == 3) acc[m][n] = tg_mfma_f16(t ? cur.al[m] : cur.ah[m], cur.bh[n], acc[m][n]);
                        else if constexpr (PL == 4) acc[m][n] = tg_mfma_f16(t == 2 ? cur.al[m] : cur.ah[m], t == 1 ? cur.bl[n] : cur.bh[n], acc[m][n]);
                        else if (PL == 2) acc[m][n] = tg_mfma(t == 2 ? cur.al[m] : cur.ah[m], t == 1 ? cur.bl[n] : cur.bh[n], acc[m][n]);
                        else acc[m][n] = tg_mfma(t ? cur.al[m] : cur.ah[m], t ? cur.bl[PL == 3 ? 0 : n] : cur.bh[n], acc[m][n]);
                    }
            };
            if (!(dbg & 8)) mfma_term(0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + NS - 1 < nstep && !(dbg & 4)) issue(s + NS - 1);
            __builtin_amdgcn_sched_barrier(0);
            if ((PL >= 2 || 2 * s + 1 < nred) && !(dbg & 8)) mfma_term(1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < nstep) read_frags(s + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT && !(dbg & 8)) mfma_term(2);
        };
        Frag fa, fb;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PW) : "memory");      // step 0 has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(0, fa);
        const int nrun = (dbg & 1) ? 0 : nstep;        // (measurement: LPM_TG_DBG, see the launcher)
        for (int s = 0; s < nrun; s += 2) {
            body(s, fa, fb);
            if (s + 1 < nstep) body(s + 1, fb, fa);
        }
        if (dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    const int N = g.cols_valid;
    if (EPI == TG_EPI_STORE) {
        // column statistics from the registers; the tile itself goes through LDS (the ring is free) so that every global
        // access is a 16-byte piece of a contiguous row segment: 32 rows x 128*NTW columns per pass
        constexpr int ESTR = 128 * NTW + 4;
        constexpr int C4 = 32 * NTW;
        float* es = reinterpret_cast<float*>(smem) + rg * 32 * ESTR;      // one staging tile per row group
        float* ob = g.out + batch * g.out_batch + split * g.out_split;
        if (RTW == 2 && g.stats) {
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int col = (cb * NTB + cw * NTW + n) * 32 + l31;
                float cs = 0.f, cq = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[m][n][r];
                        cs += v;               // statistics callers pad with zero row tiles: rows >= rows_valid add nothing
                        cq = fmaf(v, v, cq);
                    }
                cs += __shfl_xor(cs, 32, 64);
                cq += __shfl_xor(cq, 32, 64);
                if (lane < 32 && col < N) {
                    float* p = g.stats + (int64_t)(lid * MW + rg) * 2 * N;    // one statistics row per 64-row group
                    p[col] = cs;
                    p[N + col] = cq;
                }
            }
        }
        __syncthreads();                       // nothing in flight (the last step waited for vmcnt(0)): the ring is free
        const int tl = tid & 255;
        if constexpr (RTW == 4 || (MW == 2 && NS == 3)) {
            if (g.img) {
                // image epilogue: thread = 8 fixed columns (c8) x rows (tl >> 5) + 8 k of every 32-row pass; 16-byte stores per plane
                const int c8 = (tl & 31) * 8, rbase = tl >> 5;
                const int gcol = cb * NTB * 32 + c8;
                const bool colok = gcol < N;
                float bias8[8], colacc[8];
                float vmax = 0.f;                                  // max |v| of what this thread writes (img_amax)
                const float alpha = g.alpha;
                const int64_t istride = of_row_stride(N, g.img_planes), mstride = of_row_stride(N, g.mask_planes);
#pragma unroll
                for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; colacc[e] = 0.f; }
                if (g.img_kind == 1 && colok) {
                    const float4 b0 = *reinterpret_cast<const float4*>(g.img_bias + gcol), b1 = *reinterpret_cast<const float4*>(g.img_bias + gcol + 4);
                    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
                }
#pragma unroll
                for (int m = 0; m < RTW; ++m) {
                    if (m) __syncthreads();
#pragma unroll
                    for (int n = 0; n < NTW; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) es[mfma32_row(r, lane) * ESTR + (cw * NTW + n) * 32 + l31] = acc[m][n][r];
                    __syncthreads();
                    const int ft = lid * RTW * MW + rg * RTW + m;
                    const int tb = ft / g.a_tiles;
                    const int row0 = (ft - tb * g.a_tiles) * 32;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = rbase + 8 * k, grow = row0 + row;
                        if (grow < g.rows_valid && colok) {
                            const int64_t R = (int64_t)tb * g.rows_valid + grow;
                            const float4 a0 = *reinterpret_cast<const float4*>(es + row * ESTR + c8);
                            const float4 a1 = *reinterpret_cast<const float4*>(es + row * ESTR + c8 + 4);
                            float v[8] = {a0.x * alpha, a0.y * alpha, a0.z * alpha, a0.w * alpha, a1.x * alpha, a1.y * alpha, a1.z * alpha, a1.w * alpha};
                            if (g.img_kind == 1) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + bias8[e], 0.f);
                            } else {
                                const uint4 hm = *reinterpret_cast<const uint4*>(g.img_mask + R * mstride + gcol);      // hi plane of the activation
                                const unsigned mw[4] = {hm.x, hm.y, hm.z, hm.w};
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const unsigned ah = (mw[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                                    v[e] = of_positive(ah) ? v[e] : 0.f;                            // activation > 0
                                    colacc[e] += v[e];
                                }
                            }
                            vmax = of_amax8(vmax, v);
                            uint4 hi, lo;
                            of_split8(v, g.img_f16, g.img_scale, hi, lo);
                            unsigned short* rowp = g.img + R * istride + gcol;
                            const tg_u32x4 hv = {hi.x, hi.y, hi.z, hi.w}, lv = {lo.x, lo.y, lo.z, lo.w};
                            if (g.img_planes == 2) {
                                if (g.nt_store) {
                                    __builtin_nontemporal_store(hv, reinterpret_cast<tg_u32x4*>(rowp));
                                    __builtin_nontemporal_store(lv, reinterpret_cast<tg_u32x4*>(rowp + N));
                                } else {
                                    *reinterpret_cast<tg_u32x4*>(rowp) = hv;
                                    *reinterpret_cast<tg_u32x4*>(rowp + N) = lv;
                                }
                            } else if (g.nt_store) {
                                __builtin_nontemporal_store(hv, reinterpret_cast<tg_u32x4*>(rowp));
                                __builtin_nontemporal_store(g.img_kind == 1 ? lv : hv, reinterpret_cast<tg_u32x4*>(rowp + N));
                                __builtin_nontemporal_store(g.img_kind == 1 ? hv : lv, reinterpret_cast<tg_u32x4*>(rowp + 2 * (int64_t)N));
                            } else {
                                *reinterpret_cast<tg_u32x4*>(rowp) = hv;
                                *reinterpret_cast<tg_u32x4*>(rowp + N) = g.img_kind == 1 ? lv : hv;
                                *reinterpret_cast<tg_u32x4*>(rowp + 2 * (int64_t)N) = g.img_kind == 1 ? hv : lv;
                            }
                        }
                    }
                }
                if (g.img_kind == 2) {
                    // column sums of the masked gradient over this row group's 32 RTW rows: the 8 row classes of a column group meet
                    // in LDS and are added in a fixed order
                    __syncthreads();
                    float* red = reinterpret_cast<float*>(smem);               // [MW][8][256]
#pragma unroll
                    for (int e = 0; e < 8; ++e) red[(rg * 8 + rbase) * 256 + c8 + e] = colacc[e];
                    __syncthreads();
                    float sres = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sres += red[(rg * 8 + j) * 256 + tl];
                    const int col = cb * NTB * 32 + tl;
                    if (col < N) g.img_colpart[(int64_t)(lid * MW + rg) * N + col] = sres;
                }
                of_amax_commit(g.img_amax, vmax);
                return;
            }
        }
        float ssq = 0.f;                                            // g.sumsq: this thread's share of the tile's sum of squares
#pragma unroll
        for (int m = 0; m < RTW; ++m) {
            if (m) __syncthreads();
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) es[mfma32_row(r, lane) * ESTR + (cw * NTW + n) * 32 + l31] = acc[m][n][r];
            __syncthreads();
            // MW == 2: tile index in the flat sequence -> (batch, row tile within the batch)
            const int ft = lid * RTW * MW + rg * RTW + m;
            const int tb = MW == 1 ? 0 : ft / g.a_tiles;
            const int row0 = MW == 1 ? rb * 64 + m * 32 : (ft - tb * g.a_tiles) * 32;
            float* obt = ob + (int64_t)tb * g.out_batch;
            for (int i = tl; i < 32 * C4; i += 256) {
                const int row = i / C4, c4 = (i % C4) * 4;
                const int grow = row0 + row, gcol = cb * NTB * 32 + c4;
                if (grow < g.rows_valid && gcol < N && g.out_bf16) {
                    const float4 v = *reinterpret_cast<const float4*>(es + row * ESTR + c4);
                    unsigned short* pb = reinterpret_cast<unsigned short*>(g.out) + batch * g.out_batch + (int64_t)tb * g.out_batch +
                                         (int64_t)grow * g.ldo + gcol;
                    *reinterpret_cast<uint2*>(pb) = make_uint2(tg_rne(v.x) | (tg_rne(v.y) << 16), tg_rne(v.z) | (tg_rne(v.w) << 16));
                } else if (MW == 1 && g.sumsq && grow < g.rows_valid && gcol < N) {
                    const float4 v = *reinterpret_cast<const float4*>(es + row * ESTR + c4);
                    ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq); ssq = fmaf(v.z, v.z, ssq); ssq = fmaf(v.w, v.w, ssq);
                } else if (grow < g.rows_valid && gcol < N) {
                    float4 v = *reinterpret_cast<const float4*>(es + row * ESTR + c4);
                    float4* p = reinterpret_cast<float4*>(obt + (int64_t)grow * g.ldo + gcol);
                    if (g.accumulate) {
                        const float4 o = *p;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    if (dbg & 2) { if (v.x == 123456.f) *p = v; }      // (measurement: no stores)
                    else if (g.nt_store)      // a result far larger than the caches, read much later
                        __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
                    else *p = v;
                }
            }
        }
        if (MW == 1 && g.sumsq) {              // fixed order: thread (loop order) -> wave butterfly -> waves in order
            ssq = wave_sum(ssq);
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);
            if (lane == 0) red[wave] = ssq;
            __syncthreads();
            if (tid == 0) g.sumsq[blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (red[0] + red[1]) + (red[2] + red[3]);
        }
    } else if (EPI == TG_EPI_ADAM) {
        // the tile is a gradient: through LDS into the row-contiguous mapping of the prefetched param / m / v pieces, clip factor,
        // TF-Adam (clip_adam.hip's arithmetic), non-temporal stores: every byte of the three arenas is touched once per step
        constexpr int ESTR = 128 + 4;
        float* es = reinterpret_cast<float*>(smem);
        const float fac = *g.adam_factor;
        const int gcol = cb * 128 + (tid & 31) * 4;
        __syncthreads();                       // nothing in flight (the last step waited for vmcnt(0)): the ring is free
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) es[mfma32_row(r, lane) * ESTR + cw * 32 + l31] = acc[m][0][r];
            __syncthreads();
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int j = m * 4 + j4;
                const int row = (tid >> 5) + 8 * j4, grow = rb * 64 + m * 32 + row;
                if (grow < g.rows_valid && gcol < N) {
                    const float4 gg = *reinterpret_cast<const float4*>(es + row * ESTR + (tid & 31) * 4);
                    const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
                    const int64_t o = (int64_t)grow * g.ldo + gcol;
                    f32x4 pn, mn, vn;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float pe = pa[j][q], me = ma[j][q], ve = va[j][q];
                        adam_element(gv[q] * fac, pe, me, ve, g.adam_lr_t, g.adam_b1, g.adam_b2, g.adam_eps);
                        pn[q] = pe; mn[q] = me; vn[q] = ve;
                    }
                    __builtin_nontemporal_store(pn, reinterpret_cast<f32x4*>(g.adam_p + o));
                    __builtin_nontemporal_store(mn, reinterpret_cast<f32x4*>(g.adam_m + o));
                    __builtin_nontemporal_store(vn, reinterpret_cast<f32x4*>(g.adam_v + o));
                    if (g.adam_p16) {          // the bf16 compute copy (round to nearest even: what the projection's passes read next step)
                        const of_f2 p01 = {pn[0], pn[1]}, p23 = {pn[2], pn[3]};
                        uint2 w;
                        w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(p01, of_b2));
                        w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(p23, of_b2));
                        *reinterpret_cast<uint2*>(g.adam_p16 + o) = w;
                    }
                }
            }
        }
    } else {
        // dA - ctil -> LDS [64][128 NTW + 1]; then the softmax backward row by row:
        //   dlogit~[t,k] = a[t,k] (g[t,k] - sum_j a[t,j] g[t,j]),  g = dA - ctil
        constexpr int ES = tg_epi_stride(NTW);
        constexpr int KPL = 2 * NTW;
        float* ds = reinterpret_cast<float*>(smem);
        __syncthreads();                       // nothing in flight (the last step waited for vmcnt(0)): the ring is free
        const float* ct = g.ctil + (int64_t)batch * N;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int col = (wave * NTW + n) * 32 + l31;
            const float c = (col < N) ? ct[col] : 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[(m * 32 + mfma32_row(r, lane)) * ES + col] = acc[m][n][r] - c;
        }
        __syncthreads();
        for (int rr = 0; rr < 16; ++rr) {
            const int row = wave * 16 + rr, t = rb * 64 + row;
            if (t >= g.rows_valid) continue;   // wave-uniform
            const int64_t grow = (int64_t)batch * g.rows_valid + t;
            float* out = g.out + grow * N;
            if (g.softmax) {
                const float* lr = g.logits + grow * N;
                const unsigned short* lrb = reinterpret_cast<const unsigned short*>(g.logits) + grow * N;
                float a[KPL], gg[KPL];
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int c = lane + 64 * j;
                    const float lv = (c < N) ? (g.logits_bf16 ? __uint_as_float((unsigned)lrb[c] << 16) : lr[c]) : 0.f;
                    a[j] = (c < N) ? fmaf(lv, g.scale ? g.scale[c] : 1.f, g.shift ? g.shift[c] : 0.f) : -INFINITY;
                    mx = fmaxf(mx, a[j]);
                }
                mx = wave_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    a[j] = __expf(a[j] - mx);
                    sum += a[j];
                }
                sum = wave_sum(sum);
                const float inv = 1.f / sum;
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int c = lane + 64 * j;
                    a[j] *= inv;
                    gg[j] = (c < N) ? ds[row * ES + c] : 0.f;
                    dot = fmaf(a[j], gg[j], dot);
                }
                dot = wave_sum(dot);
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int c = lane + 64 * j;
                    if (c < N) out[c] = a[j] * (gg[j] - dot);
                }
            } else {
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int c = lane + 64 * j;
                    if (c < N) out[c] = ds[row * ES + c];
                }
            }
        }
    }
}

// NTW for a problem with `cols` output columns: one column block when cols <= 512 (the row-wise epilogues need that),
// 256-column blocks beyond.
int tg_ntw(int cols) { const int nt = (cols + 31) / 32; return nt <= 4 ? 1 : (nt <= 8 ? 2 : (nt <= 16 ? 4 : 2)); }

static int tg_wide_enabled() {
    static const int on = [] {
        const char* e = getenv("LPM_TILE_GEMM_WIDE");      // 0: never use the 128-row form (A/B switch)
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on;
}

// The 128-row form applies when the row tiles of all batches form one flat sequence that divides into groups of four, the
// B operand is shared, there is one reduction segment and one 256-column block: K1's forward at K = 256.
static bool tg_wide_ok(const TileGemmArgs& g, int nbatch, int splits, int ntw, int planes, int tiles_per_wg = 4) {
    return tg_wide_enabled() && ntw == 2 && (g.cols_valid <= 256 || g.cols_valid % 256 == 0) && splits == 1 && g.steps2 == 0 &&
           g.a2 == nullptr && g.b_batch == 0 &&
           g.a_batch == (int64_t)g.a_tiles * g.a_tile && g.rb_per_batch * 2 == g.a_tiles && ((int64_t)nbatch * g.a_tiles) % tiles_per_wg == 0 &&
           g.steps_per_split >= (planes == 1 ? 32 : 16);     // (ring stages >= the ring depth)
}

// allow_wide: 0 = 64-row form, 1 = 128-row form (one workgroup per CU, software-pipelined, 4-stage ring) where the shape allows,
// 2 = 128-row form with a 3-stage ring and two workgroups per CU, 3 = 256-row form (four row tiles per wave; no statistics)
constexpr int TG_WIDE_NS_DEFAULT = 4;
constexpr int TG_AP_DEFAULT = 0;
// the fp16 forms (PL == 3: two terms, B hi-plane tiles; PL == 4: three terms): 256-row workgroups with the image epilogue only
static int tg_launch_f16_image(const TileGemmArgs& g, hipStream_t stream, const char* what, int planes) {
    if (!tg_wide_ok(g, 1, 1, 2, 3, 8)) {
        set_error("%s: the fp16 image form needs row tiles a multiple of 8 and >= 16 reduction steps", what);
        return LPM_ERR_UNSUPPORTED_SHAPE;
    }
    const int nt = (g.cols_valid + 31) / 32;
    dim3 grid((unsigned)(g.a_tiles / 8), (unsigned)((nt + 7) / 8), 1u);
    TileGemmArgs gl = g;
    gl.dbg = 0;
    gl.cols_inner = 0;
    // 4 stages x (16 row-tile pieces + 8 column tiles x (hi) or (hi, lo) planes)
    const size_t lds = (size_t)4 * (16 + (planes == 3 ? 4 : 8) * 2) * 1024;
    auto k3 = tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, 3, 4>;
    auto k4 = tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, 4, 4>;
    if (hipFuncSetAttribute(planes == 3 ? (const void*)k3 : (const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %zu bytes of LDS", what, lds);
        return LPM_ERR_LAUNCH;
    }
    if (planes == 3) hipLaunchKernelGGL(k3, grid, dim3(512), lds, stream, gl);
    else hipLaunchKernelGGL(k4, grid, dim3(512), lds, stream, gl);
    return check_launch(what);
}

template <int EPI, int PL>
static int tg_launch_pl(const TileGemmArgs& g, int nbatch, int splits, hipStream_t stream, const char* what, int ntw_override,
                        int timing_tag, int allow_wide) {
    if (EPI == TG_EPI_STORE && (g.cols_valid % 4 != 0 || g.ldo % 4 != 0 || g.out_batch % 4 != 0 || g.out_split % 4 != 0 ||
                                ((uintptr_t)g.out & 15) != 0)) {
        set_error("%s: the output needs 16-byte aligned rows (columns and leading dimension multiples of 4)", what);
        return LPM_ERR_BADARG;
    }
    if (PL == 1 && (g.a2 != nullptr || g.steps2 != 0)) {
        set_error("%s: the plain-bf16 tile form has no second operand pair", what);
        return LPM_ERR_UNSUPPORTED_SHAPE;
    }
    if (g.out_bf16 && (EPI != TG_EPI_STORE || g.accumulate || splits != 1 || g.ldo % 8 != 0 || g.out_batch % 8 != 0)) {
        set_error("%s: a bf16 output is written once (no accumulate, one split) with 16-byte aligned rows", what);
        return LPM_ERR_BADARG;
    }
    const int ntw = ntw_override ? ntw_override : tg_ntw(g.cols_valid);
    const int nt = (g.cols_valid + 31) / 32;
    if ((g.sumsq || EPI == TG_EPI_ADAM) && (allow_wide || splits != 1 || nbatch != 1 || g.out_bf16 || g.accumulate)) {
        set_error("%s: the gradient-consuming epilogues need the 64-row form, one batch, one split", what);
        return LPM_ERR_BADARG;
    }
    const bool wide4 = EPI == TG_EPI_STORE && allow_wide == 3 && !g.stats && !g.out_bf16 && tg_wide_ok(g, nbatch, splits, ntw, PL, 8);
    const bool wide = wide4 || (EPI == TG_EPI_STORE && allow_wide && allow_wide != 3 && tg_wide_ok(g, nbatch, splits, ntw, PL));
    dim3 grid((unsigned)(wide ? nbatch * g.a_tiles / (wide4 ? 8 : 4) : nbatch * g.rb_per_batch), (unsigned)((nt + 4 * ntw - 1) / (4 * ntw)),
              (unsigned)splits);
    TileGemmArgs gl = g;
    // LPM_TG_DBG (measurement, pipelined forms only): 1 no main loop, 2 no stores, 4 no DMA inside the loop, 8 no MFMAs
    static const int dbg_env = [] { const char* e = getenv("LPM_TG_DBG"); return e ? atoi(e) : 0; }();
    gl.dbg = dbg_env;
    if (!wide && g.cols_inner) {
        if (g.stats) { set_error("%s: cols_inner and the statistics epilogue index workgroups differently", what); return LPM_ERR_BADARG; }
        gl.cols_inner = (int)grid.y;
        grid.x *= grid.y;
        grid.y = 1;
    } else {
        gl.cols_inner = 0;
    }
    // ring depth of the pipelined forms (LPM_TG_WIDE_NS / LPM_TG_WIDE4_NS = 4, 5: A/B; NS - 2 steps of LDS-DMA stay in flight).
    // Round 4, one box: K1 47.6 / 48.8 / 49.8 us at 4 / 5 / 6 stages; the 256-row dense form 3-10 % slower at 5 -- the loop is not
    // waiting for data in flight.
    static const int wide_ns = [] { const char* e = getenv("LPM_TG_WIDE_NS"); const int v = e ? atoi(e) : 0; return (v >= 4 && v <= 5) ? v : TG_WIDE_NS_DEFAULT; }();
    static const int wide4_ns = [] { const char* e = getenv("LPM_TG_WIDE4_NS"); const int v = e ? atoi(e) : 0; return (v >= 4 && v <= 5) ? v : 4; }();
    // LPM_TG_AP (0 / 1): the anti-phase form of the 128- / 256-row workgroups (FORM 1) instead of the pipelined one.  Built and measured in
    // round 4, NOT faster (one box: K1 59.9 vs 51.2 us, qkv fwd 432 vs 406 us, ffn1 fwd 459 vs 448 us): a LOAD phase (3-4 LDS-DMA
    // issues + 8-12 fragment reads + two barrier episodes) is longer than a 12- / 24-MFMA COMPUTE phase, and a step costs twice the
    // longer of the two.  Kept for the A/B.
    static const int ap = [] { const char* e = getenv("LPM_TG_AP"); return e ? atoi(e) : TG_AP_DEFAULT; }();
    const bool wide2 = wide && allow_wide == 2;
    const size_t lds = wide4 ? (size_t)wide4_ns * (16 + 8 * ntw) * 1024
                             : (wide ? (size_t)(wide2 ? 3 : wide_ns) * (8 + 8 * ntw) * 1024 : tg_lds_bytes(ntw, EPI));
#define LPM_TG_LAUNCH_K(KERN, THREADS)                                                                                 \
    do {                                                                                                               \
        auto kern = KERN;                                                                                              \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            (void)hipGetLastError();                                                                                   \
            set_error("%s: cannot reserve %zu bytes of LDS", what, lds);                                               \
            return LPM_ERR_LAUNCH;                                                                                     \
        }                                                                                                              \
        hipEvent_t e0, e1;                                                                                             \
        if (timing_tag && timing_request(timing_tag, &e0, &e1))                                                        \
            hipExtLaunchKernelGGL(kern, grid, dim3(THREADS), lds, stream, e0, e1, 0, gl);                              \
        else                                                                                                           \
            hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, stream, gl);                                            \
    } while (0)
#define LPM_TG_LAUNCH(NTW) LPM_TG_LAUNCH_K((tile_gemm_kernel<NTW, EPI, 1, TG_NS, PL>), 256)
    if constexpr (EPI == TG_EPI_ADAM) {
        LPM_TG_LAUNCH(1);
    } else {
        if (wide4 && ap) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, PL, 4, 1>), 512);
        else if (wide && !wide2 && ap) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, PL, 2, 1>), 512);
        else if (wide4 && wide4_ns == 5) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 5, PL, 4>), 512);
        else if (wide4 && dbg_env) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, PL, 4, 0, 1>), 512);
        else if (wide4) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, PL, 4>), 512);
        else if (wide2) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 3, PL>), 512);
        else if (wide && wide_ns == 5) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 5, PL>), 512);
        else if (wide && dbg_env) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, PL, 2, 0, 1>), 512);
        else if (wide) LPM_TG_LAUNCH_K((tile_gemm_kernel<2, TG_EPI_STORE, 2, 4, PL>), 512);
        else if (ntw == 1) LPM_TG_LAUNCH(1);
        else if (ntw == 2) LPM_TG_LAUNCH(2);
        else LPM_TG_LAUNCH(4);
    }
#undef LPM_TG_LAUNCH
#undef LPM_TG_LAUNCH_K
    return check_launch(what);
}

template <int EPI>
static int tg_launch(const TileGemmArgs& g, int nbatch, int splits, hipStream_t stream, const char* what, int ntw_override = 0,
                     int timing_tag = 0, int allow_wide = 0, int planes = 2) {
    return planes == 1 ? tg_launch_pl<EPI, 1>(g, nbatch, splits, stream, what, ntw_override, timing_tag, allow_wide)
                       : tg_launch_pl<EPI, 2>(g, nbatch, splits, stream, what, ntw_override, timing_tag, allow_wide);
}


int tile_gemm_store(const TileGemmArgs& g, int nbatch, int splits, hipStream_t stream, const char* what, int ntw, int planes) {
    return tg_launch<TG_EPI_STORE>(g, nbatch, splits, stream, what, ntw, 0, 0, planes);
}
int tile_gemm_adam(const TileGemmArgs& g, hipStream_t stream, const char* what) {
    if (!g.adam_p || !g.adam_m || !g.adam_v || !g.adam_factor || g.cols_valid % 4 != 0 || g.ldo % 4 != 0 || g.stats || g.sumsq) {
        set_error("%s: the Adam epilogue needs param / m / v / factor and 16-byte aligned rows", what);
        return LPM_ERR_BADARG;
    }
    return tg_launch_pl<TG_EPI_ADAM, 2>(g, 1, 1, stream, what, 1, 0, 0);
}
int tile_gemm_softmax_bwd(const TileGemmArgs& g, int nbatch, hipStream_t stream, const char* what, int planes) {
    return tg_launch<TG_EPI_SOFTMAX_BWD>(g, nbatch, 1, stream, what, 0, 0, 0, planes);
}
int tile_gemm_ntw(int cols) { return tg_ntw(cols); }
int tile_gemm_image(const TileGemmArgs& g, hipStream_t stream, const char* what, int planes) {
    if (!g.img || (g.img_kind != 1 && g.img_kind != 2) || (g.img_kind == 1 && !g.img_bias) || (g.img_kind == 2 && (!g.img_mask || !g.img_colpart)) ||
        g.cols_valid % 256 != 0 || ((uintptr_t)g.img & 15) != 0 || g.stats || g.sumsq || g.accumulate) {
        set_error("%s: the image epilogue needs its operands, 16-byte aligned, and a multiple of 256 columns", what);
        return LPM_ERR_BADARG;
    }
    if (!tg_wide_ok(g, 1, 1, 2, 2, 8)) {
        set_error("%s: the image epilogue runs on the 128- / 256-row forms only (row tiles a multiple of 8, >= 16 reduction steps)", what);
        return LPM_ERR_UNSUPPORTED_SHAPE;
    }
    TileGemmArgs gl = g;
    if (!(gl.alpha > 0.f)) gl.alpha = 1.f;
    if (!(gl.img_scale > 0.f)) gl.img_scale = 1.f;
    gl.out = reinterpret_cast<float*>(g.img);      // (the fp32 output checks of the launcher: an aligned non-null pointer; never written)
    gl.ldo = 4; gl.out_batch = 0; gl.out_split = 0;
    if (gl.img_planes != 2) gl.img_planes = 3;
    if (gl.mask_planes != 2) gl.mask_planes = 3;
    if (planes >= 3) {
        static const int nt3 = [] { const char* e = getenv("LPM_DENSE_IMG_NT"); return (e && e[0] == '0') ? 0 : 1; }();
        gl.nt_store = nt3;
        return tg_launch_f16_image(gl, stream, what, planes);
    }
    // non-temporal stores: the 0.5 GB image is read back by the next GEMM long after it has left the caches (-6 us of 520 measured;
    // LPM_DENSE_IMG_NT=0: plain stores, A/B)
    static const int nt = [] { const char* e = getenv("LPM_DENSE_IMG_NT"); return (e && e[0] == '0') ? 0 : 1; }();
    gl.nt_store = nt;
    return tg_launch_pl<TG_EPI_STORE, 2>(gl, 1, 1, stream, what, 2, 0, tile_gemm_image_form() == 3 ? 2 : 3);
}
// 4 (default): 256-row workgroups, one per CU; 3: 128-row workgroups, two per CU (their epilogues overlap the neighbour's main loop)
int tile_gemm_image_form() {
    static const int form = [] { const char* e = getenv("LPM_DENSE_IMG_FORM"); return (e && e[0] == '3') ? 3 : 4; }();
    return form;
}
int tile_gemm_image_row_groups(int M) { return tile_gemm_image_form() == 3 ? M / 64 : M / 128; }

// [B*T, C] fp32 (row stride ldx) -> row tiles [b][mt][cs][plane][lane]; mt < 2*ceil(T/64), rows >= T are zero.
__global__ __launch_bounds__(256) void split_rows_tiles_kernel(const float* __restrict__ x, int64_t ldx, int B, int T, int C, int MT,
                                                               uint4* __restrict__ out, const OperandFmt fmt) {
    const int CS = C / 16;
    const int64_t total = (int64_t)B * MT * CS * 64;
    float vmax = 0.f;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int lane = (int)(w & 63);
        int64_t t = w >> 6;
        const int cs = (int)(t % CS);
        t /= CS;
        const int mt = (int)(t % MT), b = (int)(t / MT);
        const int row = mt * 32 + (lane & 31), c = cs * 16 + 8 * (lane >> 5);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < T) {
            const float* p = x + ((int64_t)b * T + row) * ldx + c;
            const float4 a = *reinterpret_cast<const float4*>(p);
            const float4 q = *reinterpret_cast<const float4*>(p + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
        }
        vmax = of_amax8(vmax, v);
        uint4 hi, lo;
        of_split8(v, fmt.f16, fmt.scale, hi, lo);
        const int64_t base = (w >> 6) * 128 + lane;
        out[base] = hi;
        out[base + 64] = lo;
    }
    of_amax_commit(fmt.amax, vmax);
}

// B operand of M[R, N] (reduction R, columns N): [rs][nt][plane][lane].  transposed: the source is stored [N, R].
// planes 2: split-bf16 (hi, lo); 1: plain bf16; 3: fp16, the hi plane only (the one-plane operand of a two-term product, rounded once);
// 4: fp16 (hi, lo)
__global__ __launch_bounds__(256) void split_weight_tiles_kernel(const float* __restrict__ W, int R, int N, int transposed,
                                                                 uint4* __restrict__ wt, int planes) {
    const int RS = R / 16, NT = (N + 31) / 32;
    const int64_t total = (int64_t)RS * NT * 64;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int lane = (int)(w & 63);
        const int64_t t = w >> 6;
        const int nt = (int)(t % NT), rs = (int)(t / NT);
        const int col = nt * 32 + (lane & 31), r = rs * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (col < N) ? (transposed ? W[(int64_t)col * R + r + e] : W[(int64_t)(r + e) * N + col]) : 0.f;
        if (planes == 3) {
            wt[t * 64 + lane] = make_uint4(of_round2_f16(v[0], v[1]), of_round2_f16(v[2], v[3]), of_round2_f16(v[4], v[5]), of_round2_f16(v[6], v[7]));
            continue;
        }
        uint4 hi, lo;
        if (planes == 4) of_split8(v, 1, 1.f, hi, lo);
        else tg_split8(v, hi, lo);
        if (planes == 1) {
            wt[t * 64 + lane] = hi;
            continue;
        }
        const int64_t base = t * 128 + lane;
        wt[base] = hi;
        wt[base + 64] = lo;
    }
}

// ... of the matrix [x1 * scale | x2] (lpm_proj_fwd_parts' operand): column c < n1a is x1[r][c] * scale[r][c % ks], the rest x2[r][c - n1a]
__global__ __launch_bounds__(256) void split_weight_tiles_parts_kernel(const float* __restrict__ x1, int64_t ld1, int64_t n1a, int x1_bf16,
                                                                       const float* __restrict__ scale, int ks, const float* __restrict__ x2,
                                                                       int64_t ld2, int R, int64_t N, uint4* __restrict__ wt) {
    const int RS = R / 16;
    const int64_t NT = (N + 31) / 32;
    const int64_t total = (int64_t)RS * NT * 64;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int lane = (int)(w & 63);
        const int64_t t = w >> 6;
        const int64_t nt = t % NT;
        const int rs = (int)(t / NT);
        const int64_t col = nt * 32 + (lane & 31);
        const int r = rs * 16 + 8 * (lane >> 5);
        float v[8];
        if (col < n1a) {
            const int kc = (int)(col % ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xv = x1_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(x1)[(int64_t)(r + e) * ld1 + col] << 16)
                                         : x1[(int64_t)(r + e) * ld1 + col];
                v[e] = xv * scale[(int64_t)(r + e) * ks + kc];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (col < N) ? x2[(int64_t)(r + e) * ld2 + (col - n1a)] : 0.f;
        }
        uint4 hi, lo;
        tg_split8(v, hi, lo);
        const int64_t base = t * 128 + lane;
        wt[base] = hi;
        wt[base + 64] = lo;
    }
}

// ... four adjacent columns per thread (n1a, N multiples of 4; 16-byte aligned rows): a row of the block is one 8- / 16-byte load per thread
// instead of four 2- / 4-byte ones, the scales one float4; the thread writes its four lanes' planes as 64 contiguous bytes each
// (cfg-5, bf16 sums: 138 us for 134 MB in + 277 MB out with the one-column form)
__global__ __launch_bounds__(256) void split_weight_tiles_parts4_kernel(const float* __restrict__ x1, int64_t ld1, int64_t n1a, int x1_bf16,
                                                                        const float* __restrict__ scale, int ks, const float* __restrict__ x2,
                                                                        int64_t ld2, int R, int64_t N, uint4* __restrict__ wt) {
    const int RS = R / 16;
    const int64_t NT = (N + 31) / 32;
    const int64_t total = (int64_t)RS * NT * 16;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int q = (int)(w & 15);
        const int64_t t = w >> 4;
        const int64_t nt = t % NT;
        const int rs = (int)(t / NT);
        const int cg = q & 7, half = q >> 3;
        const int64_t col = nt * 32 + cg * 4;
        const int r = rs * 16 + 8 * half;
        float v[4][8];
        if (col < n1a) {
            const int kc = (int)(col % ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xv[4];
                if (x1_bf16) {
                    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x1) + (int64_t)(r + e) * ld1 + col);
                    xv[0] = __uint_as_float(u.x << 16); xv[1] = __uint_as_float(u.x & 0xffff0000u);
                    xv[2] = __uint_as_float(u.y << 16); xv[3] = __uint_as_float(u.y & 0xffff0000u);
                } else {
                    const float4 f = *reinterpret_cast<const float4*>(x1 + (int64_t)(r + e) * ld1 + col);
                    xv[0] = f.x; xv[1] = f.y; xv[2] = f.z; xv[3] = f.w;
                }
                const float4 sc = *reinterpret_cast<const float4*>(scale + (int64_t)(r + e) * ks + kc);
                v[0][e] = xv[0] * sc.x; v[1][e] = xv[1] * sc.y; v[2][e] = xv[2] * sc.z; v[3][e] = xv[3] * sc.w;
            }
        } else if (col < N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 f = *reinterpret_cast<const float4*>(x2 + (int64_t)(r + e) * ld2 + (col - n1a));
                v[0][e] = f.x; v[1][e] = f.y; v[2][e] = f.z; v[3][e] = f.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
        }
        const int64_t base = t * 128 + half * 32 + cg * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 hi, lo;
            tg_split8(v[j], hi, lo);
            wt[base + j] = hi;
            wt[base + 64 + j] = lo;
        }
    }
}

// out[i] = sum_z part[z][i]   (float4 granularity)
__global__ __launch_bounds__(256) void tg_reduce_splits_kernel(const float4* __restrict__ part, int Z, int64_t n4,
                                                               float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = part[i];
    for (int z = 1; z < Z; ++z) {
        const float4 p = part[(int64_t)z * n4 + i];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    out[i] = s;
}

// outs[i][k][c] = sum_z part[z][k][i * (N / nouts) + c]: the split-K partial sums of a weight-gradient GEMM [Z, K, N] added in slice order
// straight into up to three destination matrices [K, N / nouts] (the q | k | v kernels' gradient slots: one launch instead of three
// strided reductions).  float4 granularity; thread = one float4 of one destination row.
// halves == 2 (the fp16 two-product form, dW = xh^T [dyh | dyl]): part is [Z, K, 2 N], the lo half's products N columns to the right;
// per slice hi + lo, slices in order, then * alpha (1 / the operands' scales, a power of two: exact).
struct SumSplitsArgs { const float* part; float* out[3]; int Z, K, N, nouts, halves; float alpha; };
__global__ __launch_bounds__(256) void sum_splits_kernel(const SumSplitsArgs a) {
    const int Nb = a.N / a.nouts, Nb4 = Nb / 4;
    const int64_t per = (int64_t)a.K * Nb4, total = per * a.nouts;
    const int64_t ldp = (int64_t)a.N * a.halves;
    const int64_t zstride = (int64_t)a.K * ldp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o = (int)(i / per);
        const int64_t r = i - (int64_t)o * per;
        const int64_t k = r / Nb4;
        const int c = (int)(r - k * Nb4) * 4;
        const float* p = a.part + k * ldp + (int64_t)o * Nb + c;
        float4 s = *reinterpret_cast<const float4*>(p);
        if (a.halves == 2) {
            const float4 q = *reinterpret_cast<const float4*>(p + a.N);
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        for (int z = 1; z < a.Z; ++z) {
            float4 q = *reinterpret_cast<const float4*>(p + z * zstride);
            if (a.halves == 2) {
                const float4 q2 = *reinterpret_cast<const float4*>(p + z * zstride + a.N);
                q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w;
            }
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        if (a.alpha != 1.f) { s.x *= a.alpha; s.y *= a.alpha; s.z *= a.alpha; s.w *= a.alpha; }
        *reinterpret_cast<float4*>(a.out[o] + k * Nb + c) = s;
    }
}

// colpart [nblk][N] -> out [N]  (fp64 accumulation; 1024 threads per 16 columns, partial_colsums16)
__global__ __launch_bounds__(1024) void tg_colsum_reduce_kernel(const float* __restrict__ colpart, int nblk, int N, float* __restrict__ out) {
    double s, q;
    int c;
    partial_colsums16(colpart, nblk, (int64_t)N, 0, N, s, q, c);
    if (threadIdx.x < 16 && c < N) out[c] = (float)s;
}

// split-bf16 operand image [M][3K] (split_gemm.hip; lo plane lo_off elements into the row) -> row tiles [mt][cs][plane][lane]
__global__ __launch_bounds__(256) void image_row_tiles_kernel(const unsigned short* __restrict__ x3, int64_t M, int K, int lo_off, int MT,
                                                              uint4* __restrict__ out, int planes) {
    const int CS = K / 16;
    const int64_t total = (int64_t)MT * CS * 64;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int lane = (int)(w & 63);
        const int64_t t = w >> 6;
        const int cs = (int)(t % CS);
        const int64_t row = (t / CS) * 32 + (lane & 31);
        const int c = cs * 16 + 8 * (lane >> 5);
        uint4 hi = make_uint4(0u, 0u, 0u, 0u), lo = hi;
        if (row < M) {
            const unsigned short* p = x3 + row * planes * K + c;
            hi = *reinterpret_cast<const uint4*>(p);
            lo = *reinterpret_cast<const uint4*>(p + lo_off);
        }
        const int64_t base = t * 128 + lane;
        out[base] = hi;
        out[base + 64] = lo;
    }
}

static inline int row_tiles_per_clip(int T) { return 2 * ((T + 63) / 64); }
// frame steps per clip: ceil(T / 16) in the split form; the plain-bf16 form pads every clip to whole 64-frame blocks (an even
// number of steps -- a ring stage carries two -- and the same frames as the row tiles cover)
static inline int frame_steps(int T, int planes) { return planes == 1 ? 4 * ((T + 63) / 64) : (T + 15) / 16; }
static inline int dw_splits(int B, int T, int D, int K, int planes = 2) {
    const int blocks = ((D + 63) / 64) * (((K + 31) / 32 + 4 * tg_ntw(K) - 1) / (4 * tg_ntw(K)));
    const int steps = B * frame_steps(T, planes);
    int z = (512 + blocks - 1) / blocks;
    if (z > steps / 8) z = steps / 8;
    if (z > 32) z = 32;                       // the reduce pass reads z partial copies of dW
    return z < 1 ? 1 : z;
}

}  // namespace lpm

// Split-K partial sums of a weight-gradient product [Z, K, N] (what a batched library GEMM over Z slices of a long reduction leaves) ->
// the gradient, added in slice order, written straight into up to three [K, N / nouts] destinations (the column blocks of a concatenated
// weight: q | k | v): TF autodiff of tf.layers.dense at transformer_utils.py:559-561,583,701-711.  N / nouts a multiple of 4.
extern "C" int lpm_sum_splits_scaled(const float* part, int Z, int K, int N, int halves, float alpha, float* out0, float* out1, float* out2,
                                     int nouts, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(part && out0 && Z >= 1 && K > 0 && N > 0 && nouts >= 1 && nouts <= 3 && (halves == 1 || halves == 2) && alpha > 0.f, LPM_ERR_BADARG,
                "lpm_sum_splits: bad arguments");
    LPM_REQUIRE(N % nouts == 0 && (N / nouts) % 4 == 0 && (nouts < 2 || out1) && (nouts < 3 || out2), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_sum_splits: N / nouts must be a multiple of 4 and every destination given (N=%d nouts=%d)", N, nouts);
    LPM_REQUIRE((((uintptr_t)part | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)out2) & 15) == 0, LPM_ERR_BADARG, "lpm_sum_splits: 16-byte aligned pointers");
    SumSplitsArgs a{part, {out0, out1, out2}, Z, K, N, nouts, halves, alpha};
    const int64_t total = (int64_t)K * (N / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("lpm_sum_splits");
}
extern "C" int lpm_sum_splits(const float* part, int Z, int K, int N, float* out0, float* out1, float* out2, int nouts, lpm_stream_t stream) {
    return lpm_sum_splits_scaled(part, Z, K, N, 1, 1.f, out0, out1, out2, nouts, stream);
}

extern "C" size_t lpm_row_tiles_bytes(int B, int T, int C) {
    return (size_t)B * lpm::row_tiles_per_clip(T) * (C / 16) * 2048;
}
extern "C" size_t lpm_weight_tiles_bytes(int R, int N) { return (size_t)(R / 16) * ((N + 31) / 32) * 2048; }
extern "C" int lpm_assign_gemm_tiles_nblk(int B, int T) { return B * ((T + 63) / 64); }
extern "C" int lpm_assign_gemm_tiles_supported(int T, int D, int K) {
    return (D % 32 == 0 && K % 32 == 0 && K <= 512 && D > 0 && K > 0 && T > 0) ? 1 : 0;
}

extern "C" int lpm_split_rows_tiles(const float* x, int64_t ldx, int B, int T, int C, void* out, lpm_stream_t stream) {
    return lpm_split_rows_tiles_fmt(x, ldx, B, T, C, out, nullptr, stream);
}
extern "C" int lpm_split_rows_tiles_fmt(const float* x, int64_t ldx, int B, int T, int C, void* out, const LpmOperandFormat* fmt,
                                        lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && out, LPM_ERR_BADARG, "lpm_split_rows_tiles: null pointer");
    if (const int rc = operand_fmt_check(fmt, "lpm_split_rows_tiles")) return rc;
    LPM_REQUIRE(B > 0 && T > 0 && C > 0 && C % 16 == 0 && ldx >= C && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0,
                LPM_ERR_UNSUPPORTED_SHAPE, "lpm_split_rows_tiles: need C %% 16 == 0, ldx %% 4 == 0, aligned pointers (C=%d)", C);
    const int MT = row_tiles_per_clip(T);
    const int64_t total = (int64_t)B * MT * (C / 16) * 64;
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(split_rows_tiles_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       B, T, C, MT, (uint4*)out, operand_fmt(fmt));
    return check_launch("lpm_split_rows_tiles");
}

static int split_weight_tiles_impl(const float* w, int R, int N, int transposed, void* wt, int planes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(w && wt, LPM_ERR_BADARG, "lpm_split_weight_tiles: null pointer");
    LPM_REQUIRE(R > 0 && N > 0 && R % 16 == 0, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_split_weight_tiles: need R %% 16 == 0 (R=%d N=%d)", R, N);
    const int64_t total = (int64_t)(R / 16) * ((N + 31) / 32) * 64;
    hipLaunchKernelGGL(split_weight_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, R, N,
                       transposed, (uint4*)wt, planes);
    return check_launch("lpm_split_weight_tiles");
}
// the weight tiles (lpm_weight_tiles_bytes(R, N)) of [x1 * scale | x2] [R, N]: see lpm_proj_fwd_parts for the operand's description
extern "C" int lpm_split_weight_tiles_parts(const void* x1, int64_t ld1, int64_t n1a, int x1_bf16, const float* scale, int ks, const float* x2,
                                            int64_t ld2, int R, int64_t N, void* wt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x1 && scale && wt && (x2 || n1a == N), LPM_ERR_BADARG, "lpm_split_weight_tiles_parts: null pointer");
    LPM_REQUIRE(R > 0 && R % 16 == 0 && N > 0 && n1a > 0 && n1a <= N && n1a % 32 == 0 && ks > 0 && n1a % ks == 0 && ld1 >= n1a &&
                (n1a == N || ld2 >= N - n1a), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_split_weight_tiles_parts: need R %% 16 == 0, the scaled block a multiple of 32 columns and of ks (R=%d n1a=%lld ks=%d)", R,
                (long long)n1a, ks);
    const int64_t total = (int64_t)(R / 16) * ((N + 31) / 32) * 64;
    // four columns per thread where every row piece is an aligned vector (LPM_SWT_PARTS4=0: the one-column form, A/B)
    static const int four = [] { const char* e = getenv("LPM_SWT_PARTS4"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool al1 = ld1 % 4 == 0 && ((uintptr_t)x1 & (x1_bf16 ? 7 : 15)) == 0 && ks % 4 == 0 && ((uintptr_t)scale & 15) == 0;
    const bool al2 = n1a == N || (ld2 % 4 == 0 && ((uintptr_t)x2 & 15) == 0);
    if (four && n1a % 4 == 0 && N % 4 == 0 && al1 && al2) {
        const int64_t want4 = (total / 4 + 255) / 256;
        hipLaunchKernelGGL(split_weight_tiles_parts4_kernel, dim3((unsigned)(want4 < 65536 ? want4 : 65536)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x1, ld1, n1a, x1_bf16 ? 1 : 0, scale, ks, x2, ld2, R, N, (uint4*)wt);
        return check_launch("lpm_split_weight_tiles_parts");
    }
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(split_weight_tiles_parts_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x1, ld1, n1a, x1_bf16 ? 1 : 0, scale, ks, x2, ld2, R, N, (uint4*)wt);
    return check_launch("lpm_split_weight_tiles_parts");
}
extern "C" int lpm_split_weight_tiles(const float* w, int R, int N, int transposed, void* wt, lpm_stream_t stream) {
    return split_weight_tiles_impl(w, R, N, transposed, wt, 2, stream);
}
extern "C" int lpm_split_weight_tiles_bf16(const float* w, int R, int N, int transposed, void* wt, lpm_stream_t stream) {
    return split_weight_tiles_impl(w, R, N, transposed, wt, 1, stream);
}
extern "C" int lpm_split_weight_tiles_fmt(const float* w, int R, int N, int transposed, void* wt, int kind, lpm_stream_t stream) {
    LPM_REQUIRE(lpm::operand_kind_ok(kind), LPM_ERR_BADARG, "lpm_split_weight_tiles_fmt: unknown operand format %d", kind);
    return split_weight_tiles_impl(w, R, N, transposed, wt, kind == LPM_OPERAND_FP16X2 ? 3 : (kind == LPM_OPERAND_FP16X3 ? 4 : 2), stream);
}

static int assign_gemm_tiles_fwd_impl(const void* xr, const void* wt, int B, int T, int D, int K, void* logits, float* partial,
                                      int planes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(xr && wt && logits && partial, LPM_ERR_BADARG, "lpm_assign_gemm_tiles_fwd: null pointer");
    LPM_REQUIRE(B > 0 && lpm_assign_gemm_tiles_supported(T, D, K), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_assign_gemm_tiles_fwd: need D %% 32 == 0, K %% 32 == 0, K <= 512 (D=%d K=%d)", D, K);
    const int MT = row_tiles_per_clip(T), DS = D / 16, NT = K / 32;
    // fp32 storage, K a multiple of 256, D a multiple of 64: flat 96-row workgroups (assign_flat.hip; LPM_K1_FLAT=0: the forms below)
    if (planes == 2 && assign_flat_ok(B, T, D, K))
        return assign_flat_launch(xr, wt, B, T, MT, D, K, (float*)logits, partial, lpm_assign_gemm_tiles_nblk(B, T),
                                  D >= 1024 ? LPM_TIMING_K1 : 0, (hipStream_t)stream, "lpm_assign_gemm_tiles_fwd");
    // bf16 storage, K a multiple of 512: 160-row x 512-column workgroups (LPM_K1_WIDE=0: the flat form)
    if (planes == 1 && assign_wide_ok(B, T, MT, D, K, lpm_assign_gemm_tiles_nblk(B, T)))
        return assign_wide_launch(xr, wt, B, T, MT, D, K, logits, partial, lpm_assign_gemm_tiles_nblk(B, T),
                                  D >= 1024 ? LPM_TIMING_K1 : 0, (hipStream_t)stream, "lpm_assign_gemm_tiles_fwd_bf16");
    // bf16 storage: the same flat workgroups on plain tiles (double steps), bf16 logits
    if (planes == 1 && assign_flat_plain_ok(B, T, D, K))
        return assign_flat_launch(xr, wt, B, T, MT, D, K, (float*)logits, partial, lpm_assign_gemm_tiles_nblk(B, T),
                                  D >= 1024 ? LPM_TIMING_K1 : 0, (hipStream_t)stream, "lpm_assign_gemm_tiles_fwd_bf16", 1);
    const int64_t U = 64 * planes;             // 16-byte units per (tile, step)
    TileGemmArgs g{};
    g.a = (const uint4*)xr; g.a_tile = (int64_t)DS * U; g.a_step = U; g.a_batch = (int64_t)MT * DS * U; g.a_tiles = MT;
    g.b = (const uint4*)wt; g.b_tile = U; g.b_step = (int64_t)NT * U; g.b_batch = 0; g.b_tiles = NT;
    g.rb_per_batch = MT / 2; g.steps_per_split = DS; g.total_steps = DS;
    g.out = (float*)logits; g.out_bf16 = planes == 1 ? 1 : 0; g.ldo = K; g.out_batch = (int64_t)T * K; g.out_split = 0;
    g.rows_valid = T; g.cols_valid = K; g.accumulate = 0; g.stats = partial;
    // K = 512 (cfg-5's video stream): the 128-row form on two 256-column blocks instead of the 64-row x 512-column form (93.8 us there)
    const int ntw = (K == 512 && T >= 32) ? 2 : 0;
    return tg_launch<TG_EPI_STORE>(g, B, 1, (hipStream_t)stream, "lpm_assign_gemm_tiles_fwd", ntw, D >= 1024 ? LPM_TIMING_K1 : 0, 1, planes);
}
extern "C" int lpm_assign_gemm_tiles_fwd(const void* xr, const void* wt, int B, int T, int D, int K, float* logits, float* partial,
                                         lpm_stream_t stream) {
    return assign_gemm_tiles_fwd_impl(xr, wt, B, T, D, K, logits, partial, 2, stream);
}
// bf16 storage (BASELINE cfg-5): xr, wt plain bf16 tiles (lpm_frame_apply_tiles_bf16 / lpm_split_weight_tiles_bf16), one MFMA
// per product, fp32 accumulation; the logits are STORED as bf16 [B*T, K] (K %% 8 == 0), the batch statistics come from the fp32
// accumulators.
extern "C" int lpm_assign_gemm_tiles_fwd_bf16(const void* xr, const void* wt, int B, int T, int D, int K, void* logits_bf16,
                                              float* partial, lpm_stream_t stream) {
    return assign_gemm_tiles_fwd_impl(xr, wt, B, T, D, K, logits_bf16, partial, 1, stream);
}

// y[M, N] (row stride ldo) = x . w for a dense layer (transformer_utils.py:559-561,583,701-711): xr = row tiles of x [M, Kd]
// (lpm_split_rows_tiles with B = 1, T = M), wt = weight tiles of w [Kd, N].  form 0 / 2: 128-row workgroups (one per CU, pipelined);
// 1: 64-row workgroups; 3: 128-row workgroups, two per CU; 4: 256-row workgroups (four row tiles per wave).
extern "C" int lpm_dense_tiles_fwd(const void* xr, const void* wt, int M, int Kd, int N, float* y, int64_t ldo, int form,
                                   lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(xr && wt && y, LPM_ERR_BADARG, "lpm_dense_tiles_fwd: null pointer");
    LPM_REQUIRE(M > 0 && Kd > 0 && N > 0 && Kd % 16 == 0 && N % 32 == 0 && ldo >= N, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_dense_tiles_fwd: need Kd %% 16 == 0, N %% 32 == 0 (Kd=%d N=%d)", Kd, N);
    const int MT = row_tiles_per_clip(M), DS = Kd / 16, NT = N / 32;
    TileGemmArgs g{};
    g.a = (const uint4*)xr; g.a_tile = (int64_t)DS * 128; g.a_step = 128; g.a_batch = (int64_t)MT * DS * 128; g.a_tiles = MT;
    g.b = (const uint4*)wt; g.b_tile = 128; g.b_step = (int64_t)NT * 128; g.b_batch = 0; g.b_tiles = NT;
    g.rb_per_batch = MT / 2; g.steps_per_split = DS; g.total_steps = DS;
    g.out = y; g.ldo = ldo; g.out_batch = (int64_t)M * ldo; g.out_split = 0;
    g.rows_valid = M; g.cols_valid = N;
    return tg_launch<TG_EPI_STORE>(g, 1, 1, (hipStream_t)stream, "lpm_dense_tiles_fwd", 2, 0, form == 1 ? 0 : (form == 3 ? 2 : (form == 4 ? 3 : 1)), 2);
}

// ---- FeedForwardNetwork's first dense layer and its backward on the 256-row tile GEMM with operand-image epilogues -------------------------
// (transformer_utils.py:701-711: relu(y W1 + b1) feeds the second dense layer; TF autodiff for the backward).  The [M, 4F] tensor in
// the middle exists only as the split-bf16 image the library GEMMs downstream read: no fp32 round trip, no separate split pass.
extern "C" int lpm_dense_tiles_supported(int M, int Kd, int N) {
    return (M > 0 && M % 256 == 0 && Kd >= 256 && Kd % 16 == 0 && N > 0 && N % 256 == 0) ? 1 : 0;
}
// a_kind < 0: `ar` holds row tiles; else an operand image of that kind [M][planes Kd] read in place (order: the three-plane image's plane
// order -- 0 activation [hi | lo | hi], 1 gradient [hi | hi | lo])
static void dense_tiles_args(lpm::TileGemmArgs& g, const void* ar, const void* bt, int M, int Kd, int N, int a_kind = -1, int a_order = 0) {
    const int MT = lpm::row_tiles_per_clip(M), DS = Kd / 16, NT = N / 32;
    g.a = (const uint4*)ar; g.a_tile = (int64_t)DS * 128; g.a_step = 128; g.a_batch = (int64_t)MT * DS * 128; g.a_tiles = MT;
    if (a_kind >= 0) {
        const int planes = lpm::operand_kind_planes(a_kind);
        g.a_img_row = planes * Kd / 8;
        g.a_img_lo = ((planes == 3 && a_order) ? 2 * Kd : Kd) / 8;
        g.a_tile = (int64_t)32 * g.a_img_row; g.a_step = 2; g.a_batch = (int64_t)MT * g.a_tile;
    }
    g.b = (const uint4*)bt; g.b_tile = 128; g.b_step = (int64_t)NT * 128; g.b_batch = 0; g.b_tiles = NT;
    g.rb_per_batch = MT / 2; g.steps_per_split = DS; g.total_steps = DS;
    g.rows_valid = M; g.cols_valid = N;
}
// out3 [M, 3N] bf16 = the activation image [hi | lo | hi] of relu(x . w + bias); xr: row tiles of x [M, Kd], wt: weight tiles of w [Kd, N]
extern "C" int lpm_dense_tiles_act_image_fwd(const void* xr, const void* wt, const float* bias, int M, int Kd, int N, void* out3,
                                             lpm_stream_t stream) {
    return lpm_dense_tiles_act_image_fwd_fmt(xr, -1, wt, bias, M, Kd, N, 1.f, out3, nullptr, stream);
}
// one_plane_b: the weight operand as fp16 hi-plane tiles of 64 units (the two-term product of the backward)
static void dense_tiles_fmt(lpm::TileGemmArgs& g, const LpmOperandFormat* fmt, float in_inv_scale, int N, bool one_plane_b) {
    const lpm::OperandFmt f = lpm::operand_fmt(fmt);
    g.img_f16 = f.f16; g.img_planes = f.planes; g.img_scale = f.scale; g.img_amax = f.amax; g.alpha = in_inv_scale;
    if (f.f16 && one_plane_b) {
        const int NT = N / 32;
        g.b_tile = 64; g.b_step = (int64_t)NT * 64;
    }
}
extern "C" int lpm_dense_tiles_act_image_fwd_fmt(const void* xr, int x_image_kind, const void* wt, const float* bias, int M, int Kd, int N,
                                                 float in_inv_scale, void* out3, const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(xr && wt && bias && out3, LPM_ERR_BADARG, "lpm_dense_tiles_act_image_fwd: null pointer");
    LPM_REQUIRE(x_image_kind < 0 || (operand_kind_ok(x_image_kind) && Kd % 8 == 0 && ((uintptr_t)xr & 15) == 0), LPM_ERR_BADARG,
                "lpm_dense_tiles_act_image_fwd: bad operand image (kind %d)", x_image_kind);
    if (const int rc = operand_fmt_check(fmt, "lpm_dense_tiles_act_image_fwd")) return rc;
    LPM_REQUIRE(in_inv_scale > 0.f, LPM_ERR_BADARG, "lpm_dense_tiles_act_image_fwd: in_inv_scale must be positive");
    LPM_REQUIRE(lpm_dense_tiles_supported(M, Kd, N), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_dense_tiles_act_image_fwd: need M %% 256 == 0, Kd %% 16 == 0, Kd >= 256, N %% 256 == 0 (M=%d Kd=%d N=%d)", M, Kd, N);
    LPM_REQUIRE(((uintptr_t)bias & 15) == 0, LPM_ERR_BADARG, "lpm_dense_tiles_act_image_fwd: bias must be 16-byte aligned");
    TileGemmArgs g{};
    dense_tiles_args(g, xr, wt, M, Kd, N, x_image_kind, 0);
    g.img = (unsigned short*)out3; g.img_kind = 1; g.img_bias = bias;
    dense_tiles_fmt(g, fmt, in_inv_scale, N, false);      // forward: three terms, the weight tiles carry (hi, lo) like the row tiles
    return tile_gemm_image(g, (hipStream_t)stream, "lpm_dense_tiles_act_image_fwd", g.img_f16 ? 4 : 2);
}
extern "C" size_t lpm_dense_tiles_relu_bwd_workspace_bytes(int M, int N) { return (size_t)((M + 63) / 64) * N * sizeof(float); }
// g = (dy . w^T) masked by [act > 0]: out3 [M, 3N] bf16 = its gradient image [hi | hi | lo], dbias [N] = its column sums.
// dyr: row tiles of dy [M, Kd]; wtt: weight tiles of w^T [Kd, N] (lpm_split_weight_tiles of w [N, Kd] with transposed = 1);
// act3 [M, 3N]: the forward's activation image (lpm_dense_tiles_act_image_fwd's output).
extern "C" int lpm_dense_tiles_relu_bwd_image(const void* dyr, const void* wtt, const void* act3, int M, int Kd, int N, void* out3,
                                              float* dbias, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    return lpm_dense_tiles_relu_bwd_image_fmt(dyr, -1, wtt, act3, LPM_OPERAND_BF16X3, M, Kd, N, 1.f, out3, dbias, workspace, workspace_bytes, nullptr, stream);
}
extern "C" int lpm_dense_tiles_relu_bwd_image_fmt(const void* dyr, int dy_image_kind, const void* wtt, const void* act3, int act_kind, int M, int Kd,
                                                  int N, float in_inv_scale, void* out3, float* dbias, void* workspace, size_t workspace_bytes,
                                                  const LpmOperandFormat* fmt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dyr && wtt && act3 && out3 && dbias && workspace, LPM_ERR_BADARG, "lpm_dense_tiles_relu_bwd_image: null pointer");
    LPM_REQUIRE(dy_image_kind < 0 || (operand_kind_ok(dy_image_kind) && Kd % 8 == 0 && ((uintptr_t)dyr & 15) == 0), LPM_ERR_BADARG,
                "lpm_dense_tiles_relu_bwd_image: bad operand image (kind %d)", dy_image_kind);
    if (const int rc = operand_fmt_check(fmt, "lpm_dense_tiles_relu_bwd_image")) return rc;
    LPM_REQUIRE(in_inv_scale > 0.f && operand_kind_ok(act_kind), LPM_ERR_BADARG, "lpm_dense_tiles_relu_bwd_image: bad in_inv_scale / act_kind");
    LPM_REQUIRE(lpm_dense_tiles_supported(M, Kd, N), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_dense_tiles_relu_bwd_image: need M %% 256 == 0, Kd %% 16 == 0, Kd >= 256, N %% 256 == 0 (M=%d Kd=%d N=%d)", M, Kd, N);
    LPM_REQUIRE(workspace_bytes >= lpm_dense_tiles_relu_bwd_workspace_bytes(M, N), LPM_ERR_WORKSPACE,
                "lpm_dense_tiles_relu_bwd_image: workspace too small");
    LPM_REQUIRE(((uintptr_t)act3 & 15) == 0, LPM_ERR_BADARG, "lpm_dense_tiles_relu_bwd_image: act3 must be 16-byte aligned");
    TileGemmArgs g{};
    dense_tiles_args(g, dyr, wtt, M, Kd, N, dy_image_kind, 1);
    g.img = (unsigned short*)out3; g.img_kind = 2; g.img_mask = (const unsigned short*)act3; g.img_colpart = (float*)workspace;
    dense_tiles_fmt(g, fmt, in_inv_scale, N, true);       // backward: two terms against hi-plane weight tiles
    g.mask_planes = operand_kind_planes(act_kind);
    const int rc = tile_gemm_image(g, (hipStream_t)stream, "lpm_dense_tiles_relu_bwd_image", g.img_f16 ? 3 : 2);
    if (rc != LPM_OK) return rc;
    hipLaunchKernelGGL(tg_colsum_reduce_kernel, dim3((unsigned)((N + 15) / 16)), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace,
                       tile_gemm_image_row_groups(M), N, dbias);
    return check_launch("lpm_dense_tiles_relu_bwd_image");
}
// x3 [M, 3K] split-bf16 operand image (order 0: activation planes [hi | lo | hi], 1: gradient planes [hi | hi | lo]) -> row tiles of the
// matrix (lpm_row_tiles_bytes(1, M, K)): the tile GEMM's A operand from a tensor that exists only as its image
extern "C" int lpm_image_row_tiles(const void* x3, int M, int K, int order, void* out, lpm_stream_t stream) {
    return lpm_image_row_tiles_fmt(x3, M, K, order, out, LPM_OPERAND_BF16X3, stream);
}
extern "C" int lpm_image_row_tiles_fmt(const void* x3, int M, int K, int order, void* out, int kind, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(operand_kind_ok(kind), LPM_ERR_BADARG, "lpm_image_row_tiles: unknown operand format %d", kind);
    LPM_REQUIRE(x3 && out, LPM_ERR_BADARG, "lpm_image_row_tiles: null pointer");
    LPM_REQUIRE(M > 0 && K > 0 && K % 16 == 0 && (((uintptr_t)x3 | (uintptr_t)out) & 15) == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_image_row_tiles: need K %% 16 == 0 and 16-byte aligned pointers (K=%d)", K);
    const int MT = row_tiles_per_clip(M);
    const int64_t total = (int64_t)MT * (K / 16) * 64;
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(image_row_tiles_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x3, (int64_t)M, K, (operand_kind_planes(kind) == 3 && order) ? 2 * K : K, MT, (uint4*)out,
                       operand_kind_planes(kind));
    return check_launch("lpm_image_row_tiles");
}

extern "C" int lpm_assign_gemm_tiles_bwd_dx(const void* dlr, const void* wtt, int B, int T, int D, int K, float* dx, int64_t lddx,
                                            lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(dlr && wtt && dx, LPM_ERR_BADARG, "lpm_assign_gemm_tiles_bwd_dx: null pointer");
    LPM_REQUIRE(B > 0 && lpm_assign_gemm_tiles_supported(T, D, K) && lddx >= D, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_assign_gemm_tiles_bwd_dx: need D %% 32 == 0, K %% 32 == 0, K <= 512 (D=%d K=%d)", D, K);
    const int MT = row_tiles_per_clip(T), KS = K / 16, NT = D / 32;
    TileGemmArgs g{};
    g.a = (const uint4*)dlr; g.a_tile = (int64_t)KS * 128; g.a_step = 128; g.a_batch = (int64_t)MT * KS * 128; g.a_tiles = MT;
    g.b = (const uint4*)wtt; g.b_tile = 128; g.b_step = (int64_t)NT * 128; g.b_batch = 0; g.b_tiles = NT;
    g.rb_per_batch = MT / 2; g.steps_per_split = KS; g.total_steps = KS;
    g.out = dx; g.ldo = lddx; g.out_batch = (int64_t)T * lddx; g.out_split = 0;
    g.rows_valid = T; g.cols_valid = D; g.accumulate = 1; g.stats = nullptr;
    return tg_launch<TG_EPI_STORE>(g, B, 1, (hipStream_t)stream, "lpm_assign_gemm_tiles_bwd_dx");
}

extern "C" size_t lpm_assign_gemm_tiles_bwd_dw_workspace_bytes(int B, int T, int D, int K) {
    const int z2 = lpm::dw_splits(B, T, D, K, 2), z1 = lpm::dw_splits(B, T, D, K, 1);      // split-bf16 / plain-bf16 tile forms
    return (size_t)(z1 > z2 ? z1 : z2) * D * K * sizeof(float);
}

static int assign_gemm_tiles_bwd_dw_impl(const void* xt, const void* dlt, int B, int T, int D, int K, float* dW, void* workspace,
                                         size_t workspace_bytes, int planes, lpm_stream_t stream);
extern "C" int lpm_assign_gemm_tiles_bwd_dw(const void* xt, const void* dlt, int B, int T, int D, int K, float* dW, void* workspace,
                                            size_t workspace_bytes, lpm_stream_t stream) {
    return assign_gemm_tiles_bwd_dw_impl(xt, dlt, B, T, D, K, dW, workspace, workspace_bytes, 2, stream);
}
// bf16 storage: xt, dlt plain bf16 frame tiles with lpm_frame_steps_bf16(T) steps per clip (zero beyond T)
extern "C" int lpm_assign_gemm_tiles_bwd_dw_bf16(const void* xt, const void* dlt, int B, int T, int D, int K, float* dW, void* workspace,
                                                 size_t workspace_bytes, lpm_stream_t stream) {
    return assign_gemm_tiles_bwd_dw_impl(xt, dlt, B, T, D, K, dW, workspace, workspace_bytes, 1, stream);
}
extern "C" int lpm_frame_steps_bf16(int T) { return lpm::frame_steps(T, 1); }

static int assign_gemm_tiles_bwd_dw_impl(const void* xt, const void* dlt, int B, int T, int D, int K, float* dW, void* workspace,
                                         size_t workspace_bytes, int planes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(xt && dlt && dW && workspace, LPM_ERR_BADARG, "lpm_assign_gemm_tiles_bwd_dw: null pointer");
    LPM_REQUIRE(B > 0 && lpm_assign_gemm_tiles_supported(T, D, K), LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_assign_gemm_tiles_bwd_dw: need D %% 32 == 0, K %% 32 == 0, K <= 512 (D=%d K=%d)", D, K);
    LPM_REQUIRE(workspace_bytes >= lpm_assign_gemm_tiles_bwd_dw_workspace_bytes(B, T, D, K), LPM_ERR_BADARG,
                "lpm_assign_gemm_tiles_bwd_dw: workspace too small");
    const int S = frame_steps(T, planes), DT = D / 32, KT = K / 32;
    const int Z = dw_splits(B, T, D, K, planes), steps = B * S;
    const int64_t U = 64 * planes;
    TileGemmArgs g{};
    g.a = (const uint4*)xt; g.a_tile = U; g.a_step = (int64_t)DT * U; g.a_batch = 0; g.a_tiles = DT;
    g.b = (const uint4*)dlt; g.b_tile = U; g.b_step = (int64_t)KT * U; g.b_batch = 0; g.b_tiles = KT;
    g.rb_per_batch = (D + 63) / 64; g.steps_per_split = (steps + Z - 1) / Z; g.total_steps = steps;
    if (planes == 1 && (g.steps_per_split & 1)) ++g.steps_per_split;          // a ring stage carries two steps: even split boundaries
    g.out = Z > 1 ? (float*)workspace : dW; g.ldo = K; g.out_batch = 0; g.out_split = (int64_t)D * K;
    g.rows_valid = D; g.cols_valid = K; g.accumulate = 0; g.stats = nullptr;
    const int Zeff = (steps + g.steps_per_split - 1) / g.steps_per_split;     // every launched split has >= 1 step
    const int rc = tg_launch<TG_EPI_STORE>(g, 1, Zeff, (hipStream_t)stream, "lpm_assign_gemm_tiles_bwd_dw", 0, 0, 0, planes);
    if (rc != LPM_OK || Z == 1) return rc;
    const int64_t n4 = (int64_t)D * K / 4;
    hipLaunchKernelGGL(tg_reduce_splits_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)workspace, Zeff, n4, (float4*)dW);
    return check_launch("lpm_assign_gemm_tiles_bwd_dw");
}

// dW[N1, N2] = X^T . DY for a skinny batch: X [R, N1], DY [R, N2] fp32, R (the batch, a multiple of 16) is the whole
// reduction.  Both operands are weight tiles (lpm_split_weight_tiles of X and DY, not transposed): xt [R/16][N1/32],
// dyt [R/16][N2/32].  The hidden1 weight gradient of the NetVLAD models (frame_level_models.py:2314-2319 backward) at
// cfg-2 is N1 = 270336, N2 = 512, R = 80: write-bound (554 MB), written once, straight into the caller's buffer.
extern "C" int lpm_skinny_weight_grad_tiles(const void* xt, const void* dyt, int R, int N1, int N2, float* dW, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(xt && dyt && dW, LPM_ERR_BADARG, "lpm_skinny_weight_grad_tiles: null pointer");
    LPM_REQUIRE(R > 0 && R % 16 == 0 && N1 > 0 && N2 > 0 && N2 % 32 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_skinny_weight_grad_tiles: need R %% 16 == 0 and N2 %% 32 == 0 (R=%d N2=%d)", R, N2);
    const int NT1 = (N1 + 31) / 32, NT2 = N2 / 32;
    TileGemmArgs g{};
    g.a = (const uint4*)xt; g.a_tile = 128; g.a_step = (int64_t)NT1 * 128; g.a_batch = 0; g.a_tiles = NT1;
    g.b = (const uint4*)dyt; g.b_tile = 128; g.b_step = (int64_t)NT2 * 128; g.b_batch = 0; g.b_tiles = NT2;
    g.rb_per_batch = (N1 + 63) / 64; g.steps_per_split = R / 16; g.total_steps = R / 16;
    g.out = dW; g.ldo = N2; g.rows_valid = N1; g.cols_valid = N2;
    static const int nt = [] { const char* e = getenv("LPM_DW_NT_STORE"); return (e && e[0] == '0') ? 0 : 1; }();      // 0: plain stores (A/B)
    g.nt_store = nt;
    // the 4 (cfg-2) / 8 (cfg-5) column blocks of a row block as neighbours on one XCD: the activation tiles (86 MB at cfg-2, 277 MB at
    // cfg-5: more than the infinity cache) are fetched from HBM once instead of once per column block (LPM_DW_COLS_INNER=0: A/B)
    static const int inner = [] { const char* e = getenv("LPM_DW_COLS_INNER"); return (e && e[0] == '0') ? 0 : 1; }();
    g.cols_inner = inner;
    // 5 reduction steps and a 554 MB store: narrow column blocks (36 KB of LDS, 4 workgroups per CU) so that one workgroup's
    // store overlaps its neighbours' loads (measured at cfg-2: 145 us with 128-column blocks, 215 us with 512)
    return tile_gemm_store(g, 1, 1, (hipStream_t)stream, "lpm_skinny_weight_grad_tiles", 1);
}
