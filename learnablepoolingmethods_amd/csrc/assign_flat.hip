// K1, the soft-assignment GEMM (frame_level_models.py:2781-2789: tf.matmul(reshaped_input, cluster_weights) + the batch statistics
// of cluster_bn), on FLAT 96-row workgroups (round 4).
//
// The 128-row tile GEMM form (tile_gemm.hip, MW = 2) walks the per-clip row tiles four at a time: 80 clips x 10 padded tiles = 800
// tiles = 200 workgroups on 256 CUs, four tile-times on the critical path, 6 % of the rows padding, and every wave reads 8 KB of
// fragments out of LDS per 12 MFMAs (64 KB of reads + 24 KB of LDS-DMA writes per step and CU against 768 matrix-pipe cycles: the LDS
// is as busy as the matrix pipe).  Here:
//   * the rows of all clips are ONE flat sequence cut into 96-row groups (24 000 rows = 250 workgroups, no padding, three tile-times);
//     the operand is still the per-clip row-tile image K3's per-clip GEMM needs (lpm_split_rows_tiles) -- an LDS-DMA load takes one
//     global address PER LANE, so a flat tile is gathered from the one or two per-clip tiles it straddles at no cost;
//   * wave w owns column tile w (32 clusters) x three row tiles: the A fragments are shared by all eight waves and go through a ring
//     of 6 KB stages in LDS; the B fragments of a wave are its own and nobody else's -- they go from L2 STRAIGHT INTO REGISTERS
//     (global_load_dwordx4, four steps ahead), never through LDS: 48 KB of fragment reads + 6 KB of DMA writes per step and CU against
//     576 matrix-pipe cycles;
//   * the epilogue takes the column statistics from the accumulators and stores the accumulators as they are (a lane's 16 values of one
//     column: 32 lanes = one 128-byte line of a row per store) -- no staging tile, no workgroup barrier.
// PLAIN (round 5, bf16 storage: BASELINE configs[4]): the operands are plain bf16 tiles, one MFMA per product.  A plain tile buffer with
// D / 16 steps of 64 units IS a split buffer with D / 32 steps of 128 units whose "hi plane" is the even step and whose "lo plane" the
// odd one: the same loop runs on DOUBLE steps (32 reduction elements) with the products (A.h, B.h) and (A.l, B.l) -- six MFMAs per step
// instead of nine, the same loads, rings and hand-counted waits; the weight tiles' odd step sits NT tiles (not 1 KB) behind the even
// one, and the logits leave as bf16.
// LDS-DMA loads and register loads retire in order on one counter (vmcnt): every wave issues them in a fixed order -- per step
// [A piece of step s + NS (waves 0-5)] [B hi, B lo of step s + DB] -- and waits with hand-counted immediates; the ring and the registers
// are read by inline assembly behind those waits (the compiler puts vmcnt(0) in front of LDS reads it can see next to an LDS-DMA).
#include <atomic>
#include "tile_gemm.h"

namespace lpm {

// Process-wide mask of K1 forward forms switched OFF at run time (lpm_k1_forms_disable; bit 0: the flat 96-row forms, bit 1: the
// 160 x 512 plain-bf16 form).  The hand-scheduled forms hand registers to asynchronous loads behind hand-counted waits (below): the
// host side checks each against the tile-GEMM form once per process on a small problem (ops._k1_selfcheck) and switches a
// form off -- loudly -- if a toolchain change ever makes it disagree; the ISA scan of tests/test_build_flags.py is the build-time half.
static std::atomic<int> g_k1_forms_off{0};

constexpr int AF_KSTEP = 6 * 1024;         // A bytes per reduction step: three row tiles x (hi, lo)
constexpr int AF_ROWS = 96;
// A ring: NS stages of KB reduction steps each, ONE workgroup barrier per stage (LPM_K1_KB).  Measured at cfg-2's shape, kernel alone:
// 38.5 / 38.0 / 40.8 us with 1 / 2 / 4 steps per barrier -- the barrier is not what the loop waits for; two is the default.
__host__ __device__ constexpr int af_ns(int KB) { return KB == 1 ? 6 : 4; }
__host__ __device__ constexpr int af_tail(int KB, int DB) { return KB == 4 ? (DB == 8 ? 16 : 12) : 8; }   // peeled last steps: a multiple of DB and KB, >= (NS - 1) KB

struct AssignFlatArgs {
    const uint4* xr;           // per-clip row tiles [b][mt][ds][plane][lane]
    const uint4* wt;           // weight tiles [ds][nt][plane][lane]
    int M, T, MT, DS, NT, K;   // rows B*T, frames per clip, row tiles per clip, reduction steps D/16, column tiles K/32, clusters
    float* logits;             // [M, K]  (PLAIN: bf16 storage)
    float* stats;              // [nblk][2][K]: rows < gridDim.x written, the rest zeroed
    int nblk;
};
__device__ __forceinline__ unsigned af_bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__host__ __device__ constexpr int af_mod(int u, int n) { return ((u % n) + n) % n; }
// VMEM operations a wave issues in step u, counted from the start of the tail (u < 0: the steady state, and the prologue's virtual
// steps): at the first step of a stage the KB pieces of the stage NS - 1 ahead (waves 0-5), at the end of every step the two B loads
// of step u + DB
template <int KB, int DB>
__host__ __device__ constexpr int af_ops(int u, bool loader) {
    return ((loader && af_mod(u, KB) == 0 && u + (af_ns(KB) - 1) * KB < af_tail(KB, DB)) ? KB : 0) + (u + DB < af_tail(KB, DB) ? 2 : 0);
}
// ... and the vmcnt immediate at the top of step i: B(i) is in the wave's registers when at most this many younger operations are
// outstanding.  The wave's pieces of the NEXT stage, which the barrier at the top of a stage promises to everyone, were requested
// (NS - 2) KB >= DB steps ago, i.e. before B(i): they have landed as well (loads retire in order).
template <int KB, int DB>
__host__ __device__ constexpr int af_wait(int i, bool loader) {
    int w = 0;
    for (int u = i - DB + 1; u < i; ++u) w += af_ops<KB, DB>(u, loader);
    return w;
}
static_assert(af_wait<1, 4>(-400, true) == 9 && af_wait<1, 4>(-400, false) == 6, "steady-state waits");

// (the wait itself carries no operands: waves 0-5 and 6-7 wait with different immediates, and a tied operand in each arm of that branch
// made the compiler copy the registers in front of one arm's wait -- before the load had written them.  The registers are handed over by
// one empty statement behind the branch instead.)
#define AF_WAIT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : : : "memory"); break;
__device__ __forceinline__ void af_wait_vm(int n) {          // n folds to a constant where this is called
    switch (n) {
        AF_WAIT_CASE(1) AF_WAIT_CASE(2) AF_WAIT_CASE(3) AF_WAIT_CASE(4) AF_WAIT_CASE(5) AF_WAIT_CASE(6) AF_WAIT_CASE(7) AF_WAIT_CASE(8)
        AF_WAIT_CASE(9) AF_WAIT_CASE(10) AF_WAIT_CASE(11) AF_WAIT_CASE(12) AF_WAIT_CASE(13) AF_WAIT_CASE(14) AF_WAIT_CASE(15)
        AF_WAIT_CASE(16) AF_WAIT_CASE(17) AF_WAIT_CASE(18) AF_WAIT_CASE(19) AF_WAIT_CASE(20) AF_WAIT_CASE(21) AF_WAIT_CASE(22)
        AF_WAIT_CASE(23) AF_WAIT_CASE(24) AF_WAIT_CASE(25) AF_WAIT_CASE(26) AF_WAIT_CASE(27) AF_WAIT_CASE(28) AF_WAIT_CASE(29) AF_WAIT_CASE(30)
        AF_WAIT_CASE(31) AF_WAIT_CASE(32) AF_WAIT_CASE(33) AF_WAIT_CASE(34) AF_WAIT_CASE(35) AF_WAIT_CASE(36)
        default: asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); break;
    }
}
#undef AF_WAIT_CASE

template <int KB, int DB, bool AF_LUMP, bool PLAIN = false>
__global__ __launch_bounds__(512, 1) void assign_flat_kernel(const AssignFlatArgs a) {
    static_assert(!(PLAIN && AF_LUMP), "the plain-bf16 form has one loop order");
    constexpr int NS = af_ns(KB), TAIL = af_tail(KB, DB), STAGE = KB * AF_KSTEP;
    static_assert((NS - 2) * KB >= DB && TAIL % DB == 0 && TAIL % KB == 0 && TAIL >= (NS - 1) * KB && (DB == 4 || DB == 8),
                  "A(next stage) must be older than B(i); the tail is whole register rounds and whole stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wg = blockIdx.x, cb = blockIdx.y;
    const int ct = cb * 8 + wave;                                  // this wave's column tile
    const bool loader = wave < 6;
    const int nstep = a.DS;

    // A piece of this wave (waves 0-5): row tile m_ld, plane p_ld of the group; lane (row l31, reduction half) -> its row of the per-clip
    // image (rows past the end: the last row again -- masked in the epilogue)
    const uint4* asrc;
    {
        const int m_ld = wave >> 1, p_ld = wave & 1;
        const unsigned R = (unsigned)wg * AF_ROWS + m_ld * 32 + l31;                 // (B T < 2^31: assign_flat_ok)
        const unsigned Rc = R < (unsigned)a.M ? R : (unsigned)a.M - 1u;
        const int b = (int)(Rc / (unsigned)a.T), t = (int)(Rc - (unsigned)b * (unsigned)a.T);
        asrc = a.xr + ((int64_t)b * a.MT + (t >> 5)) * a.DS * 128 + p_ld * 64 + half * 32 + (t & 31);
    }
    const uint4* bbase = a.wt + (int64_t)ct * (PLAIN ? 64 : 128);   // wave-uniform; + step * NT * 128
    const int64_t bstep = (int64_t)a.NT * 128;                     // (PLAIN: a double step = two plain steps of NT * 64 units)
    const int64_t blo = PLAIN ? (int64_t)a.NT * 64 : 64;           // the second B piece of a step: the odd plain step / the lo plane
    const unsigned boff = (unsigned)lane * 16u;
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned rd_lane = smem_lds + (unsigned)lane * 16u;

    struct Frag { tg_u32x4 h[3], l[3]; };
    Frag fa[2];
    tg_u32x4 bh[DB], bl[DB];
    f32x16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    auto issue_a = [&](int q, int slot) {                          // stage q (steps q KB ..) -> ring slot; loader waves only
#pragma unroll
        for (int e = 0; e < KB; ++e)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + (int64_t)(q * KB + e) * 128),
                                             (__attribute__((address_space(3))) void*)(smem + slot * STAGE + e * AF_KSTEP + wave * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int s, tg_u32x4& h, tg_u32x4& l) {
        const uint4* p = bbase + (int64_t)s * bstep;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(h) : "v"(boff), "s"(p) : "memory");
        if (PLAIN) {
            const uint4* p2 = p + blo;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(l) : "v"(boff), "s"(p2) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(l) : "v"(boff), "s"(p) : "memory");
        }
    };
    auto read_frags = [&](unsigned byte_off, Frag& f) {            // byte_off: slot * STAGE + (step within the stage) * AF_KSTEP
        const unsigned ad = rd_lane + byte_off;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.h[m]) : "v"(ad), "n"((m * 2 + 0) * 1024) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.l[m]) : "v"(ad), "n"((m * 2 + 1) * 1024) : "memory");
        }
    };
    auto frags_ready = [&](Frag& f) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.h[0]), "+v"(f.h[1]), "+v"(f.h[2]), "+v"(f.l[0]), "+v"(f.l[1]), "+v"(f.l[2]) : : "memory");
    };

    // prologue: the operations of the virtual steps -(NS - 1) KB .. -1, in the loop's order (stages 0 .. NS - 2, B(0 .. DB - 1))
    static_assert((NS - 1) * KB >= DB, "the prologue's virtual steps cover the B loads");
#pragma unroll
    for (int u = -(NS - 1) * KB; u < 0; ++u) {
        if (loader && af_mod(u, KB) == 0) issue_a(u / KB + NS - 1, u / KB + NS - 1);
        if (u + DB >= 0) issue_b(u + DB, bh[u + DB], bl[u + DB]);
    }
    // stage 0 (the wave's first KB pieces): everything younger may fly
    if (loader) af_wait_vm((NS - 2) * KB + 2 * DB);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, fa[0]);

    int slot = 0;                                                  // (s / KB) % NS
    // one step: i = its index in the tail (< 0: steady state, same residues mod DB and KB), j = s % DB (B register set), ab = s & 1
    auto step = [&](int s, int i, int j, int ab) __attribute__((always_inline)) {
        const int e = af_mod(i, KB);                               // step within its stage
        if (loader) af_wait_vm(af_wait<KB, DB>(i, true));
        else af_wait_vm(af_wait<KB, DB>(i, false));
        asm volatile("" : "+v"(bh[j]), "+v"(bl[j]) : : "memory");          // B(s) is in these registers from here on
        frags_ready(fa[ab]);
        if (e == 0) {
            __builtin_amdgcn_s_barrier();                          // the next stage is in LDS for everyone; the previous one has been read
            asm volatile("" ::: "memory");
        }
        // ONE non-MFMA operation between consecutive MFMAs: the two waves of a SIMD leave the barrier together and run the same
        // instruction stream in phase, so a LUMP of seven LDS operations behind the first MFMA group (the first form of this loop) is a
        // lump for both -- the matrix pipe idles for its length, 15 % of the step.  A single read in front of an MFMA is covered by the
        // neighbour's MFMA (AF_LUMP, LPM_K1_LUMP=1: the first form, A/B).
        if constexpr (PLAIN) {
            // six MFMAs per double step: (A even step, B even step) x 3 row tiles, then the odd pair; the six fragment reads of the next
            // step one in front of each MFMA but the first (which carries the DMA issue), the B loads when their registers are free
            const int slot1 = slot + 1 == NS ? 0 : slot + 1;
            const bool rd = i + 1 < TAIL;
            const unsigned ad = rd_lane + (unsigned)((e + 1 == KB ? slot1 : slot) * STAGE + (e + 1 == KB ? 0 : e + 1) * AF_KSTEP);
            Frag& fn = fa[ab ^ 1];
            const int cur_slot = slot;
            if (e + 1 == KB) slot = slot1;
#define AF_RD(REG, OFF) if (rd) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ad), "n"(OFF) : "memory")
            acc[0] = tg_mfma(fa[ab].h[0], bh[j], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (e == 0 && loader && (i + (NS - 1) * KB < TAIL))
                issue_a(s / KB + NS - 1, cur_slot == 0 ? NS - 1 : cur_slot - 1);
            AF_RD(fn.h[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = tg_mfma(fa[ab].h[1], bh[j], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            AF_RD(fn.l[0], 1024);
            acc[2] = tg_mfma(fa[ab].h[2], bh[j], acc[2]);
            __builtin_amdgcn_sched_barrier(0);
            const bool more_b = i + DB < TAIL;
            const uint4* pb = bbase + (int64_t)(s + DB) * bstep;
            if (more_b) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bh[j]) : "v"(boff), "s"(pb) : "memory");
            AF_RD(fn.h[1], 2048);
            acc[0] = tg_mfma(fa[ab].l[0], bl[j], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            AF_RD(fn.l[1], 3072);
            acc[1] = tg_mfma(fa[ab].l[1], bl[j], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            AF_RD(fn.h[2], 4096);
            acc[2] = tg_mfma(fa[ab].l[2], bl[j], acc[2]);
            __builtin_amdgcn_sched_barrier(0);
            AF_RD(fn.l[2], 5120);
#undef AF_RD
            const uint4* pb2 = pb + blo;
            if (more_b) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bl[j]) : "v"(boff), "s"(pb2) : "memory");
            return;
        }
        acc[0] = tg_mfma(fa[ab].h[0], bh[j], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (e == 0 && loader && (i + (NS - 1) * KB < TAIL))
            issue_a(s / KB + NS - 1, slot == 0 ? NS - 1 : slot - 1);       // into the slot of the stage before this one
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = tg_mfma(fa[ab].h[1], bh[j], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[2] = tg_mfma(fa[ab].h[2], bh[j], acc[2]);
        __builtin_amdgcn_sched_barrier(0);
        const int slot1 = slot + 1 == NS ? 0 : slot + 1;
        const bool rd = i + 1 < TAIL;
        const unsigned ad = rd_lane + (unsigned)((e + 1 == KB ? slot1 : slot) * STAGE + (e + 1 == KB ? 0 : e + 1) * AF_KSTEP);
        Frag& fn = fa[ab ^ 1];
        if (e + 1 == KB) slot = slot1;
#define AF_RD(REG, OFF) if (rd) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ad), "n"(OFF) : "memory")
        if (AF_LUMP) { AF_RD(fn.h[0], 0); AF_RD(fn.l[0], 1024); AF_RD(fn.h[1], 2048); AF_RD(fn.l[1], 3072); AF_RD(fn.h[2], 4096); AF_RD(fn.l[2], 5120); }
        __builtin_amdgcn_sched_barrier(0);
        // (the three products that read the hi plane of B first: its registers are free for the load of step s + DB three MFMAs earlier)
        if (!AF_LUMP) AF_RD(fn.h[0], 0);
        acc[0] = tg_mfma(fa[ab].l[0], bh[j], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (!AF_LUMP) AF_RD(fn.l[0], 1024);
        acc[1] = tg_mfma(fa[ab].l[1], bh[j], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (!AF_LUMP) AF_RD(fn.h[1], 2048);
        acc[2] = tg_mfma(fa[ab].l[2], bh[j], acc[2]);
        __builtin_amdgcn_sched_barrier(0);
        const bool more_b = i + DB < TAIL;
        const uint4* pb = bbase + (int64_t)(s + DB) * bstep;
        if (!AF_LUMP && more_b) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bh[j]) : "v"(boff), "s"(pb) : "memory");
        if (!AF_LUMP) AF_RD(fn.l[1], 3072);
        acc[0] = tg_mfma(fa[ab].h[0], bl[j], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (!AF_LUMP) AF_RD(fn.h[2], 4096);
        acc[1] = tg_mfma(fa[ab].h[1], bl[j], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (!AF_LUMP) AF_RD(fn.l[2], 5120);
        acc[2] = tg_mfma(fa[ab].h[2], bl[j], acc[2]);
        __builtin_amdgcn_sched_barrier(0);
#undef AF_RD
        if (!AF_LUMP && more_b) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(bl[j]) : "v"(boff), "s"(pb) : "memory");
        if (AF_LUMP && more_b) issue_b(s + DB, bh[j], bl[j]);
    };
    const int nmain = nstep - TAIL;
    for (int s0 = 0; s0 < nmain; s0 += DB) {
#pragma unroll
        for (int j = 0; j < DB; ++j) step(s0 + j, -400 + j, j, j & 1);
    }
#pragma unroll
    for (int i = 0; i < TAIL; ++i) step(nmain + i, i, i % DB, i & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // epilogue: acc[m][r] = logits[row wg * 96 + m * 32 + mfma32_row(r, lane)][column ct * 32 + l31]
    const int col = ct * 32 + l31;
    const int64_t row0 = (int64_t)wg * AF_ROWS + 4 * half;
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + m * 32 + (r & 3) + 8 * (r >> 2);
            const float v = row < a.M ? acc[m][r] : 0.f;
            cs += v;
            cq = fmaf(v, v, cq);
            if (PLAIN) {
                if (row < a.M) reinterpret_cast<unsigned short*>(a.logits)[row * a.K + col] = (unsigned short)af_bf16_rne(acc[m][r]);
            } else {
                if (row < a.M) a.logits[row * a.K + col] = acc[m][r];
            }
        }
    cs += __shfl_xor(cs, 32, 64);
    cq += __shfl_xor(cq, 32, 64);
    if (lane < 32) {
        float* p = a.stats + (int64_t)wg * 2 * a.K;
        p[col] = cs;
        p[a.K + col] = cq;
    }
    // the statistics rows nobody owns (the array is sized for the 64-row groups of the per-clip forms)
    for (int r = gridDim.x + wg; r < a.nblk; r += gridDim.x)
        a.stats[(int64_t)r * 2 * a.K + (tid >> 8) * a.K + cb * 256 + (tid & 255)] = 0.f;
}

// ---- plain-bf16 K1 on 160-row x 512-column workgroups: one round of the chip (round 5; BASELINE configs[4]) -------------------------
// The flat form above gives every wave ONE column tile and the group's three row tiles: per double step the eight waves read the same
// 6 KB of A fragments out of LDS -- 48 KB = 384 LDS cycles against 386 matrix-pipe cycles, with 22 KB through the vector L1 on top
// (352 cycles at 64 B / clk): three resources within 10 % of each other, and 800 workgroups are 3.1 rounds of 256 CUs (4 rounds at 78 %).
// Measured 51 us = 0.31 of the bf16 peak at cfg-5 (38 400 x 1024 x 512).
// Here a workgroup owns 160 rows x all 512 columns -- cfg-5 is 240 workgroups: ONE round at 94 % -- and wave w the column tiles
// {2 w, 2 w + 1} x all five row tiles: per plain step (16 reduction elements) ten MFMAs (320 matrix-pipe cycles per wave, 640 per SIMD)
// against five fragment reads per wave (40 KB = 320 LDS cycles per CU: half) and 2 KB of B per wave straight from L2 (16 + 5 KB per
// step through the vector L1: 330 cycles: half); A leaves HBM once instead of once per column block.
//   A: the per-clip row tiles gathered per lane by LDS-DMA (as above) into a ring of three 40 KB stages of eight plain steps; wave w
//      loads the 1 KB piece of plain step w of the stage for each of the five row tiles: every wave issues the same five loads per
//      stage, so ONE set of hand-counted vmcnt immediates serves the workgroup.  LDS layout [slot][plain step][row tile][lane].
//   fragments: EIGHT registers for five tiles.  The 40 (step, tile) pairs of a stage take registers n mod 8 in turn; behind the two
//      MFMAs of pair n the fragment of pair n + 5 (the same tile, next step) is read into register (n + 5) mod 8 -- the one pair n - 3
//      used, six MFMAs back.  Five reads are always in flight and every pair waits with lgkmcnt(4).  (The first build had five
//      registers and read the next fragment into the register the MFMA in front of it had just been given: with two waves queueing on a
//      SIMD's matrix core that MFMA may not have started when the LDS answers -- every second column tile came out wrong once the
//      timing shifted (seen when one DMA piece per stage was removed: odd column tiles wrong in every row tile, the fifth row tile
//      intermittently; the same build with one more piece in flight passed every test).  A register is rewritten here only when three
//      younger MFMA pairs of the same wave have been issued behind its last reader: the core takes a wave's MFMAs in order, so that
//      reader started >= 5 x 32 cycles earlier.  tests/test_build_flags.py checks the register flow of the built loop.)
//   stages: at the top of a stage's LAST step the wave drains its reads (lgkmcnt(0): that step's fragments are in registers), the
//      barrier says the stage's slot is free and the next stage has landed for everyone, and the stage three ahead is requested.
//   B: two 1 KB fragments per plain step straight into registers, eight steps ahead (L2: the 30 workgroups of an XCD walk B together).
constexpr int AW_MT = 5, AW_ROWS = AW_MT * 32;
constexpr int AW_STEP = AW_MT * 1024;      // A bytes per plain step
constexpr int AW_KB = 8;                   // plain steps per stage (= waves: wave w loads step w's pieces)
constexpr int AW_STAGE = AW_KB * AW_STEP;
constexpr int AW_NS = 3, AW_DB = 4, AW_FR = 8;
static_assert((AW_KB * AW_MT) % AW_FR == 0 && AW_FR >= AW_MT + 2, "the fragment registers rotate with the period of a stage");
constexpr int AW_TAIL = 3 * AW_KB;         // the last three stages request no further stage
constexpr int AW_EW = 144;                 // bytes between the rows of a wave's epilogue tile (128 of bf16 logits + 16)
// VMEM operations of step u (tail-relative; u < 0: steady state and the prologue's virtual steps), in issue order: the five pieces of
// the stage three ahead at a stage's last step, the two B loads of step u + DB at its end
constexpr int AW_DMA = AW_MT;             // DMA pieces per wave and stage
__host__ __device__ constexpr int aw_ops(int u) { return ((af_mod(u, AW_KB) == AW_KB - 1 && u < 0) ? AW_DMA : 0) + (u + AW_DB < AW_TAIL ? 2 : 0); }
// B(i) was the last thing step i - DB issued: it has landed when at most the operations of the steps in between are outstanding (and
// with it everything older: the pieces the barrier promises were requested sixteen steps earlier)
__host__ __device__ constexpr int aw_wait(int i) {
    int w = 0;
    for (int u = i - AW_DB + 1; u < i; ++u) w += aw_ops(u);
    return w;
}
static_assert(AW_KB % AW_DB == 0 && AW_TAIL >= AW_DB, "B register sets by step within the stage");
static_assert(aw_wait(-800) == (AW_DB - 1) * 2 + AW_DMA && aw_wait(-800 + AW_KB - 1) == (AW_DB - 1) * 2 &&
              aw_wait(0) == (AW_DB - 1) * 2 + AW_DMA && aw_wait(AW_TAIL - 1) == 0, "steady-state and tail waits");

// DBG (timing experiments, -DLPM_K1_WIDE_EXPERIMENTS builds only, LPM_K1_WIDE_DBG=n; results are garbage): 1 no epilogue, 2 no B fragment
// loads, 4 no DMA, 8 no MFMAs, 16 no fragment reads, 64 no logits stores, 128 non-temporal logits stores
template <int DBG>
__global__ __launch_bounds__(512, 1) void assign_wide_kernel(const AssignFlatArgs a) {
    constexpr int KB = AW_KB, NS = AW_NS, DB = AW_DB, TAIL = AW_TAIL, STAGE = AW_STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wg = blockIdx.x, cb = blockIdx.y;
    const int ct0 = cb * 16 + wave * 2;                            // this wave's two column tiles
    const int nstage = a.DS / 4;                                   // a.DS = double steps; a stage = four of them

    // a lane's source unit (16 bytes) in row tile m: row l31 of the tile, reduction half `half` (rows past the end: the last row again --
    // masked in the epilogue).  Recomputed for every stage (five divisions per eight steps, beside 80 MFMAs): five registers this kernel
    // does not have -- with them resident the compiler spilled, and a scratch reload inside the loop is a VMEM operation the hand-counted
    // waits do not know.
    auto a_unit = [&](int m, int wg_) -> unsigned {
        const unsigned R = (unsigned)wg_ * AW_ROWS + m * 32 + l31;
        const unsigned Rc = R < (unsigned)a.M ? R : (unsigned)a.M - 1u;
        const unsigned b = Rc / (unsigned)a.T, t = Rc - b * (unsigned)a.T;
        return (b * (unsigned)a.MT + (t >> 5)) * (unsigned)a.DS * 128u + (unsigned)half * 32u + (t & 31u);
    };
    const uint4* abase = a.xr + wave * 64;                         // plain step w of a stage: double step w >> 1, parity w & 1
    const uint4* bbase = a.wt + (int64_t)ct0 * 64;                 // + plain step * NT * 64
    const int64_t bstep = (int64_t)a.NT * 64;
    const unsigned boff = (unsigned)lane * 16u;
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned rd_lane = smem_lds + (unsigned)lane * 16u;

    tg_u32x4 fr[AW_FR];
    tg_u32x4 fb[DB][2];
    f32x16 acc[AW_MT][2];
#pragma unroll
    for (int m = 0; m < AW_MT; ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;

    auto issue_stage = [&](int q, int slot) {
        const uint4* src = abase + (int64_t)q * 512;
        if (DBG & 4) return;
        int wg_ = wg;
        asm volatile("" : "+s"(wg_));                               // (not loop-invariant as far as the optimiser can tell)
#pragma unroll
        for (int m = 0; m < AW_MT; ++m)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a_unit(m, wg_)),
                                             (__attribute__((address_space(3))) void*)(smem + slot * STAGE + wave * AW_STEP + m * 1024), 16, 0, 0);
    };
    auto load_b = [&](int s, int j) {
        const uint4* p = bbase + (int64_t)s * bstep;
        if (DBG & 2) return;
        // ("+v": the load lands in the register the step's MFMAs have just read -- L2 is a microsecond away -- and, the old value being
        // alive up to here, the allocator cannot give that register to the fragment read in front of this load)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(fb[j][0]) : "v"(boff), "s"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "+v"(fb[j][1]) : "v"(boff), "s"(p) : "memory");
    };
#pragma unroll
    for (int j = 0; j < DB; ++j) fb[j][0] = fb[j][1] = tg_u32x4{0u, 0u, 0u, 0u};

    // prologue: the operations of the virtual steps -3 KB .. -1 in the loop's order: stages 0, 1, B(0 .. DB - 2), stage 2, B(DB - 1)
    issue_stage(0, 0);
    issue_stage(1, 1);
#pragma unroll
    for (int j = 0; j < DB - 1; ++j) load_b(j, j);
    issue_stage(2, 2);
    load_b(DB - 1, DB - 1);
    af_wait_vm(2 * AW_DMA);                                        // stage 0's pieces have landed: at most two stages' worth is outstanding
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < AW_MT; ++m) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[m]) : "v"(rd_lane), "n"(m * 1024) : "memory");
#pragma unroll
    for (int m = AW_MT; m < AW_FR; ++m) fr[m] = tg_u32x4{0u, 0u, 0u, 0u};

    int slot = 0;                                                  // (s / KB) % NS
    // one plain step: i = its index in the tail (< 0: steady state, same residue mod KB), j = s % DB
    auto step = [&](int s, int i, int j) __attribute__((always_inline)) {
        const int e = af_mod(i, KB);
        const bool last_of_stage = e == KB - 1, has_next = i + 1 < TAIL;
        // (at a stage's last step at most ONE stage of pieces may be outstanding, whatever else is: the barrier's promise does not
        // lean on fragment loads and DMA pieces retiring in one common order)
        af_wait_vm(last_of_stage && aw_wait(i) > AW_DMA ? AW_DMA : aw_wait(i));
        asm volatile("" : "+v"(fb[j][0]), "+v"(fb[j][1]) : : "memory");      // B(s) is in these registers from here on
        const int nslot = slot + 1 == NS ? 0 : slot + 1;
        if (last_of_stage) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[(e * AW_MT + 0) % AW_FR]), "+v"(fr[(e * AW_MT + 1) % AW_FR]), "+v"(fr[(e * AW_MT + 2) % AW_FR]),
                         "+v"(fr[(e * AW_MT + 3) % AW_FR]), "+v"(fr[(e * AW_MT + 4) % AW_FR]) : : "memory");
            __builtin_amdgcn_s_barrier();                          // this stage's slot is free, the next stage is in LDS for everyone
            asm volatile("" ::: "memory");
        }
        // (the step's offset within the stage rides in the instruction's immediate: as part of the address the compiler kept eight
        // loop-invariant sums in registers it did not have, spilled them, and put vmcnt(0) behind every reload)
        const unsigned ad = rd_lane + (unsigned)((last_of_stage ? nslot : slot) * STAGE);
#pragma unroll
        for (int m = 0; m < AW_MT; ++m) {
            tg_u32x4& fa = fr[(e * AW_MT + m) % AW_FR];              // this pair's fragment
            tg_u32x4& fn = fr[(e * AW_MT + m + AW_MT) % AW_FR];      // the same tile of the next step: the register of three pairs ago
            if (has_next || m == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa) : : "memory");
            else if (m == 1) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa) : : "memory");
            else if (m == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa) : : "memory");
            else if (m == 3) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa) : : "memory");
            if (!(DBG & 8)) acc[m][0] = tg_mfma(fb[j][0], fa, acc[m][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (!(DBG & 8)) acc[m][1] = tg_mfma(fb[j][1], fa, acc[m][1]);
            __builtin_amdgcn_sched_barrier(0);
            if (m == 0 && last_of_stage && i < 0) issue_stage(s / KB + NS, slot);
            if (has_next && !(DBG & 16))
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fn) : "v"(ad), "n"((last_of_stage ? 0 : (e + 1) * AW_STEP) + m * 1024) : "memory");
            // the fragment of two pairs ago stays ALIVE up to here: without this the register allocator hands the read above the
            // register this pair's MFMAs have just been given (it knows of no hazard there) and the rotation is undone
            asm volatile("" : : "v"(fr[(e * AW_MT + m + AW_FR - 2) % AW_FR]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (last_of_stage) slot = nslot;
        if (i + DB < TAIL) load_b(s + DB, j);
    };
    const int nmain = (nstage - 3) * KB;
    for (int s0 = 0; s0 < nmain; s0 += KB) {
#pragma unroll
        for (int e = 0; e < KB; ++e) step(s0 + e, -800 + e, e % DB);
    }
#pragma unroll
    for (int i = 0; i < TAIL; ++i) step(nmain + i, i, i % DB);

    if (DBG & 1) {
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < AW_MT; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[m][c][r];
        if (t == 12345.f) a.stats[0] = 0.f;
        return;
    }
    // epilogue.  The MFMAs ran TRANSPOSED (srcA = the weight fragment, srcB = the frame fragment): acc[m][c][r] =
    // logits[row wg * 160 + m * 32 + l31][column (ct0 + c) * 32 + 8 (r >> 2) + 4 half + (r & 3)] -- a lane holds four CONSECUTIVE
    // columns of one row in four consecutive registers.  (The first build had the rows in the registers: 160 two-byte stores per wave,
    // 1 280 per CU at ~16 cycles of address processing each: 20 us of a 68 us kernel.  Round 5 stored the four columns as 8 bytes: 40
    // stores per wave, each instruction 32 sixteen-byte pieces 1 KB apart -- the 39 MB left at 3 TB/s.)  Round 6: the wave's 32 x 64
    // bf16 tile of row tile m goes through a wave-private LDS tile (the ring is free behind the barrier; rows AW_EW bytes apart) and
    // leaves as 16-byte stores, eight lanes to a row's 128 bytes: 20 store instructions per wave, every one writing eight whole lines.
    unsigned short* lg = reinterpret_cast<unsigned short*>(a.logits);
    float* sp = a.stats + (int64_t)wg * 2 * a.K;
    bool row_ok[AW_MT];
#pragma unroll
    for (int m = 0; m < AW_MT; ++m) row_ok[m] = (int64_t)wg * AW_ROWS + m * 32 + l31 < a.M;
    __syncthreads();                                               // everybody's last fragment reads are behind it: the ring is free
    {
        unsigned char* tile = smem + wave * (32 * AW_EW);
        const int prow = lane >> 3, piece = lane & 7;
#pragma unroll
        for (int m = 0; m < AW_MT; ++m) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const of_f2 v01 = {acc[m][c][4 * g + 0], acc[m][c][4 * g + 1]}, v23 = {acc[m][c][4 * g + 2], acc[m][c][4 * g + 3]};
                    uint2 w;
                    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, of_b2));
                    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, of_b2));
                    *reinterpret_cast<uint2*>(tile + l31 * AW_EW + c * 64 + g * 16 + half * 8) = w;
                }
            __builtin_amdgcn_wave_barrier();                       // (a wave's LDS operations complete in order: no s_barrier)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 8 + prow;
                const tg_u32x4 v = *reinterpret_cast<const tg_u32x4*>(tile + r * AW_EW + piece * 16);
                const int64_t row = (int64_t)wg * AW_ROWS + m * 32 + r;
                unsigned short* dst = lg + row * a.K + ct0 * 32 + piece * 8;
                if (row < a.M && !(DBG & 64)) {
                    if (DBG & 128) __builtin_nontemporal_store(v, reinterpret_cast<tg_u32x4*>(dst));
                    else *reinterpret_cast<tg_u32x4*>(dst) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float cs[16], cq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[r] = cq[r] = 0.f;
#pragma unroll
        for (int m = 0; m < AW_MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = row_ok[m] ? acc[m][c][r] : 0.f;
                cs[r] += v;
                cq[r] = fmaf(v, v, cq[r]);
            }
        }
        // column sums over the 32 lanes (rows) of a half: a halving exchange -- at distance 16, 8, 4, 2 a lane keeps one half of its
        // registers and receives the partner's copy of that half -- then one plain exchange at distance 1: 16 shuffles for 16 columns.
        // Lane l31 ends with register R = l31 >> 1: column 8 (R >> 2) + 4 half + (R & 3).  A fixed tree: the same bits every run.
#pragma unroll
        for (int lvl = 0; lvl < 4; ++lvl) {
            const int n = 8 >> lvl, d = 16 >> lvl;                  // registers kept, lane distance
            const bool up = (l31 & d) != 0;
#pragma unroll
            for (int r = 0; r < n; ++r) {
                // (opaque copies: the optimiser otherwise turns "up ? cs[r + n] : cs[r]" into a DYNAMIC index into the register array --
                // a 16-way compare / select chain per access, 2 000 v_cndmask in the first build of this epilogue: 11 us)
                float lo_s = cs[r], hi_s = cs[r + n], lo_q = cq[r], hi_q = cq[r + n];
                asm volatile("" : "+v"(lo_s), "+v"(hi_s), "+v"(lo_q), "+v"(hi_q));
                const float ks = up ? hi_s : lo_s, ss = up ? lo_s : hi_s;
                const float kq = up ? hi_q : lo_q, sq = up ? lo_q : hi_q;
                cs[r] = ks + __shfl_xor(ss, d, 64);
                cq[r] = kq + __shfl_xor(sq, d, 64);
            }
        }
        cs[0] += __shfl_xor(cs[0], 1, 64);
        cq[0] += __shfl_xor(cq[0], 1, 64);
        const int R = l31 >> 1;
        const int col = (ct0 + c) * 32 + 8 * (R >> 2) + 4 * half + (R & 3);
        sp[(l31 & 1) * a.K + col] = (l31 & 1) ? cq[0] : cs[0];
    }
    // the statistics rows nobody owns (the array is sized for the 64-row groups of the per-clip forms)
    for (int r = gridDim.x + wg; r < a.nblk; r += gridDim.x) {
        float* z = a.stats + (int64_t)r * 2 * a.K + cb * 512 + tid;
        z[0] = 0.f;
        z[a.K] = 0.f;
    }
}

// LPM_K1_WIDE=0: the flat 96-row plain form (A/B)
bool assign_wide_ok(int B, int T, int MT, int D, int K, int nblk) {
    static const int on = [] { const char* e = getenv("LPM_K1_WIDE"); return (e && e[0] == '0') ? 0 : 1; }();
    const int DS = D / 32;
    const int64_t M = (int64_t)B * T;
    return on && !(g_k1_forms_off.load(std::memory_order_relaxed) & 2) && D % 128 == 0 && DS / 4 >= 3 && K % 512 == 0 && K > 0 && M < ((int64_t)1 << 31) && (M + AW_ROWS - 1) / AW_ROWS <= nblk &&
           (int64_t)B * MT * DS * 128 < ((int64_t)1 << 31);
}
int assign_wide_launch(const void* xr, const void* wt, int B, int T, int MT, int D, int K, void* logits_bf16, float* stats, int nblk,
                       int timing_tag, hipStream_t stream, const char* what) {
    AssignFlatArgs a{};
    a.xr = (const uint4*)xr; a.wt = (const uint4*)wt;
    a.M = B * T; a.T = T; a.MT = MT; a.DS = D / 32; a.NT = K / 32; a.K = K;
    a.logits = (float*)logits_bf16; a.stats = stats; a.nblk = nblk;
    const int nwg = (a.M + AW_ROWS - 1) / AW_ROWS;
    if ((((uintptr_t)xr | (uintptr_t)wt) & 15) != 0) {
        set_error("%s: internal: unaligned tiles", what);
        return LPM_ERR_BADARG;
    }
    const dim3 grid((unsigned)nwg, (unsigned)(K / 512));
    const size_t lds = (size_t)AW_NS * AW_STAGE;
    void (*kern)(const AssignFlatArgs) = assign_wide_kernel<0>;
#ifdef LPM_K1_WIDE_EXPERIMENTS                                     // tools/k1_bf16_loop.py: build with LPM_EXTRA_HIPCC_FLAGS=-DLPM_K1_WIDE_EXPERIMENTS
    static const int dbg = [] { const char* e = getenv("LPM_K1_WIDE_DBG"); return e ? atoi(e) : 0; }();
    switch (dbg) {
        case 1: kern = assign_wide_kernel<1>; break;
        case 2: kern = assign_wide_kernel<2>; break;
        case 4: kern = assign_wide_kernel<4>; break;
        case 6: kern = assign_wide_kernel<6>; break;
        case 8: kern = assign_wide_kernel<8>; break;
        case 16: kern = assign_wide_kernel<16>; break;
        case 22: kern = assign_wide_kernel<22>; break;
        case 64: kern = assign_wide_kernel<64>; break;
        case 128: kern = assign_wide_kernel<128>; break;
        default: break;
    }
#endif
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %zu bytes of LDS", what, lds);
        return LPM_ERR_LAUNCH;
    }
    hipEvent_t e0, e1;
    if (timing_tag && timing_request(timing_tag, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(512), lds, stream, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
    return check_launch(what);
}

static int af_enabled() {
    static const int on = [] {
        const char* e = getenv("LPM_K1_FLAT");         // 0: the 128-row tile GEMM form (A/B switch)
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on && !(g_k1_forms_off.load(std::memory_order_relaxed) & 1);
}

// steps per stage (LPM_K1_KB = 1, 2, 4) and steps of B fragments in flight (LPM_K1_DB = 4; 8 with four steps per stage): A/B switches
static int af_kb() {
    static const int kb = [] { const char* e = getenv("LPM_K1_KB"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 2; }();
    return kb;
}
static int af_db() {
    static const int db = [] { const char* e = getenv("LPM_K1_DB"); const int v = e ? atoi(e) : 0; return (v == 8 && af_kb() == 4) ? 8 : 4; }();
    return db;
}

bool assign_flat_ok(int B, int T, int D, int K) {
    const int DS = D / 16, db = af_db(), tail = af_tail(af_kb(), db);
    return af_enabled() && D % 16 == 0 && DS % db == 0 && DS >= tail + db && K % 256 == 0 && K > 0 && (int64_t)B * T < (int64_t)1 << 31;
}
// plain bf16 tiles: double steps of 32 reduction elements (LPM_K1_FLAT_BF16=0: the 128-row tile GEMM form, A/B)
bool assign_flat_plain_ok(int B, int T, int D, int K) {
    static const int on = [] { const char* e = getenv("LPM_K1_FLAT_BF16"); return (e && e[0] == '0') ? 0 : 1; }();
    const int DS = D / 32, tail = af_tail(2, 4);
    return on && af_enabled() && D % 32 == 0 && DS % 4 == 0 && DS >= tail + 4 && K % 256 == 0 && K > 0 && (int64_t)B * T < (int64_t)1 << 31;
}

int assign_flat_launch(const void* xr, const void* wt, int B, int T, int MT, int D, int K, float* logits, float* stats, int nblk,
                       int timing_tag, hipStream_t stream, const char* what, int planes) {
    AssignFlatArgs a{};
    a.xr = (const uint4*)xr; a.wt = (const uint4*)wt;
    a.M = B * T; a.T = T; a.MT = MT; a.DS = D / (planes == 1 ? 32 : 16); a.NT = K / 32; a.K = K;
    a.logits = logits; a.stats = stats; a.nblk = nblk;
    const int nwg = (a.M + AF_ROWS - 1) / AF_ROWS;
    if (nwg > nblk || (((uintptr_t)xr | (uintptr_t)wt) & 15) != 0) {
        set_error("%s: internal: %d row groups for %d statistics rows, or unaligned tiles", what, nwg, nblk);
        return LPM_ERR_BADARG;
    }
    const dim3 grid((unsigned)nwg, (unsigned)(K / 256));
    const int kb = af_kb(), db = af_db();
    const size_t lds = (size_t)af_ns(kb) * kb * AF_KSTEP;
    hipEvent_t e0, e1;
    const bool timed = timing_tag && timing_request(timing_tag, &e0, &e1);
#define AF_LAUNCH(KB, DB, LUMP)                                                                             \
    do {                                                                                                    \
        if (timed) hipExtLaunchKernelGGL((assign_flat_kernel<KB, DB, LUMP>), grid, dim3(512), lds, stream, e0, e1, 0, a); \
        else hipLaunchKernelGGL((assign_flat_kernel<KB, DB, LUMP>), grid, dim3(512), lds, stream, a);      \
    } while (0)
    static const int lump = [] { const char* e = getenv("LPM_K1_LUMP"); return (e && e[0] == '1') ? 1 : 0; }();
    if (planes == 1) {
        const size_t lds1 = (size_t)af_ns(2) * 2 * AF_KSTEP;
        if (timed) hipExtLaunchKernelGGL((assign_flat_kernel<2, 4, false, true>), grid, dim3(512), lds1, stream, e0, e1, 0, a);
        else hipLaunchKernelGGL((assign_flat_kernel<2, 4, false, true>), grid, dim3(512), lds1, stream, a);
        return check_launch(what);
    }
    if (lump && kb == 2) AF_LAUNCH(2, 4, true);
    else if (kb == 1) AF_LAUNCH(1, 4, false);
    else if (kb == 2) AF_LAUNCH(2, 4, false);
    else if (db == 8) AF_LAUNCH(4, 8, false);
    else AF_LAUNCH(4, 4, false);
#undef AF_LAUNCH
    return check_launch(what);
}

}  // namespace lpm

// mask: bit 0 the flat 96-row K1 forms, bit 1 the 160 x 512 plain-bf16 form; returns the previous mask.  Process-wide, like
// lpm_kernel_timing_enable: set it from one thread before the launches it is meant for.
extern "C" int lpm_k1_forms_disable(int mask) { return lpm::g_k1_forms_off.exchange(mask & 3); }
