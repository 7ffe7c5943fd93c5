// a14 + a15 for a variable whose gradient is a product of two skinny matrices -- the hidden projection's weight
// (frame_level_models.py:2314-2319): dW [N1, N2] = X^T DY with X [R, N1] the pooled descriptors and DY [R, N2] the gradient of the
// projection's output, R = the batch (all towers' clips: utils.combine_gradients' SUM over towers, utils.py:192-213, is the same
// product over the concatenated rows).  At cfg-2 the variable is 138.4 M of the model's 185 M parameters (553.6 M of 591 M at
// cfg-5) and its gradient is 554 MB (2.2 GB) that the generic path writes (lpm_skinny_weight_grad_tiles), reads for the norm
// (ca_chunk_sumsq) and reads again for the update (ca_apply).  Here the gradient never exists in memory:
//   pass 1  tile GEMM X^T DY, epilogue = sum of squares of the tile -> one partial per workgroup        (reads the operands only)
//   pass 2  the partials in a fixed order -> factor = clip / max(||dW||, clip)                          (utils.clip_gradient_norms :170-189)
//   pass 3  tile GEMM again, epilogue = factor, then TF-Adam on the tile of param / m / v in place     (24 B per parameter)
// against 4 (write) + 4 (norm) + 28 (update) = 36 B per parameter before.  The arithmetic of a gradient element is the tile GEMM's
// (split-bf16 operands, three MFMAs per product, fp32 accumulation: what lpm_skinny_weight_grad_tiles computes, bit for bit), the
// update is clip_adam.hip's.  In data-parallel training the towers exchange X and DY tiles (86 MB per tower at cfg-2, all-gather)
// instead of all-reducing the 554 MB gradient.
#include <type_traits>
#include "tile_gemm.h"

namespace lpm {

// partial[n] -> factor: 1024 threads, eight loads per thread and round, fp64, fixed order
__global__ __launch_bounds__(1024) void fa_factor_kernel(const float* __restrict__ partial, int64_t n, float clip, float* __restrict__ factor) {
    double s = 0.0;
    for (int64_t c = threadIdx.x; c < n; c += 1024 * 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (c + 1024 * u < n) ? partial[c + 1024 * u] : 0.f;
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
        factor[0] = clip > 0.f ? clip / fmaxf(nrm, clip) : 1.f;
        factor[1] = nrm;                      // for the caller's diagnostics (the norm the clip saw)
    }
}

static int64_t fa_workgroups(int N1, int N2) { return (int64_t)((N1 + 63) / 64) * ((N2 + 127) / 128); }

// ||X^T DY||_F^2 without the product: column n1 of the gradient is X[:, n1]^T DY, so its square norm is the quadratic form
// x_n1^T G x_n1 with G = DY DY^T [R, R] (a host-side 128 x 128 matrix product), and the norm^2 is the sum of these NON-NEGATIVE forms
// over the N1 columns.  One pass: Y = G X by MFMA -- A = G as split-bf16 row tiles in LDS (<= 64 KB, loaded once per workgroup),
// B = the X weight tiles the update pass reads anyway -- then sum X .* Y with X in fp32 from the projection's input.  2 R^2 N1
// flops instead of the 2 R N1 N2 of forming the gradient (8 x fewer at cfg-5), 8 bytes per element of X.  R <= 128.
constexpr int FQ_GRID = 512;
// XT (round 3): the X of "sum X .* Y" comes out of the SAME tiles (hi + lo: 2^-17 relative, a norm does not see it) instead of a second
// read of the fp32 matrix -- the pass was 8 bytes per element of X at HBM rate.  A B fragment holds, per 16-row step, rows 8 half + e of
// column l31; the accumulator wants rows 4 half + i + 8 j: for j even (rows 0-7 of a step) the half-0 lanes' elements 4 half + i, for j odd
// the half-1 lanes' -- four of the eight values are the lane's own, four its partner's (lane ^ 32): one 4-value exchange per step.
template <bool XT>
__global__ __launch_bounds__(256, 2) void fa_quadform_kernel(const uint4* __restrict__ xt, const float* __restrict__ x, int64_t ldx,
                                                             const uint4* __restrict__ gdt, int R, int N1, int NT1, int MT,
                                                             float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tg_u32x4* sg = reinterpret_cast<tg_u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
    const int S = R / 16;
    for (int i = tid; i < MT * S * 128; i += 256) {
        const uint4 v = gdt[i];
        sg[i] = tg_u32x4{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
    double tot = 0.0;
    for (int tile = blockIdx.x * 4 + wave; tile < NT1; tile += gridDim.x * 4) {
        f32x16 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        float xo[XT ? 8 : 1][4], xp[XT ? 8 : 1][4];       // per step: this lane's rows 4 half + i of its own fragment half, and of its partner's
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s >= S) break;
            const uint4 h4 = xt[(((int64_t)s * NT1 + tile) * 2 + 0) * 64 + lane], l4 = xt[(((int64_t)s * NT1 + tile) * 2 + 1) * 64 + lane];
            const tg_u32x4 xh = {h4.x, h4.y, h4.z, h4.w}, xl = {l4.x, l4.y, l4.z, l4.w};
            if constexpr (XT) {
                const int hf = lane >> 5;
                float xf[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned sh = (e & 1) * 16;
                    xf[e] = __uint_as_float((xh[e >> 1] >> sh) << 16) + __uint_as_float((xl[e >> 1] >> sh) << 16);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xo[s][i] = hf ? xf[4 + i] : xf[i];                 // own element 4 half + i
                    const float send = hf ? xf[i] : xf[4 + i];         // the partner's 4 half' + i
                    xp[s][i] = __shfl_xor(send, 32, 64);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < MT) {
                    const tg_u32x4 gh = sg[((rt * S + s) * 2 + 0) * 64 + lane], gl = sg[((rt * S + s) * 2 + 1) * 64 + lane];
                    acc[rt] = tg_mfma(gh, xh, acc[rt]);
                    acc[rt] = tg_mfma(gh, xl, acc[rt]);
                    acc[rt] = tg_mfma(gl, xh, acc[rt]);
                }
            }
        }
        // acc[rt][r] = (G X)[b = 32 rt + mfma32_row(r, lane)][n1 = 32 tile + l31]
        const int n1 = tile * 32 + l31;
        float part = 0.f;
        if constexpr (XT) {
            const int hf = lane >> 5;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < MT) {
#pragma unroll
                    for (int js = 0; js < 2; ++js) {
                        const int st = 2 * rt + js;                    // the 16-row step the rows 16 js .. 16 js + 15 of this row tile came from
                        if (st < S) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float x0 = hf ? xp[st][i] : xo[st][i];      // rows 8 (2 js) + 4 half + i: the half-0 fragment
                                const float x1 = hf ? xo[st][i] : xp[st][i];      // rows 8 (2 js + 1) + 4 half + i: the half-1 fragment
                                part = fmaf(acc[rt][4 * (2 * js) + i], x0, part);
                                part = fmaf(acc[rt][4 * (2 * js + 1) + i], x1, part);
                            }
                        }
                    }
                }
            }
        } else
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (rt < MT) {
                float xv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int b = 32 * rt + mfma32_row(r, lane);
                    xv[r] = (b < R && n1 < N1) ? x[(int64_t)b * ldx + n1] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(acc[rt][r], xv[r], part);
            }
        }
        tot += (double)part;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// Update pass, second form (round 3): long row pieces, no LDS.  The tile-GEMM form above gives a workgroup a 64 x 128 tile of the
// [N1, N2] variable: 512-byte pieces of 64 rows 2 KB apart, in three arrays -- 5.4 TB/s, where 128- or 256-wide rows (each
// workgroup's tile contiguous) stream at 6.2.  Here a workgroup owns 32 rows x 256 columns (1 KB row pieces; the column pieces of the
// same rows are consecutive workgroups) and there is NO LDS and NO barrier: every wave first requests its param / m / v elements in the MFMA
// accumulator's own layout (for a fixed register the 32 lanes of a half-wave hold 32 consecutive columns of one row: 128-byte
// segments, non-temporal), then runs its small GEMM -- R / 16 steps, the A fragment (this row block's X tile) and its two B fragments (DY
// tiles, 0.2 MB in all: L2-resident) straight from L2 into registers as 16-byte-per-lane fragment loads, double-buffered -- UNDER the
// HBM round trip of those requests, then clip factor, TF-Adam (clip_adam.hip's arithmetic) and non-temporal stores in the same layout.
// The gradient element is bit for bit the tile GEMM's (same fragments, same three MFMAs per product in the same order).
// Workgroup = 256 threads = 32 rows x 256 columns (two workgroups per CU at ~180 registers: one's loads are in flight while the other
// computes and stores); consecutive workgroups = the N2 / 256 column pieces of the same rows.  Needs N1 %% 32 == 0, N2 %% 256 == 0.
__global__ __launch_bounds__(256, 2) void fa_update_rows_kernel(const uint4* __restrict__ xt, const uint4* __restrict__ dyt, int steps, int N1,
                                                                 int N2, int NT1, int NT2, int NCB, float* __restrict__ param,
                                                                 float* __restrict__ mom, float* __restrict__ var,
                                                                 const float* __restrict__ factor, float lr_t, float b1, float b2, float eps) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int rb = blockIdx.x / NCB, cb = blockIdx.x % NCB;                    // 32-row block = X tile index; 256-column piece
    const int64_t row0 = (int64_t)rb * 32;
    const int ct0 = cb * 8 + wave * 2;                                          // this wave's first 32-column tile
    f32x16 pa[2], ma[2], va[2];
    // (uniform base per register + one per-lane offset, as the stores below: SGPR arithmetic, one VGPR of addressing)
    const unsigned voff = (unsigned)(4 * (lane >> 5) * N2 + l31) * 4u;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t o = (row0 + (r & 3) + 8 * (r >> 2)) * N2 + (ct0 + c) * 32;         // row + 4 half and column + l31 in voff
            pa[c][r] = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(param + o) + voff));
            ma[c][r] = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(mom + o) + voff));
            va[c][r] = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(var + o) + voff));
        }
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const uint4* ap = xt + (int64_t)rb * 128 + lane;                            // + s * NT1 * 128; plane: + 64
    const uint4* bp = dyt + (int64_t)ct0 * 128 + lane;                          // + s * NT2 * 128; column tile c: + c * 128; plane: + 64
    struct Frag { uint4 ah, al, bh[2], bl[2]; };
    auto load = [&](int s, Frag& f) {
        const uint4* a = ap + (int64_t)s * NT1 * 128;
        const uint4* b = bp + (int64_t)s * NT2 * 128;
        f.ah = a[0]; f.al = a[64];
        f.bh[0] = b[0]; f.bl[0] = b[64]; f.bh[1] = b[128]; f.bl[1] = b[192];
    };
    auto mac = [&](const Frag& f) {
        const tg_u32x4 ah = {f.ah.x, f.ah.y, f.ah.z, f.ah.w}, al = {f.al.x, f.al.y, f.al.z, f.al.w};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const tg_u32x4 bh = {f.bh[c].x, f.bh[c].y, f.bh[c].z, f.bh[c].w}, bl = {f.bl[c].x, f.bl[c].y, f.bl[c].z, f.bl[c].w};
            acc[c] = tg_mfma(ah, bh, acc[c]);
            acc[c] = tg_mfma(ah, bl, acc[c]);
            acc[c] = tg_mfma(al, bh, acc[c]);
        }
    };
    Frag fa, fb;
    if (steps > 0) load(0, fa);
    for (int s = 0; s < steps; s += 2) {
        if (s + 1 < steps) load(s + 1, fb);
        mac(fa);
        if (s + 1 < steps) {
            if (s + 2 < steps) load(s + 2, fa);
            mac(fb);
        }
    }
    const float fac = *factor;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pn = pa[c][r], mn = ma[c][r], vn = va[c][r];
            adam_element(acc[c][r] * fac, pn, mn, vn, lr_t, b1, b2, eps);
            const int64_t o = (row0 + (r & 3) + 8 * (r >> 2)) * N2 + (ct0 + c) * 32;
            __builtin_nontemporal_store(pn, reinterpret_cast<float*>(reinterpret_cast<char*>(param + o) + voff));
            __builtin_nontemporal_store(mn, reinterpret_cast<float*>(reinterpret_cast<char*>(mom + o) + voff));
            __builtin_nontemporal_store(vn, reinterpret_cast<float*>(reinterpret_cast<char*>(var + o) + voff));
        }
}


// Update pass, third form (round 6; VERDICT r3-r5: "the projection's input gradient folded into the update pass"): the variable keeps a
// bf16 compute copy (BASELINE configs[4]), and the projection's input gradient dX = DY W_old^T reads exactly the weights this pass
// streams anyway -- 1.1 GB of copy that lpm_proj_dx_w16 read once more per step.  A workgroup owns 64 rows x ALL N2 columns and walks
// the row block's 128-column pieces, so that the partial dX [R, 64] of the pieces meet in its registers in a fixed order (the tile-GEMM
// form's eight column blocks of a row block are eight workgroups: their partial sums would have to meet in HBM).  Per piece:
//   G = X^T DY of the 64 x 128 tile: the row block's X tiles sit in LDS for the workgroup's whole life (HBM / L2 once, not once per
//       column block), wave w's DY fragments come straight from L2 into registers; three MFMAs per product in the tile GEMM's order:
//       the gradient element is the tile GEMM's bit for bit;
//   through LDS into the row-contiguous mapping of the param / m / v pieces requested one piece ahead (16-byte non-temporal accesses),
//       clip factor, TF-Adam (clip_adam.hip's arithmetic), the new weight's bf16 copy;
//   bf16(W_old) of the tile -- the values the forward multiplied by -- into LDS [64][128] (over the gradient's staging tile), and
//       dX[b-tile w][64 rows] += DY16[b-tile w][128 columns] . W16^T: A = 16-byte pieces of the bf16 DY rows from L2, B = 16-byte row
//       pieces of the LDS tile, one MFMA per product (lpm_proj_dx_w16's arithmetic: both operands rounded once to bf16).
// Needs N1 % 64 == 0, N2 % 128 == 0, R % 16 == 0, R <= 128.  256 threads, 4 KB x R / 16 + 33 KB of LDS, two workgroups per CU.
constexpr int FF_ES = 132;            // floats between the rows of the gradient's staging tile
constexpr int FF_WS = 136;            // bf16 elements between the rows of the old weight's tile (272 bytes: fragments stay 16-byte aligned)
template <bool DX>
__global__ __launch_bounds__(256, 2) void fa_update_fold_kernel(const uint4* __restrict__ xt, const uint4* __restrict__ dyt, int RS, int N1, int N2,
                                                                 int NT1, int NT2, float* __restrict__ param, float* __restrict__ mom,
                                                                 float* __restrict__ var, unsigned short* __restrict__ p16,
                                                                 const float* __restrict__ factor, float lr_t, float b1, float b2, float eps,
                                                                 const unsigned short* __restrict__ dy16, int R, float* __restrict__ dx, int64_t ldx,
                                                                 int dbg) {
    // dbg (tools/time_factored_fold.py, LPM_FA_FOLD_DBG; results are garbage): 4 no dX stores, 16 no gradient GEMM
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tg_u32x4* sA = reinterpret_cast<tg_u32x4*>(smem);                      // [row tile][step][plane][lane]
    float* es = reinterpret_cast<float*>(smem + (size_t)RS * 4096);        // [64][FF_ES] fp32; then, over it, [64][FF_WS] bf16
    unsigned short* ws = reinterpret_cast<unsigned short*>(es);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int rb = blockIdx.x;
    const int tcol = (tid & 31) * 4, trow = tid >> 5;                      // this thread's four columns of a piece; rows trow + 8 j
    const int NCB = N2 / 128;
    for (int p = wave; p < 4 * RS; p += 4) {
        const int m = p / (2 * RS), rem = p - m * 2 * RS;
        const uint4 t = xt[((int64_t)(rem >> 1) * NT1 + rb * 2 + m) * 128 + (rem & 1) * 64 + lane];
        sA[p * 64 + lane] = tg_u32x4{t.x, t.y, t.z, t.w};
    }
    f32x4 pa[8], ma[8], va[8];
    // (a uniform base per piece and row group + ONE per-lane offset, as fa_update_rows_kernel: scalar address arithmetic, one register)
    const unsigned voff = (unsigned)(trow * N2 + tcol) * 4u;
    auto load_pmv = [&](int cb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t o = ((int64_t)rb * 64 + 8 * j) * N2 + cb * 128;
            pa[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(param + o) + voff));
            ma[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(mom + o) + voff));
            va[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(var + o) + voff));
        }
    };
    load_pmv(0);
    f32x16 dxacc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxacc[n][r] = 0.f;
    const float fac = *factor;
    const bool dxw = DX && wave * 32 < R;                                  // this wave owns a tile of clips (wave-uniform)
    const int brow = min(wave * 32 + l31, R - 1);                          // (rows past the end: the last row again, masked at the store)
    __syncthreads();
    // One piece.  LAST (compile time): no further piece to request -- the loop body is straight-line code, so that the compiler's count of
    // outstanding loads in front of each wait is exact: waits placed at the join behind a conditional load are the conservative ones, and
    // three builds of this kernel that differed in nothing else ran 2.69 / 2.89 / 3.21 ms.  Waves without a tile of clips (R < 128)
    // run the dX part on the last row again and store nothing.
    auto piece = [&](const int cb, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        // ---- G = X^T DY of the piece: wave w = column tile cb * 4 + w, both row tiles --------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        const uint4* bp = dyt + (int64_t)(cb * 4 + wave) * 128 + lane;     // + s * NT2 * 128; plane: + 64
        struct Frag { uint4 bh, bl; };
        auto load = [&](int s, Frag& f) {
            const uint4* b = bp + (int64_t)s * NT2 * 128;
            f.bh = b[0];
            f.bl = b[64];
        };
        auto mac = [&](int s, const Frag& f) {
            const tg_u32x4 bh = {f.bh.x, f.bh.y, f.bh.z, f.bh.w}, bl = {f.bl.x, f.bl.y, f.bl.z, f.bl.w};
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const tg_u32x4 ah = sA[((m * RS + s) * 2 + 0) * 64 + lane], al = sA[((m * RS + s) * 2 + 1) * 64 + lane];
                acc[m] = tg_mfma(ah, bh, acc[m]);
                acc[m] = tg_mfma(ah, bl, acc[m]);
                acc[m] = tg_mfma(al, bh, acc[m]);
            }
        };
        Frag fa, fb;
        const int RSd = (dbg & 16) ? 0 : RS;
        if (RSd > 0) load(0, fa);
        for (int s = 0; s < RSd; s += 2) {
            if (s + 1 < RSd) load(s + 1, fb);
            mac(s, fa);
            if (s + 1 < RSd) {
                if (s + 2 < RSd) load(s + 2, fa);
                mac(s + 1, fb);
            }
        }
        // ---- through LDS into the row-contiguous mapping; clip factor, TF-Adam, stores ----------------------------------------------
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) es[(m * 32 + mfma32_row(r, lane)) * FF_ES + wave * 32 + l31] = acc[m][r];
        __syncthreads();
        uint2 w16[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 gg = *reinterpret_cast<const float4*>(es + (trow + 8 * j) * FF_ES + tcol);
            const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
            const int64_t o = ((int64_t)rb * 64 + 8 * j) * N2 + cb * 128;
            f32x4 pn, mn, vn;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float pe = pa[j][q], me = ma[j][q], ve = va[j][q];
                adam_element(gv[q] * fac, pe, me, ve, lr_t, b1, b2, eps);
                pn[q] = pe; mn[q] = me; vn[q] = ve;
            }
            __builtin_nontemporal_store(pn, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(param + o) + voff));
            __builtin_nontemporal_store(mn, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(mom + o) + voff));
            __builtin_nontemporal_store(vn, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(var + o) + voff));
            const of_f2 p01 = {pn[0], pn[1]}, p23 = {pn[2], pn[3]};
            uint2 w;
            w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(p01, of_b2));
            w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(p23, of_b2));
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p16 + o) + (voff >> 1)) = w;
            if (DX) {                                   // the weight the forward multiplied by: bf16 of the OLD master
                const of_f2 o01 = {pa[j][0], pa[j][1]}, o23 = {pa[j][2], pa[j][3]};
                w16[j].x = __builtin_bit_cast(unsigned, __builtin_convertvector(o01, of_b2));
                w16[j].y = __builtin_bit_cast(unsigned, __builtin_convertvector(o23, of_b2));
            }
        }
        if (!DX) {
            if (!LAST) load_pmv(cb + 1);
            __syncthreads();                            // the staging tile is free for the next piece
            return;
        }
        // ---- dX += DY16 . W16^T over the piece's 128 columns ------------------------------------------------------------------------
        tg_u32x4 da[8];                                 // requested BEFORE the next piece's param / m / v: loads return in order
        {
            const unsigned short* dyp = dy16 + (int64_t)brow * N2 + cb * 128 + half * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) da[ks] = *reinterpret_cast<const tg_u32x4*>(dyp + ks * 16);
        }
        if (!LAST) load_pmv(cb + 1);
        __syncthreads();                                // everybody has read its gradient values: the tile takes the old weights
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<uint2*>(ws + (trow + 8 * j) * FF_WS + tcol) = w16[j];
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const tg_u32x4 wb = *reinterpret_cast<const tg_u32x4*>(ws + (n * 32 + l31) * FF_WS + ks * 16 + half * 8);
                dxacc[n] = tg_mfma(da[ks], wb, dxacc[n]);
            }
        __syncthreads();                                // the fragment reads are done: the next piece's gradient may land
    };
    for (int cb = 0; cb + 1 < NCB; ++cb) piece(cb, std::false_type{});
    piece(NCB - 1, std::true_type{});
    if (DX && !(dbg & 4)) {
        // dX [R, 64] of the row block: through the staging tile (free behind the last piece's barrier) as [clip][64 rows], so that it leaves
        // as 16-byte stores, sixteen lanes to a clip's 256 bytes (the accumulators hold one row per lane: 4-byte stores, 77 us of them)
        float* dt = es;
        if (dxw) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) dt[(wave * 32 + mfma32_row(r, lane)) * 64 + n * 32 + l31] = dxacc[n][r];
        }
        __syncthreads();
        const int c4 = (tid & 15) * 4;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int b = ps * 16 + (tid >> 4);
            if (b < R) *reinterpret_cast<float4*>(dx + (int64_t)b * ldx + (int64_t)rb * 64 + c4) = *reinterpret_cast<const float4*>(dt + b * 64 + c4);
        }
    }
}

}  // namespace lpm

extern "C" size_t lpm_factored_clip_adam_scratch_bytes(int N1, int N2) {
    return (size_t)(lpm::fa_workgroups(N1, N2) + 4) * sizeof(float);
}

static int factored_clip_adam_impl(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                   float* param, float* m, float* v, float clip_norm, float lr, float beta1, float beta2, float eps,
                                   int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream, void* param_bf16 = nullptr,
                                   const void* dy_bf16 = nullptr, float* dx = nullptr, int64_t ld_dx = 0);
// ... keeping a bf16 compute copy of the weight beside its fp32 master (SURVEY section 7 hard part 2; BASELINE configs[4]): param_bf16
// [N1, N2] receives bf16(param) from the update pass's epilogue -- the copy the projection's forward and input-gradient passes read
// (lpm_proj_fwd_parts_w16, lpm_proj_dx_w16) instead of streaming the fp32 weight twice more.  x / gdt NULL: the norm from a GEMM pass
// (lpm_factored_clip_adam), else from the quadratic forms (lpm_factored_clip_adam_q).  The tile-GEMM form of the update pass only.
extern "C" int lpm_factored_clip_adam_copy(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                           float* param, float* m, float* v, void* param_bf16, float clip_norm, float lr, float beta1,
                                           float beta2, float eps, int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(param_bf16 && ((uintptr_t)param_bf16 & 7) == 0, LPM_ERR_BADARG, "lpm_factored_clip_adam_copy: the copy must be 8-byte aligned");
    LPM_REQUIRE((x == nullptr) == (gdt == nullptr) && (!x || (ldx >= N1 && R <= 128)), LPM_ERR_BADARG,
                "lpm_factored_clip_adam_copy: x and the tiles of DY DY^T come together (row stride >= N1, R <= 128)");
    return factored_clip_adam_impl(xt, dyt, x, ldx, gdt, R, N1, N2, param, m, v, clip_norm, lr, beta1, beta2, eps, step, scratch, scratch_bytes,
                                   stream, param_bf16);
}
// lpm_factored_clip_adam_copy that also returns the projection's INPUT gradient dx [R, N1] (row stride ld_dx) = DY W_old^T, formed from the
// weights the update pass streams anyway (fa_update_fold_kernel): dy_bf16 = DY [R, N2] rounded once to bf16 (row-major), W_old by its
// bf16 rounding -- the compute copy's values, lpm_proj_dx_w16's arithmetic.  lpm_factored_fold_supported says where it applies.
extern "C" int lpm_factored_fold_supported(int R, int N1, int N2) {
    return (R > 0 && R % 16 == 0 && R <= 128 && N1 > 0 && N1 % 64 == 0 && N2 > 0 && N2 % 128 == 0) ? 1 : 0;
}
extern "C" int lpm_factored_clip_adam_copy_dx(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                              float* param, float* m, float* v, void* param_bf16, const void* dy_bf16, float* dx, int64_t ld_dx,
                                              float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step, float* scratch,
                                              size_t scratch_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(param_bf16 && ((uintptr_t)param_bf16 & 7) == 0, LPM_ERR_BADARG, "lpm_factored_clip_adam_copy_dx: the copy must be 8-byte aligned");
    LPM_REQUIRE((x == nullptr) == (gdt == nullptr) && (!x || (ldx >= N1 && R <= 128)), LPM_ERR_BADARG,
                "lpm_factored_clip_adam_copy_dx: x and the tiles of DY DY^T come together (row stride >= N1, R <= 128)");
    LPM_REQUIRE(dy_bf16 && dx && ld_dx >= N1 && ld_dx % 4 == 0 && (((uintptr_t)dy_bf16 | (uintptr_t)dx) & 15) == 0, LPM_ERR_BADARG,
                "lpm_factored_clip_adam_copy_dx: needs the bf16 DY and dx 16-byte aligned, dx with a row stride >= N1 and a multiple of 4");
    return factored_clip_adam_impl(xt, dyt, x, ldx, gdt, R, N1, N2, param, m, v, clip_norm, lr, beta1, beta2, eps, step, scratch, scratch_bytes,
                                   stream, param_bf16, dy_bf16, dx, ld_dx);
}
extern "C" int lpm_factored_clip_adam(const void* xt, const void* dyt, int R, int N1, int N2, float* param, float* m, float* v,
                                      float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step, float* scratch,
                                      size_t scratch_bytes, lpm_stream_t stream) {
    return factored_clip_adam_impl(xt, dyt, nullptr, 0, nullptr, R, N1, N2, param, m, v, clip_norm, lr, beta1, beta2, eps, step, scratch,
                                   scratch_bytes, stream);
}
// ... with the norm from the quadratic forms (fa_quadform_kernel) instead of a first tile-GEMM pass: x = the fp32 matrix X [R, N1]
// (row stride ldx) the tiles xt were split from, gdt = lpm_split_rows_tiles(G, R, 1, R, R) of G = DY DY^T [R, R] (fp32).  R <= 128.
extern "C" int lpm_factored_clip_adam_q(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                        float* param, float* m, float* v, float clip_norm, float lr, float beta1, float beta2, float eps,
                                        int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && gdt && ldx >= N1 && R <= 128, LPM_ERR_BADARG, "lpm_factored_clip_adam_q: needs x (row stride >= N1), the tiles of DY DY^T and R <= 128");
    return factored_clip_adam_impl(xt, dyt, x, ldx, gdt, R, N1, N2, param, m, v, clip_norm, lr, beta1, beta2, eps, step, scratch, scratch_bytes, stream);
}
static int factored_clip_adam_impl(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                   float* param, float* m, float* v, float clip_norm, float lr, float beta1, float beta2, float eps,
                                   int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream, void* param_bf16,
                                   const void* dy_bf16, float* dx, int64_t ld_dx) {
    using namespace lpm;
    LPM_REQUIRE(xt && dyt && param && m && v && scratch, LPM_ERR_BADARG, "lpm_factored_clip_adam: null pointer");
    LPM_REQUIRE(R > 0 && R % 16 == 0 && N1 > 0 && N2 > 0 && N2 % 32 == 0 && step >= 1, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_factored_clip_adam: need R %% 16 == 0, N2 %% 32 == 0 and a 1-based step (R=%d N2=%d)", R, N2);
    LPM_REQUIRE((((uintptr_t)param | (uintptr_t)m | (uintptr_t)v) & 15) == 0, LPM_ERR_BADARG,
                "lpm_factored_clip_adam: param / m / v must be 16-byte aligned");
    LPM_REQUIRE(scratch_bytes >= lpm_factored_clip_adam_scratch_bytes(N1, N2), LPM_ERR_WORKSPACE, "lpm_factored_clip_adam: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int NT1 = (N1 + 31) / 32, NT2 = N2 / 32;
    const int64_t nwg = fa_workgroups(N1, N2);
    float* partial = scratch;
    float* factor = scratch + nwg;
    TileGemmArgs g{};
    g.a = (const uint4*)xt; g.a_tile = 128; g.a_step = (int64_t)NT1 * 128; g.a_batch = 0; g.a_tiles = NT1;
    g.b = (const uint4*)dyt; g.b_tile = 128; g.b_step = (int64_t)NT2 * 128; g.b_batch = 0; g.b_tiles = NT2;
    g.rb_per_batch = (N1 + 63) / 64; g.steps_per_split = R / 16; g.total_steps = R / 16;
    g.out = param; g.ldo = N2; g.rows_valid = N1; g.cols_valid = N2;
    g.cols_inner = 1;                     // the column blocks of a row block as neighbours: X tiles from HBM once
    // measurement switches (tools/time_factored.py): 1 = no reduction steps in either pass (the streams alone), 2 = no norm pass
    static const int dbg = [] { const char* e = getenv("LPM_FA_DBG"); return e ? atoi(e) : 0; }();
    if (dbg & 1) g.steps_per_split = g.total_steps = 0;
    int rc = LPM_OK;
    int64_t npart = nwg;
    if (gdt) {
        const int MT = 2 * ((R + 63) / 64);
        const size_t lds = (size_t)MT * (R / 16) * 2048;
        static const int from_tiles = [] { const char* e = getenv("LPM_FQ_TILES"); return (e && e[0] == '0') ? 0 : 1; }();   // 0: X from the fp32 matrix (A/B)
        auto kern = (from_tiles && R <= 128) ? fa_quadform_kernel<true> : fa_quadform_kernel<false>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("lpm_factored_clip_adam_q: cannot reserve %zu bytes of LDS", lds);
            return LPM_ERR_LAUNCH;
        }
        npart = FQ_GRID < nwg ? FQ_GRID : nwg;
        hipLaunchKernelGGL(kern, dim3((unsigned)npart), dim3(256), lds, s, (const uint4*)xt, x, ldx, (const uint4*)gdt, R, N1, NT1, MT, partial);
    } else {
        g.sumsq = partial;
        rc = (dbg & 2) ? LPM_OK : tile_gemm_store(g, 1, 1, s, "lpm_factored_clip_adam (norm pass)", 1);
        if (rc != LPM_OK) return rc;
    }
    hipLaunchKernelGGL(fa_factor_kernel, dim3(1), dim3(1024), 0, s, (const float*)partial, npart, clip_norm, factor);
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    g.sumsq = nullptr;
    // whole-row form of the update pass (fa_update_rows_kernel): N2 a multiple of 64, at most 1024; LPM_FA_ROWS=0: the tile-GEMM form (A/B)
    static const int rows_form = [] { const char* e = getenv("LPM_FA_ROWS"); return (e && e[0] == '0') ? 0 : 1; }();
    // Measured (tools/time_factored.py, whole update incl. the ~75 us norm pass): cfg-2 one tower (R = 80, N2 = 512) 693 us against 728;
    // cfg-5 (R = 128, N2 = 1024) 2741 against 2418 and eight towers of cfg-2 (R = 640) 1855 against 1749 -- every wave fetching its own
    // fragments from L2 costs 80 + 160 KB of L2 -> CU traffic per 192 KB of HBM stream at R = 80, but 256 + 512 KB per 384 KB at R = 128:
    // the whole-row form only where the reduction is short
    if (rows_form && !param_bf16 && N2 % 256 == 0 && N1 % 32 == 0 && (int64_t)g.total_steps * N2 <= 5 * 512 && (int64_t)NT1 * (N2 / 256) < ((int64_t)1 << 31)) {
        hipLaunchKernelGGL(fa_update_rows_kernel, dim3((unsigned)(NT1 * (N2 / 256))), dim3(256), 0, s, (const uint4*)xt, (const uint4*)dyt,
                           g.total_steps, N1, N2, NT1, NT2, N2 / 256, param, m, v, (const float*)factor, (float)lr_t, beta1, beta2, eps);
        return check_launch("lpm_factored_clip_adam (update pass, rows)");
    }
    // the row-block form (fa_update_fold_kernel): always when the projection's input gradient rides along; without it only on request
    // (LPM_FA_FOLD=2: A/B of the two streaming patterns; 0: never -- then a dx request is an error)
    static const int fold = [] { const char* e = getenv("LPM_FA_FOLD"); return e ? atoi(e) : 1; }();
    const bool fold_ok = param_bf16 && lpm_factored_fold_supported(R, N1, N2) && fold != 0;
    if (dx && !fold_ok) {
        set_error("lpm_factored_clip_adam_copy_dx: needs the compute copy, N1 %% 64 == 0, N2 %% 128 == 0, R %% 16 == 0 and R <= 128 (R=%d N1=%d N2=%d)", R, N1, N2);
        return LPM_ERR_UNSUPPORTED_SHAPE;
    }
    if (fold_ok && (dx || fold == 2)) {
        const int RS = R / 16;
        static const int fdbg = [] { const char* e = getenv("LPM_FA_FOLD_DBG"); return e ? atoi(e) : 0; }();
        const size_t lds = (size_t)RS * 4096 + (size_t)64 * FF_ES * sizeof(float);
        auto kern = dx ? fa_update_fold_kernel<true> : fa_update_fold_kernel<false>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("lpm_factored_clip_adam (update pass, row blocks): cannot reserve %zu bytes of LDS", lds);
            return LPM_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(N1 / 64)), dim3(256), lds, s, (const uint4*)xt, (const uint4*)dyt, RS, N1, N2, NT1, NT2, param, m, v,
                           (unsigned short*)param_bf16, (const float*)factor, (float)lr_t, beta1, beta2, eps, (const unsigned short*)dy_bf16, R, dx, ld_dx, fdbg);
        return check_launch("lpm_factored_clip_adam (update pass, row blocks)");
    }
    g.adam_p = param; g.adam_m = m; g.adam_v = v; g.adam_factor = factor; g.adam_p16 = (unsigned short*)param_bf16;
    g.adam_lr_t = (float)lr_t; g.adam_b1 = beta1; g.adam_b2 = beta2; g.adam_eps = eps;
    rc = tile_gemm_adam(g, s, "lpm_factored_clip_adam (update pass)");
    if (rc != LPM_OK) return rc;
    return check_launch("lpm_factored_clip_adam");
}
