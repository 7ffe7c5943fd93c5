// K2 on the bf16 matrix pipe without leaving the fp32 parity bar: split-bf16 ("bf16x3").
//
// Every fp32 operand v is carried as two bf16 planes, hi = bf16(v), lo = bf16(v - hi) (same 4 bytes per
// element as fp32, relative residual 2^-17), and a product a*x is accumulated in fp32 as
// ah*xh + ah*xl + al*xh  (three v_mfma_f32_32x32x16_bf16; the dropped al*xl term is 2^-16 of 2^-8).
// 3 x 32 cycles per 32x32x16 tile instead of 8 x 64 cycles of v_mfma_f32_32x32x2_f32: 5.3x less matrix-pipe
// time for the same contraction, which moves K2 from fp32-MFMA-bound (80 us at cfg-2) towards its HBM
// roofline, at ~1e-5 relative error instead of ~1e-7.
//
// Both operands are stored in HBM in MFMA-fragment ("tile") order by light producer kernels, so that the
// aggregation kernel's operand fetch is ONE 16-byte global load per lane per fragment, lane-linear and fully
// coalesced (1 KB per wave instruction), with no LDS staging, no transposes and no barriers in the main loop:
//
//   XT[b][s][dt][p][lane][8]   = x[b, t = 16 s + 8 (lane>>5) + e, d = 32 dt + (lane&31)]   (B operand)
//   AT[b][kt][s][p][lane][8]   = a[b, t = 16 s + 8 (lane>>5) + e, k = 32 kt + (lane&31)]   (A operand)
//   p = 0: hi plane, p = 1: lo plane; frames >= T and clusters >= K are zero.
//
//   lpm_split_frames        x (fp32 row-major, row stride ldx)      -> XT
//   lpm_assign_tiles        logits -> [affine -> softmax] -> AT     (or similarities -> AT, softmax off)
//   lpm_vlad_aggregate_tiles_fwd   AT, XT, centres -> nrm, asum, colsq, csq   (same outputs as the fp32 K2)
#include "lpm_common.h"

namespace lpm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne(float v) {   // round-to-nearest-even, finite inputs
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_to_f32(unsigned h) { return __uint_as_float(h << 16); }

// pack 8 fp32 into hi / lo planes (4 dwords each)
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
    unsigned h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = bf16_rne(v[e]);
        l[e] = bf16_rne(v[e] - bf16_to_f32(h[e]));
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- x -> XT ------------------------------------------------------------------------------------
// work item = (clip b, step s, half kh, 4 consecutive d): 8 float4 row loads (coalesced across items),
// 4 x (hi, lo) 16-byte tile stores.
__global__ __launch_bounds__(256) void split_frames_kernel(const float* __restrict__ x, int64_t ldx, int B, int T,
                                                           int D, int S, uint4* __restrict__ xt, int planes) {
    const int D4 = D / 4, DT = D / 32;
    const int64_t total = (int64_t)B * S * 2 * D4;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int d4 = (int)(w % D4);
        const int64_t r = w / D4;
        const int kh = (int)(r & 1);
        const int s = (int)((r >> 1) % S), b = (int)((r >> 1) / S);
        float v[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = 16 * s + 8 * kh + e;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) f = *reinterpret_cast<const float4*>(x + ((int64_t)b * T + t) * ldx + 4 * d4);
            v[0][e] = f.x; v[1][e] = f.y; v[2][e] = f.z; v[3][e] = f.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = 4 * d4 + c, dt = d >> 5, j = d & 31;
            uint4 hi, lo;
            split8(v[c], hi, lo);
            if (planes == 1) {                 // plain bf16 tiles: one plane (S counts the padded steps of lpm_frame_steps_bf16)
                xt[(((int64_t)b * S + s) * DT + dt) * 64 + kh * 32 + j] = hi;
                continue;
            }
            const int64_t base = ((((int64_t)b * S + s) * DT + dt) * 2) * 64 + kh * 32 + j;
            xt[base] = hi;
            xt[base + 64] = lo;
        }
    }
}

// ---- logits / similarities -> AT -----------------------------------------------------------------
// one workgroup per (clip, step of 16 frames); K <= 1024.
// BF16IN (bf16 storage): `assign` holds bf16 values and AT is written as ONE plane; S then counts the padded steps
template <bool SOFTMAX, bool BF16IN>
__global__ __launch_bounds__(256) void assign_tiles_kernel(const float* __restrict__ assign,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int T, int K, int S,
                                                           int KT, uint4* __restrict__ at) {
    extern __shared__ float as[];           // [16][KP + 1], KP = KT * 32
    const int KP = KT * 32, KS = KP + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / S, s = blockIdx.x % S;
    for (int i = tid; i < 16 * KS; i += 256) as[i] = 0.f;
    __syncthreads();
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr, t = 16 * s + row;
        if (t >= T) continue;                // wave-uniform
        const float* ar = assign + ((int64_t)b * T + t) * K;
        const unsigned short* arb = reinterpret_cast<const unsigned short*>(assign) + ((int64_t)b * T + t) * K;
        auto ld = [&](int c) { return BF16IN ? bf16_to_f32(arb[c]) : ar[c]; };
        if (SOFTMAX) {
            float v[16];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = lane + 64 * j;
                v[j] = (c < K) ? fmaf(ld(c), scale ? scale[c] : 1.f, shift ? shift[c] : 0.f) : -INFINITY;
                m = fmaxf(m, v[j]);
            }
            m = wave_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v[j] = __expf(v[j] - m);
                sum += v[j];
            }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = lane + 64 * j;
                if (c < K) as[row * KS + c] = v[j] * inv;
            }
        } else {
            for (int c = lane; c < K; c += 64) as[row * KS + c] = ld(c);
        }
    }
    __syncthreads();
    for (int slot = tid; slot < KT * 64; slot += 256) {
        const int kt = slot >> 6, ln = slot & 63;
        const int kh = ln >> 5, i = ln & 31;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = as[(8 * kh + e) * KS + kt * 32 + i];
        uint4 hi, lo;
        split8(v, hi, lo);
        if (BF16IN) {
            at[(((int64_t)b * KT + kt) * S + s) * 64 + ln] = hi;
            continue;
        }
        const int64_t base = ((((int64_t)b * KT + kt) * S + s) * 2) * 64 + ln;
        at[base] = hi;
        at[base + 64] = lo;
    }
}

// Second form of the assignment-tile kernel for K = 64 VPL, VPL in {1, 2, 4, 8} (every NetVLAD stream of the BASELINE configurations):
// a lane owns VPL CONSECUTIVE clusters, so a row is one or two 16-byte loads per lane, and the four rows of a wave are loaded,
// reduced and exponentiated together -- 4 x the loads in flight and a quarter of the dependent wave-reduction chains of the
// row-at-a-time form above (cfg-2 video 17.8 -> see DESIGN; cfg-5 video, K = 512: 52.7 us before).  LDS tile [16][K + 4] (rows
// 16-byte aligned for the lane's vector store; the column-wise reads of the tile emission hit bank 4 r + i: conflict-free).
// NSTEP = 2 (round 6; bf16 logits at K = 512, BASELINE configs[4]'s video stream): a workgroup takes TWO consecutive frame steps, the second
// step's rows requested together with the first's -- 2 560 one-step workgroups over 8 resident per CU were 1.25 rounds of one dependent
// chain each (load, two butterflies, LDS round trip, store); 1 280 two-step workgroups are one round and the second chain's loads ride
// under the first chain's arithmetic.  Same reductions in the same order: bit-identical tiles.
template <bool SOFTMAX, bool BF16IN, int VPL, int NSTEP = 1>
__global__ __launch_bounds__(256) void assign_tiles2_kernel(const float* __restrict__ assign, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int T, int S, uint4* __restrict__ at) {
    constexpr int K = 64 * VPL, KS = K + 4, KT = K / 32;
    // BF16IN (round 4): the tile holds what leaves -- bf16, rounded where the row is stored -- at half the LDS (16.6 instead of 33 KB at
    // K = 512: eight workgroups per CU instead of four; the kernel is one dependent chain of load, two butterflies, LDS round trip and
    // store per workgroup, so its rate is its occupancy)
    constexpr int KS2 = K + 8;                               // bf16 row stride (rows 16-byte aligned)
    __shared__ __attribute__((aligned(16))) unsigned char as_raw[BF16IN ? 16 * KS2 * 2 : 16 * KS * 4];
    float* as = reinterpret_cast<float*>(as_raw);
    unsigned short* as16 = reinterpret_cast<unsigned short*>(as_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SB = S / NSTEP;                                // (S a multiple of NSTEP: the launcher's condition)
    const int b = blockIdx.x / SB, s0 = (blockIdx.x % SB) * NSTEP;
    const int c0 = lane * VPL;
    float sc[VPL], sh[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        sc[j] = (SOFTMAX && scale) ? scale[c0 + j] : 1.f;
        sh[j] = (SOFTMAX && shift) ? shift[c0 + j] : 0.f;
    }
    float v[4][VPL];
    auto load_rows = [&](int s) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int t = 16 * s + wave * 4 + rr;
        if (t < T) {                                         // wave-uniform
            if (BF16IN) {
                const unsigned short* p = reinterpret_cast<const unsigned short*>(assign) + ((int64_t)b * T + t) * K + c0;
                if (VPL == 8) {
                    const uint4 q = *reinterpret_cast<const uint4*>(p);
                    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[rr][2 * j] = bf16_to_f32(w[j] & 0xffffu);
                        v[rr][2 * j + 1] = bf16_to_f32(w[j] >> 16);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) v[rr][j] = bf16_to_f32(p[j]);
                }
            } else {
                const float* p = assign + ((int64_t)b * T + t) * K + c0;
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
            }
        } else {
#pragma unroll
            for (int j = 0; j < VPL; ++j) v[rr][j] = 0.f;
        }
    }
    };
    load_rows(s0);
#pragma unroll
    for (int ss = 0; ss < NSTEP; ++ss) {
    const int s = s0 + ss;
    if (SOFTMAX) {
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
        for (int rr = 0; rr < 4; ++rr) m[rr] = wave_max_dpp(m[rr]);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            sum[rr] = 0.f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                v[rr][j] = __expf(v[rr][j] - m[rr]);
                sum[rr] += v[rr][j];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) sum[rr] = wave_sum_dpp(sum[rr]);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const bool live = 16 * s + wave * 4 + rr < T;
            const float inv = live ? 1.f / sum[rr] : 0.f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) v[rr][j] *= inv;
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        if (BF16IN) {
            unsigned short* dst = as16 + (wave * 4 + rr) * KS2 + c0;
            if (VPL % 2 == 0) {                             // packed: one store of 2 VPL bytes per lane (4 / 8 / 16)
                unsigned w2[VPL / 2 > 0 ? VPL / 2 : 1];
#pragma unroll
                for (int j = 0; j < VPL / 2; ++j) w2[j] = bf16_rne(v[rr][2 * j]) | (bf16_rne(v[rr][2 * j + 1]) << 16);
                if (VPL == 8) *reinterpret_cast<uint4*>(dst) = make_uint4(w2[0], w2[1 % (VPL / 2 > 0 ? VPL / 2 : 1)], w2[2 % (VPL / 2 > 0 ? VPL / 2 : 1)], w2[3 % (VPL / 2 > 0 ? VPL / 2 : 1)]);
                else if (VPL == 4) *reinterpret_cast<uint2*>(dst) = make_uint2(w2[0], w2[1 % (VPL / 2 > 0 ? VPL / 2 : 1)]);
                else *reinterpret_cast<unsigned*>(dst) = w2[0];
            } else {
#pragma unroll
                for (int j = 0; j < VPL; ++j) dst[j] = (unsigned short)bf16_rne(v[rr][j]);
            }
            continue;
        }
        float* dst = as + (wave * 4 + rr) * KS + c0;
        if (VPL % 4 == 0) {
#pragma unroll
            for (int j = 0; j < VPL; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(v[rr][j], v[rr][j + 1], v[rr][j + 2], v[rr][j + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < VPL; ++j) dst[j] = v[rr][j];
        }
    }
    if (ss + 1 < NSTEP) load_rows(s + 1);                  // (the rows of this step are in LDS: their registers take the next step's, in flight under the emission below)
    __syncthreads();
    for (int slot = tid; slot < KT * 64; slot += 256) {
        const int kt = slot >> 6, ln = slot & 63;
        const int kh = ln >> 5, i = ln & 31;
        if (BF16IN) {
            unsigned h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = as16[(8 * kh + e) * KS2 + kt * 32 + i];
            at[(((int64_t)b * KT + kt) * S + s) * 64 + ln] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            continue;
        }
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = as[(8 * kh + e) * KS + kt * 32 + i];
        uint4 hi, lo;
        split8(w, hi, lo);
        const int64_t base = ((((int64_t)b * KT + kt) * S + s) * 2) * 64 + ln;
        at[base] = hi;
        at[base + 64] = lo;
    }
    if (ss + 1 < NSTEP) __syncthreads();                    // the tile is read: the next step may overwrite it
    }
}

// (Round 6, measured and removed: a third form for bf16 logits at K = 512 without LDS -- a WAVE owns the 8 frames of one fragment half x
// all 512 clusters, a lane 8 consecutive clusters, and each lane then holds eight consecutive 16-byte words of its tile: 128 contiguous
// bytes per lane, 5 120 independent waves, no barrier.  39.2 us against this form's 30.9 (tools/time_k2_bf16.py): one store instruction
// of it writes 64 sixteen-byte pieces 128 bytes apart -- eight partial-line writes per line where the LDS round trip below buys whole
// 1 KB tile pieces per instruction.)
template <bool BF16IN>
static bool launch_assign_tiles2(const void* assign, const float* scale, const float* shift, int B, int T, int K, int S, int softmax,
                                 void* at, hipStream_t stream, int timing_tag) {
    if (K != 64 && K != 128 && K != 256 && K != 512) return false;
    if ((((uintptr_t)assign | (uintptr_t)at) & 15) != 0) return false;
    static const int on = [] { const char* e = getenv("LPM_ASSIGN_TILES2"); return (e && e[0] == '0') ? 0 : 1; }();     // 0: first form (A/B)
    if (!on) return false;
    hipEvent_t e0, e1;
    const bool timed = timing_tag && timing_request(timing_tag, &e0, &e1);
    static const int two = [] { const char* e = getenv("LPM_ASSIGN_TILES_NSTEP"); return (e && e[0] == '1') ? 0 : 1; }();     // "1": one step per workgroup (A/B)
    if (BF16IN && K == 512 && S % 2 == 0 && two) {
        auto kern = softmax ? assign_tiles2_kernel<true, BF16IN, 8, 2> : assign_tiles2_kernel<false, BF16IN, 8, 2>;
        if (timed)
            hipExtLaunchKernelGGL(kern, dim3(B * S / 2), dim3(256), 0, stream, e0, e1, 0, (const float*)assign, scale, shift, T, S, (uint4*)at);
        else
            hipLaunchKernelGGL(kern, dim3(B * S / 2), dim3(256), 0, stream, (const float*)assign, scale, shift, T, S, (uint4*)at);
        return true;
    }
#define LPM_AT2(SM, VPL)                                                                                                            \
    do {                                                                                                                            \
        if (timed)                                                                                                                  \
            hipExtLaunchKernelGGL((assign_tiles2_kernel<SM, BF16IN, VPL>), dim3(B * S), dim3(256), 0, stream, e0, e1, 0, (const float*)assign, \
                                  scale, shift, T, S, (uint4*)at);                                                                  \
        else                                                                                                                        \
            hipLaunchKernelGGL((assign_tiles2_kernel<SM, BF16IN, VPL>), dim3(B * S), dim3(256), 0, stream, (const float*)assign, scale, shift, \
                               T, S, (uint4*)at);                                                                                   \
    } while (0)
#define LPM_AT2_K(SM)                                                                                                               \
    do {                                                                                                                            \
        if (K == 64) LPM_AT2(SM, 1); else if (K == 128) LPM_AT2(SM, 2); else if (K == 256) LPM_AT2(SM, 4); else LPM_AT2(SM, 8);      \
    } while (0)
    if (softmax) LPM_AT2_K(true); else LPM_AT2_K(false);
#undef LPM_AT2_K
#undef LPM_AT2
    return true;
}

// ---- AT, XT -> intra-normalised descriptor --------------------------------------------------------
constexpr int VT_WS = 36;    // per-wave LDS tile row stride (floats): 144 B keeps float4 accesses aligned and
                             // conflict-free for the "one row per lane" walk (144*i mod 256 are 16 distinct slots)

// The main loop is pure streaming: every wave owns NT d-tiles (NT*32 columns) x 32 clusters of accumulators and
// fetches its own fragments straight into registers; a step's second half is in flight while the first computes.
// Measured alternatives at cfg-2: one wave per SIMD with a 2-deep hand-counted asm load ring
// 122 us, 4-deep compiler-managed ring 126 us (hipcc drains it to vmcnt(0) every step) -- this form, two waves per
// SIMD sharing the matrix pipe, 99.5 us.  HBM traffic is already ~algorithmic (rocprofv3 FETCH/WRITE_SIZE: 146 + 84 MB
// vs 124 + 84 MB); the kernel is bound by L2 -> CU delivery of the 8x slab re-read of x (1.09 GB of L2 requests).
template <int NT>
__global__ __launch_bounds__(256, 2) void vlad_aggregate_tiles_kernel(
    const uint4* __restrict__ at, const uint4* __restrict__ xt, const float* __restrict__ centres, int B, int T,
    int K, int S, int KT, int residual, float* __restrict__ nrm, float* __restrict__ asum,
    float* __restrict__ colsq, float* __restrict__ csq) {
    constexpr int D = NT * 128, DT = NT * 4;
    constexpr int G = (NT >= 2) ? 2 : 1;     // software-pipeline groups per step
    constexpr int TG = NT / G;               // d-tiles per group
    __shared__ __attribute__((aligned(16))) float wt[4][32 * VT_WS];
    __shared__ float sred[4][32];
    __shared__ float ssum[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / KT, kt = lid % KT;
    const int k0 = kt * 32;

    const u32x4* ap = reinterpret_cast<const u32x4*>(at) + (((int64_t)b * KT + kt) * S) * 128 + lane;          // + s*128 (+64 lo)
    const u32x4* xp = reinterpret_cast<const u32x4*>(xt) + (((int64_t)b * S) * DT + wave * NT) * 128 + lane;   // + (s*DT+tile)*128

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float asum_l = 0.f;

    u32x4 ah, al, nah, nal;
    u32x4 xh[G][TG], xl[G][TG];
    auto load_a = [&](int s, u32x4& h, u32x4& l) {
        h = ap[(int64_t)s * 128];
        l = ap[(int64_t)s * 128 + 64];
    };
    auto load_x = [&](int s, int g) {
#pragma unroll
        for (int i = 0; i < TG; ++i) {
            const u32x4* p = xp + ((int64_t)s * DT + g * TG + i) * 128;
            xh[g][i] = p[0];
            xl[g][i] = p[64];
        }
    };
    auto compute = [&](int g, const u32x4& h, const u32x4& l) {
#pragma unroll
        for (int i = 0; i < TG; ++i) {
            acc[g * TG + i] = mfma_bf16(h, xh[g][i], acc[g * TG + i]);
            acc[g * TG + i] = mfma_bf16(h, xl[g][i], acc[g * TG + i]);
            acc[g * TG + i] = mfma_bf16(l, xh[g][i], acc[g * TG + i]);
        }
    };
    auto add_asum = [&](const u32x4& h, const u32x4& l) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            asum_l += (bf16_to_f32(h[q] & 0xffffu) + bf16_to_f32(l[q] & 0xffffu)) +
                      (bf16_to_f32(h[q] >> 16) + bf16_to_f32(l[q] >> 16));
    };

    load_a(0, ah, al);
    load_x(0, 0);
    for (int s = 0; s < S; ++s) {
        if constexpr (G == 2) {
            load_x(s, 1);                       // second half of this step in flight while the first computes
            compute(0, ah, al);
            add_asum(ah, al);
            if (s + 1 < S) {
                load_a(s + 1, nah, nal);
                load_x(s + 1, 0);
            }
            compute(1, ah, al);
            ah = nah;
            al = nal;
        } else {
            const u32x4 ch = xh[0][0], cl = xl[0][0], cah = ah, cal = al;
            if (s + 1 < S) {
                load_a(s + 1, ah, al);
                load_x(s + 1, 0);
            }
            acc[0] = mfma_bf16(cah, ch, acc[0]);
            acc[0] = mfma_bf16(cah, cl, acc[0]);
            acc[0] = mfma_bf16(cal, ch, acc[0]);
            add_asum(cah, cal);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    asum_l += __shfl_xor(asum_l, 32, 64);            // sum_t a[t, k0 + l31]
    if (wave == 0 && half == 0) ssum[l31] = asum_l;
    __syncthreads();
    float sk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sk[r] = ssum[mfma32_row(r, lane)];

    float* wl = wt[wave];
    const int d0w = wave * (NT * 32);
    float part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (residual) {
            // stage the wave's 32(d) x 32(k) tile of the centres through LDS: coalesced 128-byte rows in,
            // accumulator-layout float4 reads out
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 8 * it + (lane >> 3), c4 = (lane & 7) * 4;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + c4 < K) w = *reinterpret_cast<const float4*>(centres + (int64_t)(d0w + t * 32 + row) * K + k0 + c4);
                *reinterpret_cast<float4*>(wl + row * VT_WS + c4) = w;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(wl + l31 * VT_WS + 8 * q + 4 * half);
                acc[t][4 * q + 0] -= sk[4 * q + 0] * w.x;
                acc[t][4 * q + 1] -= sk[4 * q + 1] * w.y;
                acc[t][4 * q + 2] -= sk[4 * q + 2] * w.z;
                acc[t][4 * q + 3] -= sk[4 * q + 3] * w.w;
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = fmaf(acc[t][r], acc[t][r], part[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = half_sum(part[r]);
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sred[wave][mfma32_row(r, lane)] = part[r];
    }
    __syncthreads();
    float inv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mfma32_row(r, lane);
        inv[r] = rsqrtf(fmaxf(sred[0][row] + sred[1][row] + sred[2][row] + sred[3][row], kL2Eps));
    }
    if (tid < 32 && (k0 + tid) < K) {
        const float n = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        const float iv = rsqrtf(fmaxf(n, kL2Eps));
        const int64_t o = (int64_t)b * K + k0 + tid;
        asum[o] = ssum[tid];
        colsq[o] = n;
        csq[o] = n * iv * iv;
    }
    float* ob = nrm + (int64_t)b * D * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // accumulator layout -> LDS -> whole 128-byte rows of the d-major descriptor
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o;
            o.x = acc[t][4 * q + 0] * inv[4 * q + 0];
            o.y = acc[t][4 * q + 1] * inv[4 * q + 1];
            o.z = acc[t][4 * q + 2] * inv[4 * q + 2];
            o.w = acc[t][4 * q + 3] * inv[4 * q + 3];
            *reinterpret_cast<float4*>(wl + l31 * VT_WS + 8 * q + 4 * half) = o;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = 8 * it + (lane >> 3), c4 = (lane & 7) * 4;
            if (k0 + c4 < K) {
                const float4 o = *reinterpret_cast<const float4*>(wl + row * VT_WS + c4);
                *reinterpret_cast<float4*>(ob + (int64_t)(d0w + t * 32 + row) * K + k0 + c4) = o;
            }
        }
        __syncthreads();
    }
}

}  // namespace lpm

static inline int vt_steps(int T) { return (T + 15) / 16; }

extern "C" size_t lpm_xt_bytes(int B, int T, int D) { return (size_t)B * vt_steps(T) * (D / 32) * 2048; }
extern "C" size_t lpm_at_bytes(int B, int T, int K) { return (size_t)B * ((K + 31) / 32) * vt_steps(T) * 2048; }

extern "C" int lpm_split_frames(const float* x, int64_t ldx, int B, int T, int D, void* xt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && xt, LPM_ERR_BADARG, "lpm_split_frames: null pointer");
    LPM_REQUIRE(B > 0 && T > 0 && D > 0 && ldx >= D, LPM_ERR_BADARG, "lpm_split_frames: bad sizes");
    LPM_REQUIRE(D % 32 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)xt) & 15) == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_split_frames: need D %% 32 == 0, ldx %% 4 == 0, 16-byte aligned pointers (D=%d)", D);
    const int S = vt_steps(T);
    const int64_t total = (int64_t)B * S * 2 * (D / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(split_frames_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, B, T, D, S, (uint4*)xt, 2);
    return check_launch("lpm_split_frames");
}

// bf16 storage: fp32 rows -> plain bf16 frame tiles [b][step][column tile][lane], 4 ceil(T / 64) steps per clip (zero beyond T);
// xt holds lpm_frame_tiles_bf16_bytes(B, T, D) bytes.
extern "C" int lpm_split_frames_bf16(const float* x, int64_t ldx, int B, int T, int D, void* xt, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && xt, LPM_ERR_BADARG, "lpm_split_frames_bf16: null pointer");
    LPM_REQUIRE(B > 0 && T > 0 && D > 0 && ldx >= D, LPM_ERR_BADARG, "lpm_split_frames_bf16: bad sizes");
    LPM_REQUIRE(D % 32 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)xt) & 15) == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_split_frames_bf16: need D %% 32 == 0, ldx %% 4 == 0, 16-byte aligned pointers (D=%d)", D);
    const int S = 4 * ((T + 63) / 64);
    const int64_t total = (int64_t)B * S * 2 * (D / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(split_frames_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, B, T, D, S, (uint4*)xt, 1);
    return check_launch("lpm_split_frames_bf16");
}

extern "C" int lpm_assign_tiles(const float* assign, const float* scale, const float* shift, int B, int T, int K, int flags,
                                void* at, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(assign && at, LPM_ERR_BADARG, "lpm_assign_tiles: null pointer");
    LPM_REQUIRE(B > 0 && T > 0 && K > 0 && K <= 1024, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_assign_tiles: need 0 < K <= 1024 (K=%d)", K);
    const int S = vt_steps(T), KT = (K + 31) / 32;
    if (launch_assign_tiles2<false>(assign, scale, shift, B, T, K, S, (flags & LPM_VLAD_SOFTMAX) ? 1 : 0, at, (hipStream_t)stream,
                                    K >= 256 ? LPM_TIMING_ASSIGN_TILES : 0))
        return check_launch("lpm_assign_tiles");
    const size_t lds = (size_t)16 * (KT * 32 + 1) * sizeof(float);
    hipEvent_t e0, e1;
    if ((flags & LPM_VLAD_SOFTMAX) && K >= 256 && timing_request(LPM_TIMING_ASSIGN_TILES, &e0, &e1))
        hipExtLaunchKernelGGL((assign_tiles_kernel<true, false>), dim3(B * S), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, assign, scale,
                              shift, T, K, S, KT, (uint4*)at);
    else if (flags & LPM_VLAD_SOFTMAX)
        hipLaunchKernelGGL((assign_tiles_kernel<true, false>), dim3(B * S), dim3(256), lds, (hipStream_t)stream, assign, scale, shift, T, K, S,
                           KT, (uint4*)at);
    else
        hipLaunchKernelGGL((assign_tiles_kernel<false, false>), dim3(B * S), dim3(256), lds, (hipStream_t)stream, assign, scale, shift, T, K,
                           S, KT, (uint4*)at);
    return check_launch("lpm_assign_tiles");
}

// bf16 storage: assign is bf16 [B*T, K]; AT = plain bf16 tiles [b][cluster tile][step][lane] with 4 ceil(T / 64) steps per clip
// (B * ceil(K / 32) * steps * 1024 bytes).
extern "C" int lpm_assign_tiles_bf16(const void* assign_bf16, const float* scale, const float* shift, int B, int T, int K, int flags,
                                     void* at, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(assign_bf16 && at, LPM_ERR_BADARG, "lpm_assign_tiles_bf16: null pointer");
    LPM_REQUIRE(B > 0 && T > 0 && K > 0 && K <= 1024, LPM_ERR_UNSUPPORTED_SHAPE, "lpm_assign_tiles_bf16: need 0 < K <= 1024 (K=%d)", K);
    const int S = 4 * ((T + 63) / 64), KT = (K + 31) / 32;
    if (launch_assign_tiles2<true>(assign_bf16, scale, shift, B, T, K, S, (flags & LPM_VLAD_SOFTMAX) ? 1 : 0, at, (hipStream_t)stream,
                                   K >= 256 ? LPM_TIMING_ASSIGN_TILES : 0))
        return check_launch("lpm_assign_tiles_bf16");
    const size_t lds = (size_t)16 * (KT * 32 + 1) * sizeof(float);
    hipEvent_t e0, e1;
    if ((flags & LPM_VLAD_SOFTMAX) && K >= 256 && timing_request(LPM_TIMING_ASSIGN_TILES, &e0, &e1))
        hipExtLaunchKernelGGL((assign_tiles_kernel<true, true>), dim3(B * S), dim3(256), lds, (hipStream_t)stream, e0, e1, 0,
                              (const float*)assign_bf16, scale, shift, T, K, S, KT, (uint4*)at);
    else if (flags & LPM_VLAD_SOFTMAX)
        hipLaunchKernelGGL((assign_tiles_kernel<true, true>), dim3(B * S), dim3(256), lds, (hipStream_t)stream, (const float*)assign_bf16,
                           scale, shift, T, K, S, KT, (uint4*)at);
    else
        hipLaunchKernelGGL((assign_tiles_kernel<false, true>), dim3(B * S), dim3(256), lds, (hipStream_t)stream, (const float*)assign_bf16,
                           scale, shift, T, K, S, KT, (uint4*)at);
    return check_launch("lpm_assign_tiles_bf16");
}

extern "C" int lpm_vlad_aggregate_tiles_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                            int flags, float* nrm, float* asum, float* colsq, float* csq,
                                            lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(at && xt && nrm && asum && colsq && csq, LPM_ERR_BADARG, "lpm_vlad_aggregate_tiles_fwd: null pointer");
    const int residual = (flags & LPM_VLAD_RESIDUAL) ? 1 : 0;
    LPM_REQUIRE(!residual || centres, LPM_ERR_BADARG, "lpm_vlad_aggregate_tiles_fwd: RESIDUAL needs centres");
    LPM_REQUIRE(B > 0 && T > 0 && K > 0, LPM_ERR_BADARG, "lpm_vlad_aggregate_tiles_fwd: bad sizes");
    LPM_REQUIRE((D == 128 || D == 256 || D == 512 || D == 1024) && K % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_vlad_aggregate_tiles_fwd: need D in {128,256,512,1024} and K %% 4 == 0 (D=%d K=%d)", D, K);
    LPM_REQUIRE((((uintptr_t)at | (uintptr_t)xt | (uintptr_t)centres | (uintptr_t)nrm) & 15) == 0, LPM_ERR_BADARG,
                "lpm_vlad_aggregate_tiles_fwd: pointers must be 16-byte aligned");
    const int S = vt_steps(T), KT = (K + 31) / 32;
    dim3 grid(B * KT);
    hipStream_t s = (hipStream_t)stream;
#define LPM_VT_LAUNCH(NT)                                                                                              \
    hipLaunchKernelGGL((vlad_aggregate_tiles_kernel<NT>), grid, dim3(256), 0, s, (const uint4*)at, (const uint4*)xt, centres, B, \
                       T, K, S, KT, residual, nrm, asum, colsq, csq)
    switch (D) {
        case 128: LPM_VT_LAUNCH(1); break;
        case 256: LPM_VT_LAUNCH(2); break;
        case 512: LPM_VT_LAUNCH(4); break;
        default: LPM_VT_LAUNCH(8); break;
    }
#undef LPM_VT_LAUNCH
    return check_launch("lpm_vlad_aggregate_tiles_fwd");
}
