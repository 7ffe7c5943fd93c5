// Shared device/host helpers for liblpm_hip.so (gfx950 only: 64-wide wavefronts, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/lpm_hip.h"

namespace lpm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;
constexpr float kL2Eps = 1e-12f;  // tf.nn.l2_normalize epsilon

// thread-local error string behind lpm_last_error()
void set_error(const char* fmt, ...);
// kernel timing (lpm_api.hip): when enabled, hands out a start/stop event pair for a hipExtLaunchKernelGGL launch
enum { LPM_TIMING_K1 = 1, LPM_TIMING_K2 = 2, LPM_TIMING_ASSIGN_TILES = 3, LPM_TIMING_FINALIZE = 4 };   // (video-stream launches)
bool timing_request(int tag, hipEvent_t* e0, hipEvent_t* e1);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LPM_ERR_LAUNCH;
    }
    return LPM_OK;
}

#define LPM_REQUIRE(cond, code, ...)  \
    do {                              \
        if (!(cond)) {                \
            lpm::set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

// ---- column sums of per-block partials (the second stage of every two-stage column reduction) --------------------------------
// part: nblk rows of `row` floats; the sums over the rows of part[.][c] (-> s) and, if off2 > 0, part[.][off2 + c] (-> q), in fp64
// and in a fixed order.  For a 1024-thread workgroup that owns 16 columns (c = 16 blockIdx.x + (tid & 15)): 64 row groups, every
// thread loads its rows eight at a time (nblk = 400: ONE round of independent loads -- a 64-column / 16-row-group form walked
// seven dependent rounds of ~2 us, 14-30 us for under a megabyte of input), the four row groups of a wave meet by shuffles,
// the sixteen waves through LDS.  The sums come back on the threads with tid < 16.
// partial_colsums<4>: the same for a VERY tall array (nblk = batch x heads = 5120 rows of the attention's logits_bn statistics): four
// columns per workgroup, 256 row groups -- four times the workgroups (75 instead of 19 at 300 columns: the 16-column form took 60-76 us
// for 12 MB there, one CU's worth of loads in flight per workgroup).  The sums come back on the threads with tid < CW.
template <int CW>
__device__ __forceinline__ void partial_colsums(const float* __restrict__ part, int nblk, int64_t row, int off2, int C, double& s,
                                                double& q, int& c) {
    static_assert(CW == 16 || CW == 4, "columns per workgroup");
    constexpr int RG = 1024 / CW;                  // row groups
    __shared__ double pcs_sh[2][16][CW];
    const int cl = threadIdx.x & (CW - 1), rg = threadIdx.x / CW;
    c = blockIdx.x * CW + cl;
    s = 0.0; q = 0.0;
    if (c < C) {
        for (int b = rg; b < nblk; b += RG * 8) {
            float ps[8], pq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = b + RG * u;
                const float* p = part + (int64_t)min(bb, nblk - 1) * row + c;
                ps[u] = (bb < nblk) ? p[0] : 0.f;
                pq[u] = (bb < nblk && off2 > 0) ? p[off2] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s += (double)ps[u];
                q += (double)pq[u];
            }
        }
    }
#pragma unroll
    for (int o = CW; o < 64; o <<= 1) {            // lanes that differ only above the column bits hold the same column
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < CW) {
        pcs_sh[0][wave][cl] = s;
        pcs_sh[1][wave][cl] = q;
    }
    __syncthreads();
    if (threadIdx.x < CW) {
        s = pcs_sh[0][0][cl]; q = pcs_sh[1][0][cl];
        for (int i = 1; i < 16; ++i) {
            s += pcs_sh[0][i][cl];
            q += pcs_sh[1][i][cl];
        }
    }
}
__device__ __forceinline__ void partial_colsums16(const float* __restrict__ part, int nblk, int64_t row, int off2, int C, double& s,
                                                  double& q, int& c) {
    partial_colsums<16>(part, nblk, row, off2, C, s, q, c);
}
// columns per workgroup for a partial array of nblk rows
__host__ __device__ constexpr int partial_colsums_cw(int nblk) { return nblk >= 2048 ? 4 : 16; }

// ---- wave-level reductions (64 lanes) -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// ---- wave-wide maximum / sum on the VALU (round 4; the softmax of lpm_assign_tiles and of the in-kernel-softmax K2's row statistics -- the two
// must add in the SAME order: tests hold them bitwise equal): four DPP steps give every lane its row-of-16 total (quad permutes, half-row mirror, row
// mirror), four v_readlane + three operations join the rows.  __shfl_xor is ds_bpermute on this target: the softmax's two butterflies
// were 48 dependent LDS-crossbar round trips per wave (four rows x two reductions x six steps) in a kernel whose rate is its latency.
template <int CTRL>
__device__ __forceinline__ float dpp_lanes(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_lanes<0xB1>(v));
    v = fmaxf(v, dpp_lanes<0x4E>(v));
    v = fmaxf(v, dpp_lanes<0x141>(v));
    v = fmaxf(v, dpp_lanes<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_lanes<0xB1>(v);
    v += dpp_lanes<0x4E>(v);
    v += dpp_lanes<0x141>(v);
    v += dpp_lanes<0x140>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// reduce across the 32 lanes that share (lane >> 5): lanes of one half-wave
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact fp32.
//   A operand: lane l supplies A[i = l & 31][k = l >> 5];  B operand: B[k = l >> 5][j = l & 31]
//   C/D: lane l, reg r holds D[i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][j = l & 31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32.
//   A: lane l supplies A[i = l & 15][k = l >> 4];  B: B[k = l >> 4][j = l & 15]
//   C/D: lane l, reg r holds D[i = 4 * (l >> 4) + r][j = l & 15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// TF-Adam on one element given its CLIPPED gradient gc (train.py:332-336 -> tf.train.AdamOptimizer: m, v, then the step with both bias
// corrections folded into lr_t).  Every product and sum is rounded on its own -- no fused multiply-add: WHICH product of
// "b1 m + (1 - b1) g" the compiler fuses differed between two kernels holding the same source line (round 6: the row-block update pass
// against the tile-GEMM form -- one ulp of m in a quarter of the elements) -- so that every update kernel of the library (clip_adam.hip,
// tile_gemm.hip's Adam epilogue, factored_adam.hip's two other forms) writes the same bits for the same inputs.
__device__ __forceinline__ void adam_element(float gc, float& p, float& m, float& v, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    m = b1 * m + (1.f - b1) * gc;
    v = b2 * v + (1.f - b2) * gc * gc;
    p = p - lr_t * m / (sqrtf(v) + eps);
}

// XCD-aware block remap: hardware places consecutive block ids round-robin over the 8 XCDs
// (speed only, never correctness).  Returns a logical id such that logical ids
// [g*per, (g+1)*per) of one group land on the same XCD when nblk % 8 == 0 handling is
// bijective for any nblk (cdna guide 5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace lpm
