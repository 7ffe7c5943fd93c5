// Every operand form of every dense-layer weight of a training step in ONE launch.
//
// The encoders' dense layers (transformer_utils.py:559-561,583,701-711) and the soft-assignment GEMM (frame_level_models.py:2781-2789)
// read their weights as split-bf16 operand images / fragment tiles (split_gemm.hip, tile_gemm.hip).  A weight changes once per step --
// in the optimiser -- yet round 3 re-derived each form where it was consumed: 14 launches of lpm_split_weight /
// lpm_split_weight_tiles per cfg-2 step, 4-32 us each for 0.1-16 MB (one of them a strided transposing read: 31 us for 16 MB),
// ~150 us of a step's 8 ms and 8 % of its launches.  lpm_weight_pack takes a list of jobs -- a weight (or a column block of a
// concatenated weight: q | k | v) and the forms wanted of it -- and writes all of them from ONE read of each 32 x 32 tile:
//   w3n  [Ntot, 3K]   rows [Wh^T | Wh^T | Wl^T]            y  = X3 w3n^T          (lpm_split_weight)
//   w3k  [K, 3 Ntot]  rows [Wh | Wl | Wh]                  dx = DY3 w3k^T         (lpm_split_weight)
//   wt   weight tiles of W  [K/16][Ntot/32][plane][lane]   tile GEMM B operand    (lpm_split_weight_tiles, transposed = 0)
//   wtt  weight tiles of W^T [N/16][K/32][plane][lane]     tile GEMM B operand of the input gradient (lpm_split_weight_tiles of the
//                                                          same storage with transposed = 1)
// Bit for bit the outputs of the single-weight entry points (same round-to-nearest-even split).
// Round 5: a job of an fp16 kind writes the fp16 forms: w3n [Ntot, 3K] = [Wh^T | Wh^T | Wl^T] (forward, three-term), w3k [K, 2 Ntot] =
// [Wh | Wh] (input gradient, two-term: the weight rounded once), wt with fp16 (hi, lo) planes, wtt the fp16 hi plane only (64 units per
// tile and step).
#include "lpm_common.h"
#include "operand_format.h"

namespace lpm {

__device__ __forceinline__ unsigned wp_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

struct WPArgs {
    int njobs;
    int first[LPM_WEIGHT_PACK_MAX_JOBS + 1];       // first workgroup of job j; first[njobs] = grid size
    LpmWeightPackJob job[LPM_WEIGHT_PACK_MAX_JOBS];
};

// one workgroup = one 32 (k) x 32 (n) tile of one job's source
__global__ __launch_bounds__(256) void weight_pack_kernel(const WPArgs a) {
    __shared__ unsigned short th[32][34], tl[32][34];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.first[j + 1]) ++j;      // workgroup-uniform, <= 15 steps
    const LpmWeightPackJob& g = a.job[j];
    const int t = (int)blockIdx.x - a.first[j];
    const int tiles_n = g.N / 32;
    const int kt = t / tiles_n, nt = t % tiles_n;
    const int k0 = kt * 32, n0 = nt * 32;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    unsigned short* w3k = (unsigned short*)g.w3k;
    unsigned short* w3n = (unsigned short*)g.w3n;
    const int f16 = g.kind != LPM_OPERAND_BF16X3 ? 1 : 0;          // workgroup-uniform
    const int npk = f16 ? 2 : 3;                                   // planes of a w3k row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kl = ty + 8 * i;
        const float v = g.w[(int64_t)(k0 + kl) * g.ldw + n0 + tx];
        unsigned h, l;
        if (f16) {
            of_split1_f16(v, h, l);
        } else {
            h = wp_rne(v);
            l = wp_rne(v - __uint_as_float(h << 16));
        }
        th[kl][tx] = (unsigned short)h;
        tl[kl][tx] = (unsigned short)l;
    }
    __syncthreads();
    // the two images leave as 16-byte pieces (8 bf16): threads 0-127 write the w3k rows (row k: 8 consecutive n), threads 128-255 the w3n
    // rows (row n: 8 consecutive k, the tile read down its columns) -- 2-byte stores made this kernel 51 us for 12 jobs at cfg-2
    {
        const int r = (tid & 127) >> 2, c8 = (tid & 3) * 8;
        unsigned hv[4], lv[4];
        if (tid < 128) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hv[e] = (unsigned)th[r][c8 + 2 * e] | ((unsigned)th[r][c8 + 2 * e + 1] << 16);
                lv[e] = (unsigned)tl[r][c8 + 2 * e] | ((unsigned)tl[r][c8 + 2 * e + 1] << 16);
            }
            if (w3k) {
                unsigned short* row = w3k + (int64_t)(k0 + r) * npk * g.Ntot + g.n_off + n0 + c8;
                const uint4 H = make_uint4(hv[0], hv[1], hv[2], hv[3]), L = make_uint4(lv[0], lv[1], lv[2], lv[3]);
                *reinterpret_cast<uint4*>(row) = H;
                *reinterpret_cast<uint4*>(row + g.Ntot) = f16 ? H : L;
                if (!f16) *reinterpret_cast<uint4*>(row + 2 * (int64_t)g.Ntot) = H;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hv[e] = (unsigned)th[c8 + 2 * e][r] | ((unsigned)th[c8 + 2 * e + 1][r] << 16);
                lv[e] = (unsigned)tl[c8 + 2 * e][r] | ((unsigned)tl[c8 + 2 * e + 1][r] << 16);
            }
            if (w3n) {
                unsigned short* row = w3n + (int64_t)(g.n_off + n0 + r) * 3 * g.K + k0 + c8;
                const uint4 H = make_uint4(hv[0], hv[1], hv[2], hv[3]), L = make_uint4(lv[0], lv[1], lv[2], lv[3]);
                *reinterpret_cast<uint4*>(row) = H;
                *reinterpret_cast<uint4*>(row + g.K) = H;
                *reinterpret_cast<uint4*>(row + 2 * (int64_t)g.K) = L;
            }
        }
    }
    // fragment tiles: thread = (16-deep step s2, plane, lane); tile[plane][lane][e] = M[outer = 32 tile + (lane & 31)][red = 16 step + 8 (lane >> 5) + e]
    const int s2 = tid >> 7, plane = (tid >> 6) & 1, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int tp = f16 ? 1 : 2;                                     // planes per fragment tile of W^T (fp16: the hi plane only)
    if (g.wt) {          // weight tiles of W: outer = column n, reduction = row k
        unsigned v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kl = 16 * s2 + 8 * half + 2 * e;
            const unsigned lo = plane ? tl[kl][l31] : th[kl][l31], hi = plane ? tl[kl + 1][l31] : th[kl + 1][l31];
            v[e] = lo | (hi << 16);
        }
        const int64_t step = (int64_t)(k0 / 16 + s2), tile = (g.n_off + n0) / 32;
        ((uint4*)g.wt)[((step * (g.Ntot / 32) + tile) * 2 + plane) * 64 + lane] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (g.wtt && plane < tp) {         // weight tiles of W^T: outer = row k, reduction = column n
        unsigned v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int nl = 16 * s2 + 8 * half + 2 * e;
            const unsigned lo = plane ? tl[l31][nl] : th[l31][nl], hi = plane ? tl[l31][nl + 1] : th[l31][nl + 1];
            v[e] = lo | (hi << 16);
        }
        const int64_t step = (int64_t)(n0 / 16 + s2), tile = k0 / 32;
        ((uint4*)g.wtt)[((step * (g.K / 32) + tile) * tp + plane) * 64 + lane] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace lpm

extern "C" int lpm_weight_pack(const LpmWeightPackJob* jobs, int njobs, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(jobs && njobs > 0 && njobs <= LPM_WEIGHT_PACK_MAX_JOBS, LPM_ERR_BADARG, "lpm_weight_pack: 1 .. %d jobs (got %d)",
                LPM_WEIGHT_PACK_MAX_JOBS, njobs);
    WPArgs a{};
    a.njobs = njobs;
    int64_t total = 0;
    for (int j = 0; j < njobs; ++j) {
        const LpmWeightPackJob& g = jobs[j];
        LPM_REQUIRE(g.w && g.K > 0 && g.N > 0 && g.K % 32 == 0 && g.N % 32 == 0 && g.ldw >= g.N, LPM_ERR_UNSUPPORTED_SHAPE,
                    "lpm_weight_pack: job %d needs K %% 32 == 0, N %% 32 == 0, ldw >= N (K=%d N=%d)", j, g.K, g.N);
        LPM_REQUIRE(g.Ntot >= g.n_off + g.N && g.n_off >= 0 && g.n_off % 32 == 0 && g.Ntot % 32 == 0, LPM_ERR_BADARG,
                    "lpm_weight_pack: job %d: column block [%d, %d) of %d", j, g.n_off, g.n_off + g.N, g.Ntot);
        LPM_REQUIRE(g.w3n || g.w3k || g.wt || g.wtt, LPM_ERR_BADARG, "lpm_weight_pack: job %d asks for nothing", j);
        LPM_REQUIRE(operand_kind_ok(g.kind), LPM_ERR_BADARG, "lpm_weight_pack: job %d: unknown operand format %d", j, g.kind);
        LPM_REQUIRE(!g.wtt || (g.n_off == 0 && g.Ntot == g.N), LPM_ERR_UNSUPPORTED_SHAPE,
                    "lpm_weight_pack: job %d: the transposed tiles are written for whole weights only", j);
        LPM_REQUIRE((((uintptr_t)g.wt | (uintptr_t)g.wtt | (uintptr_t)g.w3n | (uintptr_t)g.w3k) & 15) == 0, LPM_ERR_BADARG,
                    "lpm_weight_pack: job %d: misaligned output", j);
        a.first[j] = (int)total;
        a.job[j] = g;
        total += (int64_t)(g.K / 32) * (g.N / 32);
        LPM_REQUIRE(total < ((int64_t)1 << 30), LPM_ERR_UNSUPPORTED_SHAPE, "lpm_weight_pack: too many tiles");
    }
    a.first[njobs] = (int)total;
    hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("lpm_weight_pack");
}
