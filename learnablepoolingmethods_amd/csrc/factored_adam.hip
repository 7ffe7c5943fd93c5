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
        for (int s = 0; s < S; ++s) {
            const uint4 h4 = xt[(((int64_t)s * NT1 + tile) * 2 + 0) * 64 + lane], l4 = xt[(((int64_t)s * NT1 + tile) * 2 + 1) * 64 + lane];
            const tg_u32x4 xh = {h4.x, h4.y, h4.z, h4.w}, xl = {l4.x, l4.y, l4.z, l4.w};
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

}  // namespace lpm

extern "C" size_t lpm_factored_clip_adam_scratch_bytes(int N1, int N2) {
    return (size_t)(lpm::fa_workgroups(N1, N2) + 4) * sizeof(float);
}

static int factored_clip_adam_impl(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                   float* param, float* m, float* v, float clip_norm, float lr, float beta1, float beta2, float eps,
                                   int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream);
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
                                   int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream) {
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
        if (hipFuncSetAttribute((const void*)fa_quadform_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("lpm_factored_clip_adam_q: cannot reserve %zu bytes of LDS", lds);
            return LPM_ERR_LAUNCH;
        }
        npart = FQ_GRID < nwg ? FQ_GRID : nwg;
        hipLaunchKernelGGL(fa_quadform_kernel, dim3((unsigned)npart), dim3(256), lds, s, (const uint4*)xt, x, ldx, (const uint4*)gdt, R, N1, NT1, MT,
                           partial);
    } else {
        g.sumsq = partial;
        rc = (dbg & 2) ? LPM_OK : tile_gemm_store(g, 1, 1, s, "lpm_factored_clip_adam (norm pass)", 1);
        if (rc != LPM_OK) return rc;
    }
    hipLaunchKernelGGL(fa_factor_kernel, dim3(1), dim3(1024), 0, s, (const float*)partial, npart, clip_norm, factor);
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    g.sumsq = nullptr;
    g.adam_p = param; g.adam_m = m; g.adam_v = v; g.adam_factor = factor;
    g.adam_lr_t = (float)lr_t; g.adam_b1 = beta1; g.adam_b2 = beta2; g.adam_eps = eps;
    rc = tile_gemm_adam(g, s, "lpm_factored_clip_adam (update pass)");
    if (rc != LPM_OK) return rc;
    return check_launch("lpm_factored_clip_adam");
}
