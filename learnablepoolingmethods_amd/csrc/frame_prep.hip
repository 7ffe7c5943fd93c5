// a2 + a3 -- SampleUniformFrames fused with the input batch-norm.
//   idx[b,j] = int32(fp32(j * fp32(1/S)) * fp32(num_frames[b]))      model_utils.py:112-118
//   y[b*S+j, :] = raw[b, idx[b,j], :] * scale + shift                 model_utils.py:119-122 +
//                                                                      frame_level_models.py:2265-2271
// The gathered [B,S,F] tensor is never materialised un-normalised: one pass reduces the column
// statistics of the gathered rows (per-block partials -> lpm_bn_fold), a second pass gathers again
// and writes the normalised rows that K1 and K2 consume.  Both are pure HBM streams (float4,
// row-contiguous 4.6 KB reads).
#include "lpm_common.h"

namespace lpm {

constexpr int FP_ROWS = 32;  // gathered rows per statistics block

__device__ __forceinline__ int sample_index(int j, float step, int nf) {
    // fp32 product then truncation toward zero, exactly as tf.linspace * num_frames -> tf.cast(int32)
    const float v = __fmul_rn((float)j, step);
    return (int)__fmul_rn(fminf(v, 1.0f), (float)nf);
}

__global__ __launch_bounds__(256) void frame_stats_kernel(const float* __restrict__ raw,
                                                          const int32_t* __restrict__ num_frames, int B,
                                                          int max_frames, int F, int S, float step,
                                                          float* __restrict__ partial) {
    const int r0 = blockIdx.x * FP_ROWS;
    const int rows = B * S;
    const int r1 = min(rows, r0 + FP_ROWS);
    __shared__ int64_t base[FP_ROWS];
    if (threadIdx.x < FP_ROWS) {
        const int r = r0 + threadIdx.x;
        if (r < rows) {
            const int b = r / S, j = r % S;
            int idx = sample_index(j, step, num_frames[b]);
            idx = max(0, min(idx, max_frames - 1));
            base[threadIdx.x] = ((int64_t)b * max_frames + idx) * F;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += 256) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < r1 - r0; ++r) {
            const float v = raw[base[r] + c];
            s += v;
            q = fmaf(v, v, q);
        }
        float* p = partial + (int64_t)blockIdx.x * 2 * F;
        p[c] = s;
        p[F + c] = q;
    }
}

__global__ __launch_bounds__(256) void frame_apply_kernel(const float* __restrict__ raw,
                                                          const int32_t* __restrict__ num_frames, int B,
                                                          int max_frames, int F, int S, float step,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ y) {
    const int F4 = F / 4;
    const int64_t total = (int64_t)B * S * F4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / F4), c = (int)(i % F4) * 4;
        const int b = r / S, j = r % S;
        int idx = sample_index(j, step, num_frames[b]);
        idx = max(0, min(idx, max_frames - 1));
        float4 v = *reinterpret_cast<const float4*>(raw + ((int64_t)b * max_frames + idx) * F + c);
        if (scale) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + c);
            const float4 sh = *reinterpret_cast<const float4*>(shift + c);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        *reinterpret_cast<float4*>(y + (int64_t)r * F + c) = v;
    }
}

// Same as frame_apply_kernel, but one work item owns 8 consecutive sampled frames of 4 columns so that it can
// ALSO emit the split-bf16 MFMA-fragment tiles K2 consumes (vlad_tiles.hip: XT[b][s][d/32][plane][lane][8]) for
// the rgb columns [0, Dv) and the audio columns [Dv, Dv+Da) -- the normalised frames are then written once in
// fp32 (for K1 / the backward GEMMs) and once in tile order, with no separate re-read for the split.
__device__ __forceinline__ unsigned fp_bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__global__ __launch_bounds__(256) void frame_apply_tiles_kernel(const float* __restrict__ raw,
                                                                const int32_t* __restrict__ num_frames, int B,
                                                                int max_frames, int F, int S, float step,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* __restrict__ y,
                                                                uint4* __restrict__ xtv, int Dv,
                                                                uint4* __restrict__ xta, int Da, float* __restrict__ y2) {
    // y2 != null: the two column blocks leave as TWO contiguous matrices, y [B S, Dv] and y2 [B S, Da] (NetVladV2: each stream's encoder
    // and aggregation want whole rows -- no strided slices, no contiguous copies of them)
    const int F4 = F / 4, NS = (S + 15) / 16;
    const int64_t total = (int64_t)B * NS * 2 * F4;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(w % F4);
        const int64_t r = w / F4;
        const int kh = (int)(r & 1), st = (int)((r >> 1) % NS), b = (int)((r >> 1) / NS);
        const int c = 4 * c4;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) {
            sc = *reinterpret_cast<const float4*>(scale + c);
            sh = *reinterpret_cast<const float4*>(shift + c);
        }
        const int nf = num_frames[b];
        float v[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 16 * st + 8 * kh + e;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < S) {
                int idx = sample_index(j, step, nf);
                idx = max(0, min(idx, max_frames - 1));
                f = *reinterpret_cast<const float4*>(raw + ((int64_t)b * max_frames + idx) * F + c);
                f.x = fmaf(f.x, sc.x, sh.x); f.y = fmaf(f.y, sc.y, sh.y);
                f.z = fmaf(f.z, sc.z, sh.z); f.w = fmaf(f.w, sc.w, sh.w);
                const int64_t row = (int64_t)b * S + j;
                if (!y2) *reinterpret_cast<float4*>(y + row * F + c) = f;
                else if (c < Dv) *reinterpret_cast<float4*>(y + row * Dv + c) = f;
                else *reinterpret_cast<float4*>(y2 + row * Da + (c - Dv)) = f;
            }
            v[0][e] = f.x; v[1][e] = f.y; v[2][e] = f.z; v[3][e] = f.w;
        }
        uint4* xt = (c < Dv) ? xtv : xta;
        const int DT = ((c < Dv) ? Dv : Da) / 32;
        const int cb = (c < Dv) ? c : c - Dv;
        if (xt == nullptr) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = fp_bf16_rne(v[q][e]);
                l[e] = fp_bf16_rne(v[q][e] - __uint_as_float(h[e] << 16));
            }
            const int d = cb + q, dt = d >> 5, jj = d & 31;
            const int64_t base = ((((int64_t)b * NS + st) * DT + dt) * 2) * 64 + kh * 32 + jj;
            xt[base] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            xt[base + 64] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
        }
    }
}

// column partials of (sum dy, sum dy * x) over the gathered rows, for dgamma/dbeta of input_bn
__global__ __launch_bounds__(256) void frame_bn_bwd_partial_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                   const float* __restrict__ raw,
                                                                   const int32_t* __restrict__ num_frames, int B,
                                                                   int max_frames, int F, int S, float step,
                                                                   float* __restrict__ partial, const float* __restrict__ dy2,
                                                                   int64_t lddy2, int Dv) {
    // dy2 != null: the gradient arrives as two matrices, columns [0, Dv) in dy and [Dv, F) in dy2 (lpm_frame_apply_tiles_split's outputs)
    const int r0 = blockIdx.x * FP_ROWS;
    const int rows = B * S;
    const int r1 = min(rows, r0 + FP_ROWS);
    __shared__ int64_t base[FP_ROWS];
    if (threadIdx.x < FP_ROWS) {
        const int r = r0 + threadIdx.x;
        if (r < rows) {
            const int b = r / S, j = r % S;
            int idx = sample_index(j, step, num_frames[b]);
            idx = max(0, min(idx, max_frames - 1));
            base[threadIdx.x] = ((int64_t)b * max_frames + idx) * F;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += 256) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < r1 - r0; ++r) {
            const float g = (dy2 && c >= Dv) ? dy2[(int64_t)(r0 + r) * lddy2 + (c - Dv)] : dy[(int64_t)(r0 + r) * lddy + c];
            s += g;
            q = fmaf(g, raw[base[r] + c], q);
        }
        float* p = partial + (int64_t)blockIdx.x * 2 * F;
        p[c] = s;
        p[F + c] = q;
    }
}

__global__ __launch_bounds__(1024) void frame_bn_bwd_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                                   int F, const float* __restrict__ mean,
                                                                   const float* __restrict__ var, float eps,
                                                                   float* dgamma, float* dbeta) {
    double s, q;
    int c;
    partial_colsums16(partial, nblk, 2 * (int64_t)F, F, F, s, q, c);
    if (threadIdx.x < 16 && c < F) {
        // sum dy * xhat = rstd * (sum dy*x - mean * sum dy)
        const double rstd = 1.0 / sqrt((double)var[c] + (double)eps);
        dbeta[c] = (float)s;
        dgamma[c] = (float)(rstd * (q - (double)mean[c] * s));
    }
}

// The reader's output folded into the same pass (readers.py:176-193 + utils.py:28-43 + train.py:262-264): quantised uint8
// frames -> Dequantize (q * range/255 + range/512 + min) -> zero the frames at and beyond num_frames (the reader pads
// AFTER dequantising, so padding is exactly 0) -> L2-normalise each frame.  1 byte read, 4 bytes written per feature.
__global__ __launch_bounds__(256) void dequantize_l2_normalize_kernel(const unsigned char* __restrict__ q,
                                                                      const int32_t* __restrict__ num_frames, int64_t rows,
                                                                      int max_frames, int F, float scalar, float bias,
                                                                      float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int F4 = F >> 2;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int b = (int)(r / max_frames), t = (int)(r % max_frames);
        float4* dst = reinterpret_cast<float4*>(y + r * F);
        if (t >= num_frames[b]) {                       // wave-uniform
            for (int c = lane; c < F4; c += 64) dst[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const uchar4* src = reinterpret_cast<const uchar4*>(q + r * F);
        float4 v[8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < F4) {
                const uchar4 u = src[c];
                v[i] = make_float4(fmaf((float)u.x, scalar, bias), fmaf((float)u.y, scalar, bias), fmaf((float)u.z, scalar, bias),
                                   fmaf((float)u.w, scalar, bias));
            }
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        ss = wave_sum(ss);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            if (c < F4) dst[c] = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        }
    }
}

// Input normalisation of the training step (train.py:262-264, tf.nn.l2_normalize(model_input_raw, 2)): every frame row
// x <- x * rsqrt(max(sum x^2, 1e-12)).  One wave per row, float4 lanes; one read and one write of the batch.
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, int64_t rows, int F,
                                                                float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int F4 = F >> 2;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const float4* src = reinterpret_cast<const float4*>(x + r * F);
        float4 v[8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            v[i] = (c < F4) ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        ss = wave_sum(ss);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        float4* dst = reinterpret_cast<float4*>(y + r * F);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            if (c < F4) dst[c] = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        }
    }
}

// bf16 storage (BASELINE cfg-5): the sampled, batch-normalised frames leave ONLY as plain bf16 operand tiles -- frame tiles
// [b][step][column tile][lane] (K2's operand, reduction over frames) and row tiles [b][row tile][column step][lane] (K1's
// operand, reduction over features), both padded with zero frames to whole 64-frame blocks (NSP = 4 ceil(S / 64) steps, MT = NSP / 2
// row tiles) -- and, when y is given, as the fp32 matrix.  Work item = (clip, step, frame half, 8 consecutive columns).
// PL = 2 (fp32 storage, lpm_frame_apply_tiles2): the same pass writes the SPLIT-bf16 forms -- frame tiles with ceil(S / 16) steps per
// clip (the layout of lpm_split_frames) and row tiles with 2 ceil(S / 64) tiles per clip (the layout of lpm_split_rows_tiles), hi and
// lo planes -- so that K1 needs no tile-split pass of its own over the fp32 matrix.
template <int PL>
__global__ __launch_bounds__(256) void frame_apply_tiles_bf16_kernel(const float* __restrict__ raw, const int32_t* __restrict__ num_frames,
                                                                     int B, int max_frames, int F, int S, float step,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     float* __restrict__ y, uint4* __restrict__ xtv, uint4* __restrict__ xrv,
                                                                     int Dv, uint4* __restrict__ xta, uint4* __restrict__ xra, int Da) {
    const int F8 = F / 8, NSP = 4 * ((S + 63) / 64), MT = NSP / 2;
    const int64_t total = (int64_t)B * NSP * 2 * F8;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(w % F8);
        const int64_t r = w / F8;
        const int kh = (int)(r & 1), st = (int)((r >> 1) % NSP), b = (int)((r >> 1) / NSP);
        const int c = 8 * c8;
        float sc[8], sh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sc[q] = scale ? scale[c + q] : 1.f;
            sh[q] = scale ? shift[c + q] : 0.f;
        }
        const int nf = num_frames[b];
        float v[8][8];                    // [frame e][column q]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 16 * st + 8 * kh + e;
            float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
            if (j < S) {
                int idx = sample_index(j, step, nf);
                idx = max(0, min(idx, max_frames - 1));
                const float* src = raw + ((int64_t)b * max_frames + idx) * F + c;
                f0 = *reinterpret_cast<const float4*>(src);
                f1 = *reinterpret_cast<const float4*>(src + 4);
                f0.x = fmaf(f0.x, sc[0], sh[0]); f0.y = fmaf(f0.y, sc[1], sh[1]); f0.z = fmaf(f0.z, sc[2], sh[2]); f0.w = fmaf(f0.w, sc[3], sh[3]);
                f1.x = fmaf(f1.x, sc[4], sh[4]); f1.y = fmaf(f1.y, sc[5], sh[5]); f1.z = fmaf(f1.z, sc[6], sh[6]); f1.w = fmaf(f1.w, sc[7], sh[7]);
                if (y) {
                    float* dst = y + ((int64_t)b * S + j) * F + c;
                    *reinterpret_cast<float4*>(dst) = f0;
                    *reinterpret_cast<float4*>(dst + 4) = f1;
                }
            }
            v[e][0] = f0.x; v[e][1] = f0.y; v[e][2] = f0.z; v[e][3] = f0.w; v[e][4] = f1.x; v[e][5] = f1.y; v[e][6] = f1.z; v[e][7] = f1.w;
        }
        const bool video = c < Dv;
        uint4* xt = video ? xtv : xta;
        uint4* xr = video ? xrv : xra;
        if (xt == nullptr) continue;
        const int Dn = video ? Dv : Da, cb = video ? c : c - Dv;
        unsigned h[8][8], l[8][8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                h[e][q] = fp_bf16_rne(v[e][q]);
                if (PL == 2) l[e][q] = fp_bf16_rne(v[e][q] - __uint_as_float(h[e][q] << 16));
            }
        const int DT = Dn / 32, CS = Dn / 16;
        const int NSF = PL == 2 ? (S + 15) / 16 : NSP;        // frame-tile steps per clip
        if (st < NSF) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {       // frame tiles: lane = (frame half, column), 8 frames per lane
                const int d = cb + q;
                const int64_t base = ((((int64_t)b * NSF + st) * DT + (d >> 5)) * PL) * 64 + kh * 32 + (d & 31);
                xt[base] = make_uint4(h[0][q] | (h[1][q] << 16), h[2][q] | (h[3][q] << 16), h[4][q] | (h[5][q] << 16), h[6][q] | (h[7][q] << 16));
                if (PL == 2)
                    xt[base + 64] = make_uint4(l[0][q] | (l[1][q] << 16), l[2][q] | (l[3][q] << 16), l[4][q] | (l[5][q] << 16), l[6][q] | (l[7][q] << 16));
            }
        }
        if (xr == nullptr) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {       // row tiles: lane = (column half, frame), 8 columns per lane
            const int j = 16 * st + 8 * kh + e;
            const int64_t base = ((((int64_t)b * MT + (j >> 5)) * CS + (cb >> 4)) * PL) * 64 + ((cb >> 3) & 1) * 32 + (j & 31);
            xr[base] = make_uint4(h[e][0] | (h[e][1] << 16), h[e][2] | (h[e][3] << 16), h[e][4] | (h[e][5] << 16), h[e][6] | (h[e][7] << 16));
            if (PL == 2)
                xr[base + 64] = make_uint4(l[e][0] | (l[e][1] << 16), l[e][2] | (l[e][3] << 16), l[e][4] | (l[e][5] << 16), l[e][6] | (l[e][7] << 16));
        }
    }
}

}  // namespace lpm

static inline int fp_nblk(int B, int S) { return (B * S + lpm::FP_ROWS - 1) / lpm::FP_ROWS; }

extern "C" size_t lpm_frame_stats_workspace_bytes(int B, int S, int F) {
    return (size_t)fp_nblk(B, S) * 2 * F * sizeof(float);
}
extern "C" int lpm_frame_stats_nblk(int B, int S) { return fp_nblk(B, S); }

#define LPM_FRAME_CHECK(name)                                                                                       \
    LPM_REQUIRE(raw && num_frames, LPM_ERR_BADARG, name ": null pointer");                                          \
    LPM_REQUIRE(B > 0 && max_frames > 0 && F > 0 && S > 0, LPM_ERR_BADARG, name ": bad sizes");                      \
    LPM_REQUIRE(F % 4 == 0, LPM_ERR_UNSUPPORTED_SHAPE, name ": need F %% 4 == 0 (F=%d)", F)

extern "C" int lpm_frame_stats(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                               float* partial, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_stats");
    LPM_REQUIRE(partial, LPM_ERR_BADARG, "lpm_frame_stats: null workspace");
    const float step = 1.0f / (float)S;
    hipLaunchKernelGGL(frame_stats_kernel, dim3(fp_nblk(B, S)), dim3(256), 0, (hipStream_t)stream, raw, num_frames, B,
                       max_frames, F, S, step, partial);
    return check_launch("lpm_frame_stats");
}

extern "C" int lpm_frame_apply(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                               const float* scale, const float* shift, float* y, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_apply");
    LPM_REQUIRE(y && ((scale == nullptr) == (shift == nullptr)), LPM_ERR_BADARG, "lpm_frame_apply: bad pointers");
    const float step = 1.0f / (float)S;
    const int64_t total = (int64_t)B * S * (F / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(frame_apply_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, raw,
                       num_frames, B, max_frames, F, S, step, scale, shift, y);
    return check_launch("lpm_frame_apply");
}

extern "C" int lpm_frame_apply_tiles(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                                     const float* scale, const float* shift, float* y, void* xt_video, int Dv,
                                     void* xt_audio, int Da, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_apply_tiles");
    LPM_REQUIRE(y && ((scale == nullptr) == (shift == nullptr)), LPM_ERR_BADARG, "lpm_frame_apply_tiles: bad pointers");
    LPM_REQUIRE(Dv > 0 && Da >= 0 && Dv + Da == F && Dv % 32 == 0 && Da % 32 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_frame_apply_tiles: need Dv + Da == F, both multiples of 32 (F=%d Dv=%d Da=%d)", F, Dv, Da);
    const float step = 1.0f / (float)S;
    const int64_t total = (int64_t)B * ((S + 15) / 16) * 2 * (F / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(frame_apply_tiles_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, raw,
                       num_frames, B, max_frames, F, S, step, scale, shift, y, (uint4*)xt_video, Dv, (uint4*)xt_audio, Da, (float*)nullptr);
    return check_launch("lpm_frame_apply_tiles");
}
// ... with the two column blocks as two contiguous matrices y_video [B S, Dv] and y_audio [B S, Da] (round 6: NetVladV2)
extern "C" int lpm_frame_apply_tiles_split(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                                           const float* scale, const float* shift, float* y_video, float* y_audio, void* xt_video, int Dv,
                                           void* xt_audio, int Da, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_apply_tiles_split");
    LPM_REQUIRE(y_video && y_audio && ((scale == nullptr) == (shift == nullptr)), LPM_ERR_BADARG, "lpm_frame_apply_tiles_split: bad pointers");
    LPM_REQUIRE(Dv > 0 && Da > 0 && Dv + Da == F && Dv % 32 == 0 && Da % 32 == 0 && (((uintptr_t)y_video | (uintptr_t)y_audio) & 15) == 0,
                LPM_ERR_UNSUPPORTED_SHAPE, "lpm_frame_apply_tiles_split: need Dv + Da == F, both multiples of 32, aligned outputs (F=%d Dv=%d Da=%d)",
                F, Dv, Da);
    const float step = 1.0f / (float)S;
    const int64_t total = (int64_t)B * ((S + 15) / 16) * 2 * (F / 4);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(frame_apply_tiles_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, raw,
                       num_frames, B, max_frames, F, S, step, scale, shift, y_video, (uint4*)xt_video, Dv, (uint4*)xt_audio, Da, y_audio);
    return check_launch("lpm_frame_apply_tiles_split");
}

// bf16 storage: see frame_apply_tiles_bf16_kernel.  y may be NULL (then the fp32 frames are not written at all); the tile buffers
// hold lpm_frame_tiles_bf16_bytes(B, S, D) bytes each (frame tiles and row tiles have the same size).
extern "C" size_t lpm_frame_tiles_bf16_bytes(int B, int S, int D) { return (size_t)B * 4 * ((S + 63) / 64) * (D / 32) * 1024; }
extern "C" int lpm_frame_apply_tiles_bf16(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                                          const float* scale, const float* shift, float* y, void* xt_video, void* xr_video, int Dv,
                                          void* xt_audio, void* xr_audio, int Da, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_apply_tiles_bf16");
    LPM_REQUIRE(xt_video && xr_video && ((scale == nullptr) == (shift == nullptr)) && ((xt_audio == nullptr) == (xr_audio == nullptr)),
                LPM_ERR_BADARG, "lpm_frame_apply_tiles_bf16: bad pointers");
    LPM_REQUIRE(Dv > 0 && Da >= 0 && Dv + Da == F && Dv % 32 == 0 && Da % 32 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_frame_apply_tiles_bf16: need Dv + Da == F, both multiples of 32 (F=%d Dv=%d Da=%d)", F, Dv, Da);
    const float step = 1.0f / (float)S;
    const int64_t total = (int64_t)B * 4 * ((S + 63) / 64) * 2 * (F / 8);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(frame_apply_tiles_bf16_kernel<1>, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream,
                       raw, num_frames, B, max_frames, F, S, step, scale, shift, y, (uint4*)xt_video, (uint4*)xr_video, Dv,
                       (uint4*)xt_audio, (uint4*)xr_audio, Da);
    return check_launch("lpm_frame_apply_tiles_bf16");
}

// fp32 storage: a2 + a3 -> y (fp32 [B*S, F]) AND the split-bf16 frame tiles (lpm_xt_bytes each: K2's operand) AND row tiles
// (lpm_row_tiles_bytes each: K1's operand; NULL: not wanted) of both streams in one pass over the sampled frames.
extern "C" int lpm_frame_apply_tiles2(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                                      const float* scale, const float* shift, float* y, void* xt_video, void* xr_video, int Dv,
                                      void* xt_audio, void* xr_audio, int Da, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_apply_tiles2");
    LPM_REQUIRE(y && xt_video && ((scale == nullptr) == (shift == nullptr)) && (Da == 0 || xt_audio), LPM_ERR_BADARG,
                "lpm_frame_apply_tiles2: bad pointers");
    LPM_REQUIRE(Dv > 0 && Da >= 0 && Dv + Da == F && Dv % 32 == 0 && Da % 32 == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_frame_apply_tiles2: need Dv + Da == F, both multiples of 32 (F=%d Dv=%d Da=%d)", F, Dv, Da);
    const float step = 1.0f / (float)S;
    const int64_t total = (int64_t)B * 4 * ((S + 63) / 64) * 2 * (F / 8);
    const int64_t want = (total + 255) / 256;
    hipLaunchKernelGGL(frame_apply_tiles_bf16_kernel<2>, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream,
                       raw, num_frames, B, max_frames, F, S, step, scale, shift, y, (uint4*)xt_video, (uint4*)xr_video, Dv,
                       (uint4*)xt_audio, (uint4*)xr_audio, Da);
    return check_launch("lpm_frame_apply_tiles2");
}

extern "C" int lpm_frame_bn_bwd(const float* dy, int64_t lddy, const float* raw, const int32_t* num_frames, int B,
                                int max_frames, int F, int S, const float* mean, const float* var, float eps,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_bn_bwd");
    LPM_REQUIRE(dy && mean && var && dgamma && dbeta && workspace && lddy >= F, LPM_ERR_BADARG, "lpm_frame_bn_bwd: bad pointers");
    LPM_REQUIRE(workspace_bytes >= lpm_frame_stats_workspace_bytes(B, S, F), LPM_ERR_WORKSPACE, "lpm_frame_bn_bwd: workspace too small");
    const float step = 1.0f / (float)S;
    const int nblk = fp_nblk(B, S);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_bn_bwd_partial_kernel, dim3(nblk), dim3(256), 0, s, dy, lddy, raw, num_frames, B, max_frames, F, S,
                       step, (float*)workspace, (const float*)nullptr, (int64_t)0, F);
    hipLaunchKernelGGL(frame_bn_bwd_reduce_kernel, dim3((F + 15) / 16), dim3(1024), 0, s, (const float*)workspace, nblk, F, mean,
                       var, eps, dgamma, dbeta);
    return check_launch("lpm_frame_bn_bwd");
}
// ... with the gradient as two matrices: dy_video [B S, Dv] (row stride ldv) and dy_audio [B S, F - Dv] (row stride lda)
extern "C" int lpm_frame_bn_bwd_split(const float* dy_video, int64_t ldv, const float* dy_audio, int64_t lda, int Dv, const float* raw,
                                      const int32_t* num_frames, int B, int max_frames, int F, int S, const float* mean, const float* var,
                                      float eps, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, lpm_stream_t stream) {
    using namespace lpm;
    LPM_FRAME_CHECK("lpm_frame_bn_bwd_split");
    LPM_REQUIRE(dy_video && dy_audio && mean && var && dgamma && dbeta && workspace && Dv > 0 && Dv < F && ldv >= Dv && lda >= F - Dv,
                LPM_ERR_BADARG, "lpm_frame_bn_bwd_split: bad pointers / strides");
    LPM_REQUIRE(workspace_bytes >= lpm_frame_stats_workspace_bytes(B, S, F), LPM_ERR_WORKSPACE, "lpm_frame_bn_bwd_split: workspace too small");
    const float step = 1.0f / (float)S;
    const int nblk = fp_nblk(B, S);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_bn_bwd_partial_kernel, dim3(nblk), dim3(256), 0, s, dy_video, ldv, raw, num_frames, B, max_frames, F, S,
                       step, (float*)workspace, dy_audio, lda, Dv);
    hipLaunchKernelGGL(frame_bn_bwd_reduce_kernel, dim3((F + 15) / 16), dim3(1024), 0, s, (const float*)workspace, nblk, F, mean,
                       var, eps, dgamma, dbeta);
    return check_launch("lpm_frame_bn_bwd_split");
}

extern "C" int lpm_l2_normalize_rows(const float* x, int64_t rows, int F, float* y, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(x && y, LPM_ERR_BADARG, "lpm_l2_normalize_rows: null pointer");
    LPM_REQUIRE(rows > 0 && F > 0 && F % 4 == 0 && F <= 2048 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, LPM_ERR_UNSUPPORTED_SHAPE,
                "lpm_l2_normalize_rows: need F %% 4 == 0, F <= 2048, 16-byte aligned pointers (F=%d)", F);
    const int64_t want = (rows + 3) / 4;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       F, y);
    return check_launch("lpm_l2_normalize_rows");
}

extern "C" int lpm_dequantize_l2_normalize(const unsigned char* q, const int32_t* num_frames, int B, int max_frames, int F,
                                           float max_quantized_value, float min_quantized_value, float* y, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(q && num_frames && y, LPM_ERR_BADARG, "lpm_dequantize_l2_normalize: null pointer");
    LPM_REQUIRE(max_quantized_value > min_quantized_value, LPM_ERR_BADARG, "lpm_dequantize_l2_normalize: empty quantisation range");
    LPM_REQUIRE(B > 0 && max_frames > 0 && F > 0 && F % 4 == 0 && F <= 2048 && (((uintptr_t)q & 3) | ((uintptr_t)y & 15)) == 0,
                LPM_ERR_UNSUPPORTED_SHAPE, "lpm_dequantize_l2_normalize: need F %% 4 == 0, F <= 2048, aligned pointers (F=%d)", F);
    const float range = max_quantized_value - min_quantized_value;
    const float scalar = range / 255.0f, bias = range / 512.0f + min_quantized_value;
    const int64_t rows = (int64_t)B * max_frames, want = (rows + 3) / 4;
    hipLaunchKernelGGL(dequantize_l2_normalize_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, q,
                       num_frames, rows, max_frames, F, scalar, bias, y);
    return check_launch("lpm_dequantize_l2_normalize");
}
