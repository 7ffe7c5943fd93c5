// Split-bf16 tile GEMM on the bf16 matrix pipe: C[rows, cols] (+)= A . B with both operands stored as MFMA-fragment
// tiles in HBM.  A fragment tile is 32 outer indices (rows of A / columns of B) x 16 reduction elements, hi and lo bf16
// planes, 16 bytes per lane and lane-linear:
//     tile[plane][lane][e] = M[outer = 32 * tile_index + (lane & 31)][red = 16 * step + 8 * (lane >> 5) + e]
// so one 1 KB plane is exactly one wave-wide LDS-DMA (global_load_lds_dwordx4) and one conflict-free ds_read_b128.
// The (tile, step, batch) -> address map is three strides per operand, which lets the same kernel read
//   row tiles     [b][mt][cs]  (lpm_split_rows_tiles: frames x features, reduction along the feature axis),
//   weight tiles  [rs][nt]     (lpm_split_weight_tiles),
//   frame tiles   [b][s][ct]   (lpm_split_frames: reduction along the frame axis, clips folded into the step index),
//   per-clip dU tiles          (vlad_backward_tiles.hip).
// a * b = ah*bh + ah*bl + al*bh with fp32 accumulation: ~5e-6 relative error (the 1e-3 parity bar with room to spare).
//
// Workgroup = 256 threads = 64 rows x (128 * NTW) columns: wave w owns the 32*NTW-column slice w, both row tiles.
// Per reduction step the workgroup brings 2 row tiles + 4*NTW column tiles (hi, lo) into a 3-stage LDS ring by LDS-DMA
// with ONE raw s_barrier per step and hand-counted vmcnt (cdna guide 5: no __syncthreads with glds in flight, a single
// extern LDS object); 6*NTW MFMAs per wave per step.
// 128-row form (tile_gemm.hip, MW = 2; K1's forward at K = 256): 512 threads = two row groups of four waves over one flat
// row-tile sequence, 4-stage ring, the fragment reads of step s + 1 issued under the MFMAs of step s.
// Round 3 (the encoder's dense layers, lpm_dense_tiles_*): the 128-row form with a 3-stage ring and TWO workgroups per CU (no software
// pipelining: the neighbour covers barriers and epilogues), and the 256-row form (RTW = 4: four row tiles per wave, 128 x 64
// accumulators, 12 KB of fragment reads per 24 MFMAs instead of 8 KB per 12) whose STORE epilogue can write split-bf16 operand IMAGES
// with bias + ReLU / ReLU-mask + bias-gradient fused (TileGemmArgs::img).  Measured per shape against hipBLASLt in DESIGN.md section 4.
#pragma once
#include "lpm_common.h"
#include "operand_format.h"

namespace lpm {

typedef __bf16 tg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 tg_mfma(tg_u32x4 a, tg_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tg_bf16x8, a), __builtin_bit_cast(tg_bf16x8, b), c, 0, 0, 0);
}
// fp16 operands (PL == 3, the two-product form of round 5: A = (hi, lo) fp16 planes, B = one fp16 plane)
typedef _Float16 tg_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 tg_mfma_f16(tg_u32x4 a, tg_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tg_f16x8, a), __builtin_bit_cast(tg_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned tg_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// fp32 -> (hi, lo) bf16 planes, round-to-nearest-even both: v_cvt_pk_bf16_f32 does two elements per instruction (~3 VALU
// operations per element against ~16 for the integer emulation tg_rne spells out; same bits for finite values)
typedef __bf16 tg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tg_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const tg_f32x2 v = {a, b};
    const tg_bf16x2 h = __builtin_convertvector(v, tg_bf16x2);
    const tg_f32x2 hf = __builtin_convertvector(h, tg_f32x2);
    const tg_bf16x2 l = __builtin_convertvector(v - hf, tg_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void tg_split8(const float* v, uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tg_split2(v[2 * i], v[2 * i + 1], h[i], l[i]);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

constexpr int TG_NS = 3;     // LDS ring stages
enum { TG_EPI_STORE = 0, TG_EPI_SOFTMAX_BWD = 1, TG_EPI_ADAM = 2 };

struct TileGemmArgs {
    const uint4* a;            // strides below are in 16-byte units; a (tile, step) pair is 128 units (hi plane, lo plane)
    const uint4* b;
    int64_t a_tile, a_step, a_batch;
    int64_t b_tile, b_step, b_batch;
    int a_tiles, b_tiles;      // valid tile counts per batch (indices beyond are clamped; their results are masked)
    // round 5, 128- / 256-row forms: A straight from an operand IMAGE [rows][planes K] (split_gemm.hip) instead of row tiles -- an LDS-DMA
    // load takes one global address per lane, so lane (row l31, reduction half) fetches its 16 bytes of row 32 tile + l31 itself:
    // a_img_row = the image's row stride in 16-byte units (0: row tiles), a_img_lo = the lo plane's offset in the same units; the
    // launcher sets a_tile = 32 a_img_row and a_step = 2.  The row-tile copy of the tensor (a pass of its own) is not made.
    int a_img_row, a_img_lo;
    // optional second operand pair, reduced after the first into the same accumulators (C = A.B + A2.B2; splits == 1):
    const uint4* a2;
    const uint4* b2;
    int64_t a2_tile, a2_step, a2_batch;
    int64_t b2_tile, b2_step, b2_batch;
    int b2_tiles, steps2;      // (the row tiling of A2 is A's)
    int rb_per_batch;          // 64-row blocks per batch: blockIdx.x = batch * rb_per_batch + rb
    int steps_per_split;       // blockIdx.z = split: reduction steps [z * steps_per_split, ...) clipped to total_steps
    int total_steps;
    float* out;                // STORE: out[batch * out_batch + split * out_split + row * ldo + col]
    int out_bf16;              // STORE: out points at bf16 storage (same index arithmetic, in elements); no accumulate, one split
    int64_t ldo, out_batch, out_split;
    int rows_valid, cols_valid;
    int accumulate;            // STORE: out += result
    int nt_store;              // STORE: non-temporal stores (the 0.5-2 GB hidden1 weight gradient)
    int dbg;                   // measurement only (LPM_TG_DBG)
    int cols_inner;            // 64-row form: the column blocks of a row block are CONSECUTIVE workgroups of a one-dimensional grid (same
                               // XCD, same moment: the row block's A tiles come from HBM once and from L2 for the other column blocks)
                               // instead of gridDim.y slices a whole grid row apart
    // STORE variants for a result that is a GRADIENT nobody needs to see (csrc/factored_adam.hip); 64-row form only:
    float* sumsq;              // != null: nothing is stored; sumsq[workgroup] = sum of squares of the workgroup's valid outputs
    float* adam_p;             // ADAM epilogue: nothing is stored; the tile is the gradient of adam_p[row * ldo + col]: scaled by *adam_factor
    float* adam_m;             //          (the clip), then TF-Adam on adam_p / adam_m / adam_v in place (clip_adam.hip's arithmetic)
    float* adam_v;
    const float* adam_factor;
    unsigned short* adam_p16;  //          optional: the bf16 compute copy of adam_p (same shape and row stride in elements), written beside it
    float adam_lr_t, adam_b1, adam_b2, adam_eps;
    // STORE, 256-row form only: the result leaves as a split-bf16 operand IMAGE [rows][3 * cols_valid] (split_gemm.hip's format for
    // the library GEMMs) instead of fp32 -- the [M, 4F] pre-activation of FeedForwardNetwork (transformer_utils.py:701-711) and its
    // gradient never make their fp32 round trip through HBM:
    //   img_kind 1 (activation): v = relu(acc + img_bias[col])                  -> planes [hi | lo | hi]
    //   img_kind 2 (gradient):   v = acc where the forward activation was > 0 (img_mask = ITS image: the hi plane's sign) else 0
    //                            -> planes [hi | hi | lo];  img_colpart [gridDim.x * 2][cols_valid] = per-128-row-group column sums
    //                            of v (the bias gradient's partial sums)
    unsigned short* img;
    const float* img_bias;
    const unsigned short* img_mask;
    float* img_colpart;
    int img_kind;
    // round 5 (operand_format.h): the image leaves in any operand format -- img_f16 = 1: fp16 planes of v * img_scale, img_planes = 3
    // ([hi | lo | hi] / [hi | hi | lo]) or 2 ([hi | lo]); img_amax (nullable): max |v| of the launch; mask_planes: the planes per row of
    // img_mask's image (its row stride); alpha: the accumulators are multiplied by it first (1 / the scale of the data operand's tiles;
    // the launchers set 1 when it is left 0)
    int img_f16, img_planes, mask_planes;
    float img_scale, alpha;
    float* img_amax;
    float* stats;              // STORE, optional: [gridDim.x][2][cols_valid] per-workgroup column (sum, sum of squares)
    // SOFTMAX_BWD (needs gridDim.y == gridDim.z == 1): out = dlogit~ with a = softmax(logits*scale + shift) recomputed
    const float* logits;       // [batch * rows_valid + row][cols_valid]
    int logits_bf16;           // ... stored as bf16
    const float* scale;        // [cols] or null
    const float* shift;        // [cols] or null
    const float* ctil;         // [batch][cols]
    int softmax;               // 0: out = result - ctil (similarities given, NetVLAD-V2 form)
};

// Launchers (defined in tile_gemm.hip, the only translation unit that instantiates the kernel).  nbatch * rb_per_batch
// workgroup rows, ceil(cols / (128 NTW)) column blocks, `splits` reduction splits.
// planes: 2 = split-bf16 operands (hi, lo), 1 = plain bf16 operands; fp16 operands (round 5; image epilogue of the 256-row form only):
// 3 = the two-term product (A fp16 (hi, lo) tiles, B fp16 hi-plane tiles of 64 units per (tile, step)), 4 = the three-term product (both
// operands fp16 (hi, lo) tiles)
int tile_gemm_store(const TileGemmArgs& g, int nbatch, int splits, hipStream_t stream, const char* what, int ntw = 0, int planes = 2);   // ntw 0 = by column count
int tile_gemm_adam(const TileGemmArgs& g, hipStream_t stream, const char* what);      // 64 x 128 tiles, one batch, one split, split-bf16
int tile_gemm_softmax_bwd(const TileGemmArgs& g, int nbatch, hipStream_t stream, const char* what, int planes = 2);
int tile_gemm_ntw(int cols);
// 256-row form with an image epilogue (g.img, g.img_kind, ...): one batch, one split, columns a multiple of 256, row tiles a multiple of 8
int tile_gemm_image(const TileGemmArgs& g, hipStream_t stream, const char* what, int planes = 2);   // planes 3: the fp16 two-product form
int tile_gemm_image_form();
int tile_gemm_image_row_groups(int M);

// K1's forward on flat 96-row workgroups (assign_flat.hip): the per-clip row tiles gathered per lane, B fragments straight into registers.
// stats: [nblk][2][K], nblk >= ceil(B T / 96) rows (the rows past the row groups are zeroed).
bool assign_flat_ok(int B, int T, int D, int K);
bool assign_flat_plain_ok(int B, int T, int D, int K);       // planes == 1: plain bf16 tiles, bf16 logits (round 5)
// plain bf16 tiles on 160-row x 512-column workgroups (K a multiple of 512: BASELINE configs[4] in one round of the chip)
bool assign_wide_ok(int B, int T, int MT, int D, int K, int nblk);
int assign_wide_launch(const void* xr, const void* wt, int B, int T, int MT, int D, int K, void* logits_bf16, float* stats, int nblk,
                       int timing_tag, hipStream_t stream, const char* what);
int assign_flat_launch(const void* xr, const void* wt, int B, int T, int MT, int D, int K, float* logits, float* stats, int nblk,
                       int timing_tag, hipStream_t stream, const char* what, int planes = 2);

}  // namespace lpm
