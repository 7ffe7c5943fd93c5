// The keep mask of tf.layers.dropout (transformer_utils.py:450: rate = 1 - attention_dropout, i.e. 90 % of the output transform's
// activations are dropped while training) as one HBM-bound launch: byte i = 1 with probability keep_prob, drawn from a counter-based
// hash of (seed, i) -- 16 random bits per element, a 32-bit word for two.  torch's bernoulli_ on a uint8 tensor runs one Philox stream per
// element group: 29 us for the 24.6 M frame activations of cfg-3 (0.9 B/ns); this is a store stream (24.6 MB: ~8 us).
// Not the reference's random stream (TF's, which nothing here could reproduce anyway): parity tests and bench.py's parity leg hand the
// masks in; this kernel only serves training runs that draw their own.  Same seed, same mask: the trainer's seeds come from torch's CPU
// generator, so torch.manual_seed makes runs repeatable.
#include "lpm_common.h"

namespace lpm {

__device__ __forceinline__ unsigned dm_mix(unsigned x) {          // lowbias32 (full avalanche in two multiply-xorshift rounds)
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void dropout_keep_mask_kernel(uint4* __restrict__ out, int64_t n16, unsigned s0, unsigned s1, unsigned thresh) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned bytes = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                         // one hash word = two elements
                const unsigned long long c = (unsigned long long)i * 8ull + (unsigned)(q * 2 + h);
                const unsigned r = dm_mix((unsigned)c ^ dm_mix((unsigned)(c >> 32) ^ s1) ^ s0);
                bytes |= ((r & 0xffffu) < thresh ? 1u : 0u) << (16 * h);
                bytes |= ((r >> 16) < thresh ? 1u : 0u) << (16 * h + 8);
            }
            w[q] = bytes;
        }
        out[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace lpm

extern "C" int lpm_dropout_keep_mask(void* mask, int64_t n, float keep_prob, uint64_t seed, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(mask && n > 0 && n % 16 == 0 && ((uintptr_t)mask & 15) == 0, LPM_ERR_BADARG,
                "lpm_dropout_keep_mask: needs a 16-byte aligned mask of a multiple of 16 bytes (n=%lld)", (long long)n);
    LPM_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, LPM_ERR_BADARG, "lpm_dropout_keep_mask: keep_prob must be in (0, 1]");
    const double t = (double)keep_prob * 65536.0;
    const unsigned thresh = t >= 65536.0 ? 65536u : (unsigned)(t + 0.5);
    const int64_t n16 = n / 16;
    const int64_t want = (n16 + 255) / 256;
    hipLaunchKernelGGL(dropout_keep_mask_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, (uint4*)mask, n16,
                       (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), thresh);
    return check_launch("lpm_dropout_keep_mask");
}
