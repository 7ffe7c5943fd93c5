// Standalone: time every hipBLASLt heuristic solution for the encoder's bf16->fp32 GEMM shapes (row-major A[M,K] . B^T, B stored [N,K]).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void fill_rand(unsigned short* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    float f = ((x & 0xffffff) / 8388608.0f - 1.0f) * 1.7f;      // ~uniform(-1.7, 1.7): unit variance
    p[i] = (unsigned short)(__float_as_uint(f) >> 16);
}
#define CK(x) do { auto _s = (x); if (_s != 0) { printf("err %d at %s:%d\n", (int)_s, __FILE__, __LINE__); return 1; } } while (0)

// row-major C[M,N] = A[M,K] * B[N,K]^T  ==  column-major C^T[N,M] = B(op T)[N,K] * A^T...: in column-major terms:
//   C^T (N x M, ld N) = (B^T)^T ... we describe: matA_cm = B as K x N col-major (ld K) with op T -> N x K;  matB_cm = A as K x M col-major (ld K) op N.
int run(hipblasLtHandle_t h, int M, int N, int K, bool b_is_nk, const char* name) {
    void *A, *B, *C, *ws;
    size_t wsz = 128 << 20;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&ws, wsz));
    fill_rand<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>((unsigned short*)A, (size_t)M * K, 1u);
    fill_rand<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>((unsigned short*)B, (size_t)N * K, 2u);
    hipblasLtMatmulDesc_t desc;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t opA = b_is_nk ? HIPBLAS_OP_T : HIPBLAS_OP_N, opB = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opA, sizeof(opA)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opB, sizeof(opB)));
    hipblasLtMatrixLayout_t la, lb, lc;
    // cm "A" operand = our B: if stored [N,K] row-major == K x N col-major (ld K), op T;  if stored [K,N] row-major == N x K col-major (ld N), op N
    if (b_is_nk) CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K)); else CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, N, K, N));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));      // our A [M,K] row-major == K x M col-major
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, N, M, N));       // our C [M,N] row-major == N x M col-major
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(128);
    int n = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 128, res.data(), &n));
    float alpha = 1.f, beta = 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::pair<float,int>> t;
    for (int i = 0; i < n; ++i) {
        bool ok = true;
        for (int w = 0; w < 2 && ok; ++w) ok = hipblasLtMatmul(h, desc, &alpha, B, la, A, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0) == HIPBLAS_STATUS_SUCCESS;
        if (!ok) continue;
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipblasLtMatmul(h, desc, &alpha, B, la, A, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        t.push_back({ms / 5 * 1e3f, i});
    }
    if (t.empty()) { printf("%s: no algo\n", name); return 0; }
    float first = t[0].first;
    std::sort(t.begin(), t.end());
    double fl = 2.0 * M * N * K;
    printf("%-28s M=%d N=%d K=%d  algos %d  heuristic-first %.1f us (%.0f TF)  best %.1f us (%.0f TF, #%d)  gain %.1f%%\n", name, M, N, K, n, first,
           fl / first / 1e6, t[0].first, fl / t[0].first / 1e6, t[0].second, 100.0 * (first - t[0].first) / first);
    hipFree(A); hipFree(B); hipFree(C); hipFree(ws);
    return 0;
}
int main() {
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    const int M = 20480;
    run(h, M, 3072, 3072, true, "qkv fwd");
    run(h, M, 1024, 3072, true, "o fwd / o dx");
    run(h, M, 4096, 3072, true, "ffn1 fwd / ffn2 dx");
    run(h, M, 1024, 12288, true, "ffn2 fwd / ffn1 dx");
    run(h, M, 1024, 9216, true, "qkv dx");
    return 0;
}
