// Probe: how much faster than hipBLASLt's first heuristic choice is the best of its top-N algorithms, for the bf16 x bf16 -> fp32
// GEMM shapes of the encoders (operand images of split-bf16 arithmetic: the reduction is 3x the layer's)?
//   hipcc -O2 --offload-arch=gfx950 tools/probe/hipblaslt_algos.cpp -lhipblaslt -o /tmp/hipblaslt_algos && /tmp/hipblaslt_algos
// Column-major convention: D[m, n] = op(A)[m, k] op(B)[k, n].  A row-major product C[M, N] = X[M, K] W[K, N] is m = N, n = M, A = W (no
// transpose, lda = N), B = X (no transpose, ldb = K).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int m, n, k; hipblasOperation_t ta, tb; int batch; };

int main() {
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    const size_t wsz = 128u << 20;
    void* ws; CK(hipMalloc(&ws, wsz));
    const Shape shapes[] = {
        {"out / ffn2-like fwd  C[20480,1024] = X[20480,3072] W[3072,1024]", 1024, 20480, 3072, HIPBLAS_OP_N, HIPBLAS_OP_N, 1},
        {"qkv fwd              C[20480,3072] = X[20480,3072] W[3072,3072]", 3072, 20480, 3072, HIPBLAS_OP_N, HIPBLAS_OP_N, 1},
        {"ffn2 fwd             C[20480,1024] = X[20480,12288] W[12288,1024]", 1024, 20480, 12288, HIPBLAS_OP_N, HIPBLAS_OP_N, 1},
        {"dX                   C[20480,1024] = dY[20480,3072] Wt (W stored [1024,3072])", 1024, 20480, 3072, HIPBLAS_OP_T, HIPBLAS_OP_N, 1},
        {"dX (ffn1)            C[20480,1024] = dY[20480,12288] Wt (W stored [1024,12288])", 1024, 20480, 12288, HIPBLAS_OP_T, HIPBLAS_OP_N, 1},
        {"dW split-K batch 8   C[1024,1024] = Xt[1024,7680] dY[7680,1024]  x8", 1024, 1024, 7680, HIPBLAS_OP_N, HIPBLAS_OP_T, 8},
        {"dW split-K batch 4   C[1024,4096] = Xt[1024,15360] dY[15360,4096]  x4", 4096, 1024, 15360, HIPBLAS_OP_N, HIPBLAS_OP_T, 4},
    };
    for (const Shape& s : shapes) {
        const int64_t ar = s.ta == HIPBLAS_OP_N ? s.m : s.k, ac = s.ta == HIPBLAS_OP_N ? s.k : s.m;
        const int64_t br = s.tb == HIPBLAS_OP_N ? s.k : s.n, bc = s.tb == HIPBLAS_OP_N ? s.n : s.k;
        void *A, *B, *C;
        CK(hipMalloc(&A, (size_t)ar * ac * 2 * s.batch)); CK(hipMalloc(&B, (size_t)br * bc * 2 * s.batch)); CK(hipMalloc(&C, (size_t)s.m * s.n * 4 * s.batch));
        CK(hipMemset(A, 0x3c, (size_t)ar * ac * 2 * s.batch)); CK(hipMemset(B, 0x3c, (size_t)br * bc * 2 * s.batch));
        hipblasLtMatrixLayout_t la, lb, lc;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, ar, ac, ar));
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, br, bc, br));
        CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, s.m, s.n, s.m));
        if (s.batch > 1) {
            int32_t bt = s.batch; int64_t sa = ar * ac, sb = br * bc, sc = (int64_t)s.m * s.n;
            CK(hipblasLtMatrixLayoutSetAttribute(la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bt, sizeof(bt)));
            CK(hipblasLtMatrixLayoutSetAttribute(lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bt, sizeof(bt)));
            CK(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bt, sizeof(bt)));
            CK(hipblasLtMatrixLayoutSetAttribute(la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sa, sizeof(sa)));
            CK(hipblasLtMatrixLayoutSetAttribute(lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sb, sizeof(sb)));
            CK(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sc, sizeof(sc)));
        }
        hipblasLtMatmulDesc_t d;
        CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &s.ta, sizeof(s.ta)));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &s.tb, sizeof(s.tb)));
        hipblasLtMatmulPreference_t pref;
        CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
        std::vector<hipblasLtMatmulHeuristicResult_t> res(96);
        int got = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, d, la, lb, lc, lc, pref, (int)res.size(), res.data(), &got));
        const float alpha = 1.f, beta = 0.f;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<std::pair<float, int>> times;
        for (int i = 0; i < got; ++i) {
            bool ok = true;
            for (int w = 0; w < 2 && ok; ++w)
                ok = hipblasLtMatmul(h, d, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) continue;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 10; ++r) hipblasLtMatmul(h, d, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            times.push_back({ms / 10 * 1e3f, i});
        }
        const double gf = 2.0 * s.m * s.n * (double)s.k * s.batch * 1e-9;
        float first = -1.f;
        for (auto& t : times) if (t.second == 0) first = t.first;
        std::sort(times.begin(), times.end());
        printf("%s\n  %d algorithms; heuristic's first: %.1f us (%.0f TF/s); best: #%d %.1f us (%.0f TF/s); next: ", s.name, got, first,
               gf / first * 1e3, times.empty() ? -1 : times[0].second, times.empty() ? 0.f : times[0].first, times.empty() ? 0.0 : gf / times[0].first * 1e3);
        for (size_t i = 1; i < times.size() && i < 5; ++i) printf("#%d %.1f  ", times[i].second, times[i].first);
        printf("\n");
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
