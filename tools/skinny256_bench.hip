// Micro-benchmark of ONE launch-per-layer linear layer at the streaming shape (M = 256 rows, N = K = 1024, fp32 MFMA, packed operands),
// in a dependent chain over 24 weight matrices, hipGraph-replayed.  What bounds the 12 us such a launch takes in a streaming tick?
//   TM x TN = 16-row x 16-column tiles per workgroup (operand fragments shared in registers), K split over 8 waves (= the 8 chunks
//   of the summation order), U = k-blocks in flight per wave, LOADS = 0: no operand loads (the fixed cost of the grid)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/skinny256_bench tools/skinny256_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int M = 256, N = 1024, K = 1024, NW = 8;

template <int TM, int TN, int U, int LOADS>
__global__ __launch_bounds__(NW * 64) void layer(const float *__restrict__ x, const float *__restrict__ w,
                                                 const float *__restrict__ bias, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float red[];       // [NW][TM][TN][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    constexpr int mg = M / 16 / TM, nb = K / 16;
    const int ng = (slot / mg) * 8 + xcd, mgi = slot % mg;            // the row groups of one column group share an XCD (weights cross the fabric once)
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lo = nb * wave / NW, hi = nb * (wave + 1) / NW;
    if (LOADS) {
        for (int kb = lo; kb < hi; kb += U) {
            f32x4 xv[U][TM], wv[U][TN];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < TN; ++j) wv[u][j] = *reinterpret_cast<const f32x4 *>(w + (((size_t)(ng * TN + j) * nb + kb + u) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i) xv[u][i] = *reinterpret_cast<const f32x4 *>(x + (((size_t)(mgi * TM + i) * nb + kb + u) * 64 + lane) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(xv[u][i][e], wv[u][j][e], acc[i][j]);
        }
    }
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((wave * TM + i) * TN + j) * 256 + ((g * 4 + e) << 4) + r] = acc[i][j][e];
    __syncthreads();
    // all 512 threads fold: TM*TN*256 outputs
    for (int o = tid; o < TM * TN * 256; o += NW * 64) {
        const int t = o >> 8, idx = o & 255, i = t / TN, j = t % TN;
        float s = red[((0 * TM + i) * TN + j) * 256 + idx];
#pragma unroll
        for (int q = 1; q < NW; ++q) s += red[((q * TM + i) * TN + j) * 256 + idx];
        const int row = idx >> 4, col = idx & 15;
        const int n = (ng * TN + j) * 16 + col;
        float v = s + bias[n];
        v = v > 0.f ? v : expf(v) - 1.0f;
        const int kb = n >> 4, gg = (n & 15) >> 2, e = n & 3;
        y[(((size_t)(mgi * TM + i) * (N / 16) + kb) * 64 + gg * 16 + row) * 4 + e] = v;
    }
}

template <int TM, int TN, int U, int LOADS>
double run(const char *name, std::vector<float *> &Wp, float *bias, float *a, float *b, hipStream_t s) {
    const int L = (int)Wp.size(), REPLAY = 30;
    const int grid = (M / 16 / TM) * (N / 16 / TN);
    const size_t lds = (size_t)NW * TM * TN * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(layer<TM, TN, U, LOADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL((layer<TM, TN, U, LOADS>), dim3(grid), dim3(NW * 64), lds, s, (l & 1) ? b : a, Wp[l], bias, (l & 1) ? a : b);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < REPLAY; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::high_resolution_clock::now();
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (REPLAY * L);
        if (us < best) best = us;
    }
    printf("%-64s %4d workgroups  %6.2f us per layer\n", name, grid, best);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main() {
    const int L = 24;
    std::vector<float *> Wp(L);
    std::vector<float> h((size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f * 0.03f - 0.03f;
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&Wp[l], h.size() * 4)); CK(hipMemcpy(Wp[l], h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    float *a, *b, *bias;
    CK(hipMalloc(&a, M * K * 4)); CK(hipMalloc(&b, M * K * 4)); CK(hipMalloc(&bias, N * 4));
    CK(hipMemset(a, 0, M * K * 4)); CK(hipMemset(b, 0, M * K * 4)); CK(hipMemset(bias, 0, N * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    run<1, 1, 2, 0>("1 x 1 tile, no loads (fixed cost of 1024 workgroups)", Wp, bias, a, b, s);
    run<1, 1, 2, 1>("1 x 1 tile, 2 k-blocks in flight (the shipped shape)", Wp, bias, a, b, s);
    run<1, 1, 4, 1>("1 x 1 tile, 4 k-blocks in flight", Wp, bias, a, b, s);
    run<1, 1, 8, 1>("1 x 1 tile, 8 k-blocks in flight", Wp, bias, a, b, s);
    run<2, 1, 4, 1>("2 x 1 tiles, 4 in flight", Wp, bias, a, b, s);
    run<4, 1, 4, 1>("4 x 1 tiles, 4 in flight", Wp, bias, a, b, s);
    run<2, 2, 2, 0>("2 x 2 tiles, no loads", Wp, bias, a, b, s);
    run<2, 2, 2, 1>("2 x 2 tiles, 2 in flight", Wp, bias, a, b, s);
    run<2, 2, 4, 1>("2 x 2 tiles, 4 in flight", Wp, bias, a, b, s);
    run<2, 2, 8, 1>("2 x 2 tiles, 8 in flight", Wp, bias, a, b, s);
    run<4, 2, 4, 1>("4 x 2 tiles, 4 in flight", Wp, bias, a, b, s);
    run<2, 4, 4, 1>("2 x 4 tiles, 4 in flight", Wp, bias, a, b, s);
    run<4, 4, 2, 1>("4 x 4 tiles, 2 in flight", Wp, bias, a, b, s);
    return 0;
}
