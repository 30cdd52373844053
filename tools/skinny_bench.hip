// Micro-benchmark of the recurrent-layer kernel shape (M=64 rows, N=1024, K=1024, fp32 MFMA) in a
// dependent chain over 24 distinct weight matrices (96 MB, like one BVRNN step), hipGraph-replayed.
// Variants explore where the per-layer time goes (operand layout, waves per workgroup, fixed cost).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int M = 64, N = 1024, K = 1024;

// MODE 0: natural layouts (x[M][K], w[N][K]);  1: w packed [ntile][kb][lane][4];  2: both packed;
//      3: no loads at all (fixed cost);  4: natural, bias loaded up front
template <int NW, int U, int MODE>
__global__ __launch_bounds__(NW * 64) void layer(const float *__restrict__ x, const float *__restrict__ w,
                                                 const float *__restrict__ bias, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int m_tiles = M / 16;
    const int ntile = (slot / m_tiles) * 8 + xcd, mtile = slot % m_tiles;
    const int m0 = mtile * 16, n0 = ntile * 16, r = lane & 15, g = lane >> 4;
    float b_early = 0.f;
    if (MODE == 4 && tid < 256) b_early = bias[n0 + (tid & 15)];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nb = K / 16, lo = nb * wave / NW, hi = nb * (wave + 1) / NW;
    if (MODE != 3) {
        for (int kb = lo; kb < hi; kb += U) {
            f32x4 xv[U], wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 2) xv[u] = *reinterpret_cast<const f32x4 *>(x + (((size_t)mtile * nb + kb + u) * 64 + lane) * 4);
                else           xv[u] = *reinterpret_cast<const f32x4 *>(x + (size_t)(m0 + r) * K + (kb + u) * 16 + g * 4);
                if (MODE == 1 || MODE == 2) wv[u] = *reinterpret_cast<const f32x4 *>(w + (((size_t)ntile * nb + kb + u) * 64 + lane) * 4);
                else           wv[u] = *reinterpret_cast<const f32x4 *>(w + (size_t)(n0 + r) * K + (kb + u) * 16 + g * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma16(xv[u][e], wv[u][e], acc);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave * 256 + ((g * 4 + e) << 4) + r] = acc[e];
    __syncthreads();
    if (tid >= 256) return;
    float s = red[tid];
#pragma unroll
    for (int q = 1; q < NW; ++q) s += red[q * 256 + tid];
    const int i = tid >> 4, j = tid & 15;
    float o = s + (MODE == 4 ? b_early : bias[n0 + j]);
    o = o > 0.f ? o : expf(o) - 1.0f;
    if (MODE == 2) {      // write the output in the packed A-operand layout of the next layer
        const int n = n0 + j, kb = n >> 4, gg = (n & 15) >> 2, e = n & 3;
        y[(((size_t)mtile * (N / 16) + kb) * 64 + gg * 16 + i) * 4 + e] = o;
    } else {
        y[(size_t)(m0 + i) * N + n0 + j] = o;
    }
}

template <int NW, int U, int MODE>
double run(const char *name, std::vector<float *> &W, std::vector<float *> &Wp, float *bias, float *a, float *b,
           hipStream_t s) {
    const int L = (int)W.size(), REPLAY = 30;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; ++l) {
        const float *w = (MODE == 1 || MODE == 2) ? Wp[l] : W[l];
        hipLaunchKernelGGL((layer<NW, U, MODE>), dim3(256), dim3(NW * 64), NW * 1024, s, (l & 1) ? b : a, w, bias,
                           (l & 1) ? a : b);
    }
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
    printf("%-44s %6.2f us per layer\n", name, best);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main() {
    const int L = 24;
    std::vector<float *> W(L), Wp(L);
    std::vector<float> h((size_t)N * K), hp((size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f * 0.03f - 0.03f;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int kb = 0; kb < K / 16; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e)
                    hp[(((size_t)nt * (K / 16) + kb) * 64 + l) * 4 + e] = h[(size_t)(nt * 16 + (l & 15)) * K + kb * 16 + (l >> 4) * 4 + e];
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&W[l], h.size() * 4)); CK(hipMalloc(&Wp[l], h.size() * 4));
        CK(hipMemcpy(W[l], h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Wp[l], hp.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    float *a, *b, *bias;
    CK(hipMalloc(&a, M * K * 4)); CK(hipMalloc(&b, M * K * 4)); CK(hipMalloc(&bias, N * 4));
    CK(hipMemset(a, 0, M * K * 4)); CK(hipMemset(b, 0, M * K * 4)); CK(hipMemset(bias, 0, N * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    run<8, 8, 3>("no loads (fixed cost), 8 waves", W, Wp, bias, a, b, s);
    run<8, 8, 0>("natural layouts, 8 waves x 8 blocks", W, Wp, bias, a, b, s);
    run<8, 8, 4>("natural, bias loaded up front", W, Wp, bias, a, b, s);
    run<16, 4, 0>("natural, 16 waves x 4 blocks", W, Wp, bias, a, b, s);
    run<4, 16, 0>("natural, 4 waves x 16 blocks", W, Wp, bias, a, b, s);
    run<8, 4, 0>("natural, 8 waves, 2 chunks of 4", W, Wp, bias, a, b, s);
    run<8, 8, 1>("W packed (1 KiB per load instr)", W, Wp, bias, a, b, s);
    run<8, 8, 2>("W and X packed", W, Wp, bias, a, b, s);
    run<16, 4, 2>("W and X packed, 16 waves", W, Wp, bias, a, b, s);
    run<4, 16, 2>("W and X packed, 4 waves", W, Wp, bias, a, b, s);
    return 0;
}
