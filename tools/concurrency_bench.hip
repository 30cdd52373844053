// Do independent chains of recurrent layers (M=64 rows, 1024x1024, fp32 MFMA, packed operands) overlap on the chip?
// C chains (one stream + one hipGraph each, 24 distinct weight matrices per chain = 96 MB) run concurrently;
// MT = 16-row tiles per workgroup (1: 256 workgroups per layer, 2: 128, 4: 64 - each weight fragment shared by MT tiles).
// Prints the aggregate layer rate.  Build: hipcc -O3 --offload-arch=gfx950 tools/concurrency_bench.hip -o tools/concurrency_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int M = 64, N = 1024, K = 1024;

template <int NW, int U, int MT, int NT = 0>
__global__ __launch_bounds__(NW * 64) void layer(const float *__restrict__ x, const float *__restrict__ w,
                                                 const float *__restrict__ bias, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][MT][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    constexpr int m_groups = M / 16 / MT;
    const int ntile = (slot / m_groups) * 8 + xcd, mg = slot % m_groups;
    const int n0 = ntile * 16, g = lane >> 4, r = lane & 15;
    f32x4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nb = K / 16, lo = nb * wave / NW, hi = nb * (wave + 1) / NW;
    for (int kb = lo; kb < ((NT & 4) ? lo : hi); kb += U) {      // NT & 4: no operand loads at all (fixed cost of a layer)
        f32x4 xv[U][MT], wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 *wp_ = reinterpret_cast<const f32x4 *>(w + (((size_t)ntile * nb + kb + u) * 64 + lane) * 4);
            wv[u] = (NT & 1) ? __builtin_nontemporal_load(wp_) : *wp_;
#pragma unroll
            for (int j = 0; j < MT; ++j)
            {
                const f32x4 *xp_ = reinterpret_cast<const f32x4 *>(x + (((size_t)(mg * MT + j) * nb + kb + u) * 64 + lane) * 4);
                xv[u][j] = (NT & 2) ? __builtin_nontemporal_load(xp_) : *xp_;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[j] = mfma16(xv[u][j][e], wv[u][e], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * MT + j) * 256 + ((g * 4 + e) << 4) + r] = acc[j][e];
    __syncthreads();
    if (tid >= 256) return;
    const int i = tid >> 4, jj = tid & 15;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        float s = red[j * 256 + tid];
#pragma unroll
        for (int q = 1; q < NW; ++q) s += red[(q * MT + j) * 256 + tid];
        float o = s + bias[n0 + jj];
        o = o > 0.f ? o : expf(o) - 1.0f;
        const int n = n0 + jj, kb = n >> 4, gg = (n & 15) >> 2, e = n & 3;
        y[(((size_t)(mg * MT + j) * (N / 16) + kb) * 64 + gg * 16 + i) * 4 + e] = o;
    }
}

struct Chain { std::vector<float *> Wp; float *a, *b; hipStream_t s; hipGraph_t g; hipGraphExec_t ge; };

template <int NW, int U, int MT, int NT = 0>
void run(const char *name, std::vector<Chain> &ch, float *bias, int C) {
    const int L = (int)ch[0].Wp.size(), REPLAY = 20;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(layer<NW, U, MT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, NW * MT * 1024));
    for (int c = 0; c < C; ++c) {
        CK(hipStreamBeginCapture(ch[c].s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL((layer<NW, U, MT, NT>), dim3(256 / MT), dim3(NW * 64), NW * MT * 1024, ch[c].s, (l & 1) ? ch[c].b : ch[c].a,
                               ch[c].Wp[l], bias, (l & 1) ? ch[c].a : ch[c].b);
        CK(hipStreamEndCapture(ch[c].s, &ch[c].g));
        CK(hipGraphInstantiate(&ch[c].ge, ch[c].g, nullptr, nullptr, 0));
    }
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < REPLAY; ++i)
            for (int c = 0; c < C; ++c) CK(hipGraphLaunch(ch[c].ge, ch[c].s));
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::high_resolution_clock::now();
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (us < best) best = us;
    }
    const double layers = (double)REPLAY * L * C;
    printf("%-28s chains %d: %6.2f us per layer per chain, aggregate %5.3f layers/us (%5.1f TFLOP/s)\n", name, C,
           best / (REPLAY * L), layers / best, layers * 2.0 * M * N * K / best * 1e-6);
    for (int c = 0; c < C; ++c) { CK(hipGraphExecDestroy(ch[c].ge)); CK(hipGraphDestroy(ch[c].g)); }
}

int main() {
    const int L = 24, CMAX = 4;
    std::vector<float> hp((size_t)N * K);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (float)((i * 2654435761u) % 2001) / 1000.0f * 0.03f - 0.03f;
    std::vector<Chain> ch(CMAX);
    float *bias;
    CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
    for (int c = 0; c < CMAX; ++c) {
        ch[c].Wp.resize(L);
        for (int l = 0; l < L; ++l) {
            CK(hipMalloc(&ch[c].Wp[l], hp.size() * 4));
            CK(hipMemcpy(ch[c].Wp[l], hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        }
        if (getenv("CB_SAMEW"))                 // every layer of every chain reads ONE 4 MB matrix: it stays in the L2s
            for (int l = 0; l < L; ++l) ch[c].Wp[l] = ch[0].Wp[0];
        CK(hipMalloc(&ch[c].a, M * K * 4)); CK(hipMalloc(&ch[c].b, M * K * 4));
        CK(hipMemset(ch[c].a, 0, M * K * 4)); CK(hipMemset(ch[c].b, 0, M * K * 4));
        CK(hipStreamCreateWithFlags(&ch[c].s, hipStreamNonBlocking));
    }
    for (int C = 1; C <= CMAX; ++C) {
        run<8, 4, 1>("MT1 (256 WG x 8 waves)", ch, bias, C);
        if (getenv("CB_VARIANTS")) {
            run<8, 4, 1, 4>("MT1, no operand loads", ch, bias, C);
            run<8, 4, 1, 1>("MT1, W loads nt", ch, bias, C);
            run<8, 4, 1, 3>("MT1, W and X loads nt", ch, bias, C);
            run<8, 2, 1>("MT1, 8 waves x U2", ch, bias, C);
            run<8, 1, 1>("MT1, 8 waves x U1", ch, bias, C);
            run<16, 1, 1>("MT1, 16 waves x U1", ch, bias, C);
            run<16, 2, 1>("MT1, 16 waves x U2", ch, bias, C);
            run<8, 8, 1>("MT1, 8 waves x U8", ch, bias, C);
            run<16, 4, 1>("MT1, 16 waves x U4", ch, bias, C);
            run<4, 8, 1>("MT1, 4 waves x U8", ch, bias, C);
            continue;
        }
        run<8, 4, 2>("MT2 (128 WG x 8 waves)", ch, bias, C);
        run<16, 4, 2>("MT2 (128 WG x 16 waves)", ch, bias, C);
        run<16, 4, 4>("MT4 (64 WG x 16 waves)", ch, bias, C);
    }
    return 0;
}
