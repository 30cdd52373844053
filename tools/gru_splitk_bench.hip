// Does a GRU-shaped launch (M=64 rows, N=1024 outputs x 3 gates, K=3072, fp32 MFMA, packed operands) get faster when
// every workgroup moves fewer bytes through its L1?  Variant A: the product's shape (16x16x3-gate tile per workgroup,
// 256 workgroups x 16 waves, full K).  Variant B: MT row tiles per workgroup share each weight fragment and K is split
// over S workgroups (still 256 workgroups); the LAST workgroup of a group to arrive sums the S partial tiles in index
// order (deterministic, no spinning) and runs the epilogue.  1-3 independent chains as in concurrency_bench.hip.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gru_splitk_bench.hip -o tools/gru_splitk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int M = 64, N = 1024, K = 3072, NB = K / 16, NG = 3;

// x: packed [M/16][NB][64][4]; w: packed [gate][N/16][NB][64][4]; y: packed [M/16][N/16][64][4] (first 1024 of the next x)
template <int NW, int U, int MT, int S, int IL = 0>
__global__ __launch_bounds__(NW * 64) void gru_like(const float *__restrict__ x, const float *__restrict__ w,
                                                    float *__restrict__ y, float *__restrict__ part, unsigned *__restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][NG][MT][256]
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    constexpr int m_groups = M / 16 / MT;
    const int ks = slot % S, gslot = slot / S;                      // the S parts of a group are neighbours on one XCD
    const int ntile = (gslot / m_groups) * 8 + xcd, mg = gslot % m_groups;
    const int group = ntile * m_groups + mg;
    f32x4 acc[MT][NG];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int q = 0; q < NG; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int klo = NB * ks / S, khi = NB * (ks + 1) / S, nbw = khi - klo;
    const int lo = klo + nbw * wave / NW, hi = klo + nbw * (wave + 1) / NW;
    for (int kb = lo; kb < hi; kb += U) {
        f32x4 xv[U][MT], wv[U][NG];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = kb + u < hi ? kb + u : hi - 1;
#pragma unroll
            for (int q = 0; q < NG; ++q)
                wv[u][q] = (IL & 1) ? *reinterpret_cast<const f32x4 *>(w + ((((size_t)ntile * NB + k) * NG + q) * 64 + lane) * 4)      // gates interleaved: 3 KB per k-block
                              : *reinterpret_cast<const f32x4 *>(w + ((((size_t)q * (N / 16) + ntile) * NB + k) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < MT; ++j)
                xv[u][j] = *reinterpret_cast<const f32x4 *>(x + (((size_t)(mg * MT + j) * NB + k) * 64 + lane) * 4);
        }
        if (IL & 4) {       // finer grain: X and gate 0 first, then one gate at a time (loads of gate q+1 behind the MFMAs of gate q)
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < MT; ++j) acc[j][q] = mfma16(xv[0][j][e], wv[0][q][e], acc[j][q]);
            }
            continue;
        }
        if (!(IL & 2)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (kb + u < hi) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
#pragma unroll
                        for (int q = 0; q < NG; ++q) acc[j][q] = mfma16(xv[u][j][e], wv[u][q][e], acc[j][q]);
            }
    }
    const int g = lane >> 4, r = lane & 15;
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((wave * NG + q) * MT + j) * 256 + ((g * 4 + e) << 4) + r] = acc[j][q][e];
    __syncthreads();
    if (tid >= 256) return;
    float v[MT][NG];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            float s = red[(q * MT + j) * 256 + tid];
#pragma unroll
            for (int wv_ = 1; wv_ < NW; ++wv_) s += red[((wv_ * NG + q) * MT + j) * 256 + tid];
            v[j][q] = s;
        }
    if (S > 1) {
        float *mine = part + ((size_t)group * S + ks) * (MT * NG * 256);
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int q = 0; q < NG; ++q) mine[(j * NG + q) * 256 + tid] = v[j][q];
        __threadfence();                                            // release: the partial tile before the arrival count
        __syncthreads();                                            // (tid < 256 only: 4 waves; all must have written)
        if (tid == 0) {
            const unsigned old = atomicAdd(&cnt[group], 1u);
            last_flag = (old == (unsigned)(S - 1));
            if (last_flag) cnt[group] = 0;                          // self-cleaning for the next launch
        }
        __syncthreads();
        if (!last_flag) return;
        __threadfence();                                            // acquire
        const float *all = part + (size_t)group * S * (MT * NG * 256);
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                float s = 0.0f;
                for (int p = 0; p < S; ++p) s += __builtin_nontemporal_load(all + ((size_t)p * MT * NG + j * NG + q) * 256 + tid);
                v[j][q] = s;
            }
    }
    const int i = tid >> 4, jj = tid & 15, n = ntile * 16 + jj;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        float o = v[j][0] + v[j][1] * 0.5f + v[j][2] * 0.25f;
        o = o > 0.f ? o : expf(o) - 1.0f;
        const int kbk = n >> 4, gg = (n & 15) >> 2, e = n & 3;
        y[(((size_t)(mg * MT + j) * NB + kbk) * 64 + gg * 16 + i) * 4 + e] = o;      // first third of the next layer's x
    }
}

struct Chain { std::vector<float *> W; float *a, *b, *part; unsigned *cnt; hipStream_t s; hipGraph_t g; hipGraphExec_t ge; };

template <int NW, int U, int MT, int S, int IL = 0>
void run(const char *name, std::vector<Chain> &ch, int C) {
    const int L = (int)ch[0].W.size(), REPLAY = 20;
    const int lds = NW * NG * MT * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_like<NW, U, MT, S, IL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int c = 0; c < C; ++c) {
        CK(hipStreamBeginCapture(ch[c].s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL((gru_like<NW, U, MT, S, IL>), dim3(256 / MT * S), dim3(NW * 64), lds, ch[c].s, (l & 1) ? ch[c].b : ch[c].a,
                               ch[c].W[l], (l & 1) ? ch[c].a : ch[c].b, ch[c].part, ch[c].cnt);
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
    printf("%-44s chains %d: %6.2f us per launch per chain, aggregate %5.3f launches/us (%5.1f TFLOP/s)\n", name, C,
           best / (REPLAY * L), layers / best, layers * 2.0 * M * N * NG * K / best * 1e-6);
    for (int c = 0; c < C; ++c) { CK(hipGraphExecDestroy(ch[c].ge)); CK(hipGraphDestroy(ch[c].g)); }
}

int main() {
    const int L = 6, CMAX = 3;                       // 6 x 37.7 MB of weights per chain
    std::vector<float> hw((size_t)NG * N * K);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 2001) / 1000.0f * 0.01f - 0.01f;
    std::vector<Chain> ch(CMAX);
    for (int c = 0; c < CMAX; ++c) {
        ch[c].W.resize(L);
        for (int l = 0; l < L; ++l) { CK(hipMalloc(&ch[c].W[l], hw.size() * 4)); CK(hipMemcpy(ch[c].W[l], hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); }
        CK(hipMalloc(&ch[c].a, (size_t)M * K * 4)); CK(hipMalloc(&ch[c].b, (size_t)M * K * 4));
        CK(hipMemset(ch[c].a, 0, (size_t)M * K * 4)); CK(hipMemset(ch[c].b, 0, (size_t)M * K * 4));
        CK(hipMalloc(&ch[c].part, (size_t)256 * 4 * NG * 256 * 4 * 4)); CK(hipMalloc(&ch[c].cnt, 1024 * 4)); CK(hipMemset(ch[c].cnt, 0, 1024 * 4));
        CK(hipStreamCreateWithFlags(&ch[c].s, hipStreamNonBlocking));
    }
    for (int C = 1; C <= CMAX; ++C) {
        run<16, 2, 1, 1>("A: 16x16 tile, 256 WG x 16 waves, full K", ch, C);
        run<16, 2, 1, 1, 1>("A with the 3 gates interleaved per k-block", ch, C);
        run<16, 1, 1, 1, 1>("A interleaved, U=1", ch, C);
        run<12, 1, 1, 1, 1>("A interleaved, U=1, 12 waves", ch, C);
        run<12, 1, 1, 1, 5>("A interleaved, U=1, 12 waves, MFMAs per gate as loads land", ch, C);
        run<16, 1, 1, 1, 5>("A interleaved, U=1, 16 waves, MFMAs per gate as loads land", ch, C);
        run<16, 1, 1, 1, 0>("A NOT interleaved, U=1", ch, C);
        run<16, 1, 1, 1, 3>("A interleaved, U=1, no sched barrier", ch, C);
        run<16, 2, 1, 1, 3>("A interleaved, U=2, no sched barrier", ch, C);
        run<8, 2, 1, 1, 1>("A interleaved, 8 waves U=2", ch, C);
        run<8, 1, 1, 1, 1>("A interleaved, 8 waves U=1", ch, C);
        run<12, 2, 1, 1, 1>("A interleaved, 12 waves U=2", ch, C);
        run<8, 3, 2, 2>("B: 32x16 tile, split-K 2, 256 WG x 8 waves", ch, C);
        run<8, 2, 4, 4>("B: 64x16 tile, split-K 4, 256 WG x 8 waves", ch, C);
        run<8, 3, 2, 1>("   32x16 tile, full K, 128 WG x 8 waves", ch, C);
    }
    return 0;
}
