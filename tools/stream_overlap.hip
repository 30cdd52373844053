// Micro-benchmark: do graph-replayed chains of small dependent kernels on G streams overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// ~ one recurrent layer for a 16-row group: 64 workgroups x 512 threads streaming 64 KB each
__global__ __launch_bounds__(512) void layer_kernel(const float4 *__restrict__ w, const float *__restrict__ in,
                                                    float *__restrict__ out, int iters) {
    const int tid = threadIdx.x;
    const float4 *p = w + (size_t)blockIdx.x * 512 * iters + tid;
    float acc = in[(blockIdx.x * 16 + (tid & 15)) & 1023];
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[(size_t)(i % iters) * 512];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    __shared__ float red[512];
    red[tid] = acc;
    __syncthreads();
    if (tid < 16) {
        float s = 0;
        for (int k = tid; k < 512; k += 16) s += red[k];
        out[(blockIdx.x * 16 + tid) & 1023] = s * 1e-6f;
    }
}

int main() {
    const int G = 8, NODES = 60, REPLAYS = 200, WGS = 64;
    float4 *w; CK(hipMalloc(&w, (size_t)256 * 512 * 8 * sizeof(float4) * 4));
    CK(hipMemset(w, 0, (size_t)256 * 512 * 8 * sizeof(float4) * 4));
    std::vector<hipStream_t> st(G);
    std::vector<float *> a(G), b(G);
    std::vector<hipGraphExec_t> ge(G);
    for (int g = 0; g < G; ++g) {
        CK(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
        CK(hipMalloc(&a[g], 4096)); CK(hipMalloc(&b[g], 4096));
        CK(hipMemset(a[g], 0, 4096)); CK(hipMemset(b[g], 0, 4096));
        hipGraph_t gr;
        CK(hipStreamBeginCapture(st[g], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NODES; ++i)
            hipLaunchKernelGGL(layer_kernel, dim3(WGS), dim3(512), 0, st[g], w + (size_t)(i % 16) * 64 * 512 * 8,
                               (i & 1) ? a[g] : b[g], (i & 1) ? b[g] : a[g], 8);
        CK(hipStreamEndCapture(st[g], &gr));
        CK(hipGraphInstantiate(&ge[g], gr, nullptr, nullptr, 0));
    }
    for (int ng = 1; ng <= G; ng *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int r = 0; r < REPLAYS; ++r)
                for (int g = 0; g < ng; ++g) CK(hipGraphLaunch(ge[g], st[g]));
            auto t1 = std::chrono::high_resolution_clock::now();
            CK(hipDeviceSynchronize());
            auto t2 = std::chrono::high_resolution_clock::now();
            double host = std::chrono::duration<double, std::micro>(t1 - t0).count();
            double tot = std::chrono::duration<double, std::micro>(t2 - t0).count();
            if (rep)
                printf("%d stream(s): %.2f us per node per chain, aggregate %.2f nodes/us (host enqueue %.1f%% of wall)\n",
                       ng, tot / (REPLAYS * NODES), ng * REPLAYS * NODES / tot, 100 * host / tot);
        }
    }
    // single chain but 256 workgroups per node (the un-split batch)
    {
        hipGraph_t gr; hipGraphExec_t gx;
        CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NODES; ++i)
            hipLaunchKernelGGL(layer_kernel, dim3(256), dim3(512), 0, st[0], w + (size_t)(i % 4) * 256 * 512 * 8,
                               (i & 1) ? a[0] : b[0], (i & 1) ? b[0] : a[0], 8);
        CK(hipStreamEndCapture(st[0], &gr));
        CK(hipGraphInstantiate(&gx, gr, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(gx, st[0]));
            CK(hipDeviceSynchronize());
            auto t2 = std::chrono::high_resolution_clock::now();
            double tot = std::chrono::duration<double, std::micro>(t2 - t0).count();
            if (rep) printf("1 stream, 256 WGs per node: %.2f us per node\n", tot / (REPLAYS * NODES));
        }
    }
    return 0;
}
