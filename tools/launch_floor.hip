// Micro-benchmark: cost of a dependent kernel boundary on this box (eager vs hipGraph), for a
// trivial kernel and for a kernel that dirties a few hundred KB (like one recurrent layer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(float *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void touch_kernel(const float *__restrict__ in, float *__restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.0001f + 1.0f;
}

int main() {
    float *a, *b;
    const int n = 64 * 1024;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (float *)nullptr);
                if (variant == 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, s, (float *)nullptr);
                if (variant == 2) hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, n);
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("eager variant %d: %.2f us per launch\n", variant, ms * 1e3 / N);
        }
        // graph of 50 launches replayed N/50 times
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 50; ++i) {
            if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (float *)nullptr);
            if (variant == 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, s, (float *)nullptr);
            if (variant == 2) hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, n);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N / 50; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("graph variant %d: %.2f us per kernel node\n", variant, ms * 1e3 / N);
        }
    }
    // two streams concurrently (eager, variant 2 on each)
    hipStream_t s2; CK(hipStreamCreate(&s2));
    float *c, *d; CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s)); CK(hipEventRecord(f0, s2));
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, n);
            hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s2, (i & 1) ? c : d, (i & 1) ? d : c, n);
        }
        CK(hipEventRecord(e1, s)); CK(hipEventRecord(f1, s2));
        CK(hipDeviceSynchronize());
        float m1, m2; CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, f0, f1));
        if (rep) printf("2 streams eager: %.2f / %.2f us per launch per stream\n", m1 * 1e3 / N, m2 * 1e3 / N);
    }
    return 0;
}
