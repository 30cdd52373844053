// Do MFMA and ordinary vector instructions of DIFFERENT waves overlap on a SIMD of gfx950?
// A workgroup of 8 waves (two per SIMD: waves w and w + 4 share SIMD w % 4).  Modes:
//   0: waves 0-3 run an MFMA loop, waves 4-7 return at once            -> T_mfma
//   1: waves 4-7 run a packed-fp32 FMA loop, waves 0-3 return at once  -> T_valu
//   2: waves 0-3 MFMA loop, waves 4-7 FMA loop                         -> max(T_mfma, T_valu) if they overlap, the sum if not
//   3: all eight waves run half the MFMA loop and half the FMA loop one after the other (what the vocoder's kernels do)
//   4 / 5: waves 4-7 only: the flops of mode 1 as ordinary v_fma_f32 (16 chains) / as v_pk_fma_f32 on 16 chains
//   6 / 7: all eight waves, half the work each: ordinary / packed FMAs with two waves per SIMD
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_valu_overlap tools/mfma_valu_overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma_loop(int iters, float seed, float *sink) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    const float a = seed + 1.0f, b = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) *sink = s;
}
__device__ __forceinline__ void valu_loop(int iters, float seed, float *sink) {
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){seed + i, seed - i};
    const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.678f) *sink = s;
}

// the same flops as valu_loop with ordinary (one float per lane) FMAs: 16 independent chains
__device__ __forceinline__ void scalar_loop(int iters, float seed, float *sink) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) *sink = s;
}
// packed FMAs on 16 independent chains (more instruction-level parallelism than valu_loop's 8)
__device__ __forceinline__ void valu16_loop(int iters, float seed, float *sink) {
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = (f32x2){seed + i, seed - i};
    const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.678f) *sink = s;
}

// integer vector instructions (address arithmetic, masks): 16 independent chains
__device__ __forceinline__ void int_loop(int iters, float seed, float *sink) {
    unsigned v[16];
    for (int i = 0; i < 16; ++i) v[i] = (unsigned)(seed * 1000.f) + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (v[i] << 1) + (v[i] ^ 0x9E3779B9u);       // v_lshl_add / v_xor: two per chain and trip
    }
    unsigned s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345678u) *sink = (float)s;
}
// one wave: MFMAs with independent vector instructions between them (does a wave's own vector work hide behind its MFMAs?)
template <int KIND>
__device__ __forceinline__ void mixed_loop(int iters, float seed, float *sink) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    const float a = seed + 1.0f, b = seed * 0.5f;
    float v[16];
    f32x2 w[8];
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int i = 0; i < 8; ++i) w[i] = (f32x2){seed + i, seed - i};
    const float m = 1.0000001f, c = 1e-7f;
    const f32x2 m2 = {1.0000001f, 0.9999999f}, c2 = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (KIND == 0) { v[2 * i] = __builtin_fmaf(v[2 * i], m, c); v[2 * i + 1] = __builtin_fmaf(v[2 * i + 1], m, c); }     // two v_fma_f32 per MFMA
            else w[i] = __builtin_elementwise_fma(w[i], m2, c2);                                                                // one v_pk_fma_f32 per MFMA
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + w[i][0] + w[i][1];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) *sink = s;
}

__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, float seed, float *sink) {
    const int wave = threadIdx.x >> 6;
    if (mode == 0) { if (wave < 4) mfma_loop(n_mfma, seed, sink); }
    else if (mode == 1) { if (wave >= 4) valu_loop(n_valu, seed, sink); }
    else if (mode == 2) { if (wave < 4) mfma_loop(n_mfma, seed, sink); else valu_loop(n_valu, seed, sink); }
    else if (mode == 3) { mfma_loop(n_mfma / 2, seed, sink); valu_loop(n_valu / 2, seed, sink); }
    else if (mode == 4) { if (wave >= 4) scalar_loop(n_valu, seed, sink); }           // 16 v_fma_f32 per trip = the flops of 8 v_pk_fma_f32
    else if (mode == 5) { if (wave >= 4) valu16_loop(n_valu / 2, seed, sink); }       // 16 v_pk_fma_f32 per trip, half the trips
    else if (mode == 6) { scalar_loop(n_valu / 2, seed, sink); }                      // two waves per SIMD, ordinary FMAs
    else if (mode == 7) { valu16_loop(n_valu / 4, seed, sink); }                      // two waves per SIMD, packed FMAs
    else if (mode == 8) { if (wave < 4) mfma_loop(n_mfma, seed, sink); else scalar_loop(n_valu, seed, sink); }     // MFMA wave + ordinary-FMA wave per SIMD
    else if (mode == 9) { if (wave >= 4) int_loop(n_valu, seed, sink); }
    else if (mode == 10) { if (wave < 4) mfma_loop(n_mfma, seed, sink); else int_loop(n_valu, seed, sink); }       // MFMA wave + integer wave per SIMD
    else if (mode == 11) { if (wave < 4) mixed_loop<0>(n_mfma, seed, sink); }          // one wave per SIMD: 2 v_fma_f32 behind every MFMA
    else if (mode == 12) { if (wave < 4) mixed_loop<1>(n_mfma, seed, sink); }          // one wave per SIMD: 1 v_pk_fma_f32 behind every MFMA
    else if (mode == 13) { mixed_loop<0>(n_mfma / 2, seed, sink); }                    // two waves per SIMD, each half the trips of mode 11
    else { mixed_loop<1>(n_mfma / 2, seed, sink); }
}

int main(int argc, char **argv) {
    const int n_mfma = argc > 1 ? atoi(argv[1]) : 20000;       // x 8 MFMAs of 32 clocks
    const int n_valu = argc > 2 ? atoi(argv[2]) : 80000;       // x 8 packed FMAs
    float *sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 15; ++mode) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_valu, 0.001f, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            static const char *what[15] = {"waves 0-3: MFMA loop", "waves 4-7: v_pk_fma_f32, 8 chains", "waves 0-3 MFMA + waves 4-7 v_pk_fma_f32", "all waves: half MFMA loop, then half v_pk_fma_f32 loop",
                                          "waves 4-7: v_fma_f32, 16 chains (same flops)", "waves 4-7: v_pk_fma_f32, 16 chains", "all waves: v_fma_f32, half the trips each", "all waves: v_pk_fma_f32 (16 chains), half the trips each",
                                          "waves 0-3 MFMA + waves 4-7 v_fma_f32 (modes 0 + 4 at once)", "waves 4-7: integer (v_lshl_add, v_xor), 16 chains", "waves 0-3 MFMA + waves 4-7 integer (modes 0 + 9 at once)",
                                          "waves 0-3: every MFMA followed by 2 independent v_fma_f32", "waves 0-3: every MFMA followed by 1 independent v_pk_fma_f32",
                                          "all waves: MFMA + 2 v_fma_f32, half the trips each", "all waves: MFMA + 1 v_pk_fma_f32, half the trips each"};
            if (rep) printf("mode %d: %.3f ms  %s  (%d x 8 MFMA 16x16x4 f32, %d x 8 v_pk_fma_f32 or their flops per wave-pair; 256 workgroups x 8 waves)\n", mode, ms, what[mode], n_mfma, n_valu);
        }
    return 0;
}
