// Micro-benchmark: a chain of recurrent-layer shaped GEMMs (M=64, N=K=1024, fp32 MFMA, ELU) run
//   (a) as one kernel per layer, hipGraph-replayed (what round 1 shipped), and
//   (b) inside ONE persistent kernel, one workgroup per 16x16 output tile, layers chained by
//       "poisoned buffer" dataflow: every activation buffer is pre-filled with a NaN sentinel,
//       producers store their 1 KiB output block write-through (sc1), consumers load their operand
//       blocks with sc1 loads and simply re-load a block while it still contains the sentinel.
//       The data is the flag: no counters, no fences, no grid barrier.  The weights of the next
//       layer are requested before the wait, so they travel while the producers finish.
// Both produce the same bits (same k order, same reduction order); the program checks that.
//   hipcc -O3 --offload-arch=gfx950 -o tools/persist_bench tools/persist_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int M = 64, N = 1024, K = 1024, NB = K / 16, NT = N / 16, MT = M / 16;
constexpr unsigned POISON = 0xFFFFDEADu;
constexpr int AUX_SC1 = 16;
constexpr int RING = 4;

struct Args {
    const float *W;            // [L] packed weight matrices [ntile][kb][lane][4], back to back  (A operand: rows = output features)
    const float *bias;         // [N]
    float *ring;               // [RING][MT][NB][64][4] activations, packed B-operand fragments
    int nlayers, L;
    unsigned *status;          // != 0: a wait timed out
    unsigned long long *stamps;// [2]: min start / max end (s_memrealtime, 100 MHz)
    unsigned spin_limit;
    unsigned long long *diag;  // DIAG: [nlayers][2 waves][6] stamps of workgroup 0 (s_memrealtime) + poll counts
};

__device__ __forceinline__ f32x4 elu4(f32x4 v) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? v[j] : expf(v[j]) - 1.0f;
    return o;
}

// Reduction + epilogue shared by both forms: partial tiles in LDS as [wave][lane] float4, fixed order.
template <int NW>
__device__ __forceinline__ f32x4 reduce_tile(const float *red, int lane, const float *bias, int ntile) {
    f32x4 s = *reinterpret_cast<const f32x4 *>(red + lane * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        const f32x4 p = *reinterpret_cast<const f32x4 *>(red + (w * 64 + lane) * 4);
        s += p;
    }
    const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + ntile * 16 + (lane >> 4) * 4);
    return elu4(s + b);
}

// ---------------------------------------------------------------- (a) one launch per layer
template <int NW, bool PBA = false>
__global__ __launch_bounds__(NW * 64) void layer_ref(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ bias, float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float red[NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int ntile = (slot / MT) * 8 + xcd, mtile = slot % MT;
    constexpr int PER = NB / NW;
    const int kb0 = wave * PER;
    f32x4 wv[PER], xv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        wv[u] = *reinterpret_cast<const f32x4 *>(w + (((size_t)ntile * NB + kb0 + u) * 64 + lane) * 4);
        xv[u] = *reinterpret_cast<const f32x4 *>(x + (((size_t)mtile * NB + kb0 + u) * 64 + lane) * 4);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!PBA) {
#pragma unroll
        for (int u = 0; u < PER; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(wv[u][e], xv[u][e], acc);     // D[n][m]: lane = (n/4)*16 + m holds n%4 = 0..3
    } else {                     // one accumulator per k-block, summed in block order (what the streaming consumer does)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            f32x4 ab = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) ab = mfma16(wv[u][e], xv[u][e], ab);
            if (u == 0) acc = ab; else acc += ab;
        }
    }
    *reinterpret_cast<f32x4 *>(red + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave != 0) return;
    const f32x4 o = reduce_tile<NW>(red, lane, bias, ntile);
    *reinterpret_cast<f32x4 *>(y + (((size_t)mtile * NT + ntile) * 64 + lane) * 4) = o;
}

// ---------------------------------------------------------------- (b) persistent, poisoned-buffer dataflow
// XMODE: how the operand blocks are fetched once their flags are up.  0: sc1 loads (device scope per access).  1: an
// agent-scope acquire (buffer_inv sc1) after the flag poll, then ordinary loads.  2: ordinary loads with no invalidate
// (timing reference only: may read stale lines).
template <int NW, bool PREFETCH_W, int POLL, bool DIAG, int XMODE = 0>
__global__ __launch_bounds__(NW * 64) void persist(Args a) {
    __shared__ __attribute__((aligned(16))) float red[2][NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int ntile = (slot / MT) * 8 + xcd, mtile = slot % MT;
    constexpr int PER = NB / NW;
    const int kb0 = wave * PER;
    unsigned long long t_start = 0;
    if (tid == 0) t_start = __builtin_amdgcn_s_memrealtime();

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, RING * M * K * 4, 0x00020000);
    const unsigned x_lane_off = (unsigned)((((size_t)mtile * NB + kb0) * 64 + lane) * 16);     // bytes inside a ring slot
    const unsigned y_lane_off = (unsigned)((((size_t)mtile * NT + ntile) * 64 + lane) * 16);
    constexpr unsigned SLOT_BYTES = M * K * 4;
    bool give_up = false;

    f32x4 wv[PER];
    if (PREFETCH_W) {
        const float *w = a.W;
#pragma unroll
        for (int u = 0; u < PER; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(w + (((size_t)ntile * NB + kb0 + u) * 64 + lane) * 4);
    }
    for (int i = 0; i < a.nlayers; ++i) {
        if (!PREFETCH_W) {
            const float *w = a.W + (size_t)(i % a.L) * N * K;
#pragma unroll
            for (int u = 0; u < PER; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(w + (((size_t)ntile * NB + kb0 + u) * 64 + lane) * 4);
        }
        unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0;
        const bool dg = DIAG && bid == 0 && (wave == 0 || wave == 3);
        if (dg) d0 = __builtin_amdgcn_s_memrealtime();
        // ---- operand blocks of the previous layer's output: load, re-load while poisoned
        const unsigned xbase = (unsigned)((i + RING - 1) % RING) * SLOT_BYTES + x_lane_off;
        unsigned spins = 0;
        f32x4 bpre = {0.f, 0.f, 0.f, 0.f};
        if (POLL == 2) {
            if (wave == 0) bpre = *reinterpret_cast<const f32x4 *>(a.bias + ntile * 16 + (lane >> 4) * 4);
            // lane j < PER watches the last dword of producer block j (4 bytes per producer per poll)
            const unsigned foff = xbase - (unsigned)lane * 16u + (unsigned)(lane < PER ? lane : PER - 1) * 1024u + 63u * 16u + 12u;
            bool bad;
            do {
                const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(rs, foff, 0, AUX_SC1);
                bad = __any(t == POISON);
                if (bad) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) { give_up = true; if (lane == 0) atomicExch(a.status, 1u + (unsigned)i); }
                }
            } while (bad && !give_up);
        }
        u32x4 xr[PER];
        if (XMODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
#pragma unroll
        for (int u = 0; u < PER; ++u) xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xbase + u * 1024, 0, XMODE == 0 ? AUX_SC1 : 0));
        unsigned pending = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const bool bad = xr[u][0] == POISON || xr[u][1] == POISON || xr[u][2] == POISON || xr[u][3] == POISON;
            if (__any(bad)) pending |= 1u << u;
        }
        if (dg) d1 = __builtin_amdgcn_s_memrealtime();
        while (POLL >= 1 && pending && !give_up) {      // every pass fetches all blocks again (8 KiB per wave per pass)
            __builtin_amdgcn_s_sleep(2);
#pragma unroll
            for (int u = 0; u < PER; ++u) xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xbase + u * 1024, 0, AUX_SC1));
            pending = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const bool b2 = xr[u][0] == POISON || xr[u][1] == POISON || xr[u][2] == POISON || xr[u][3] == POISON;
                if (__any(b2)) pending |= 1u << u;
            }
            if (++spins > a.spin_limit) { give_up = true; if (lane == 0) atomicExch(a.status, 1u + (unsigned)i); }
        }
        while (POLL == 0 && pending && !give_up) {
            // poll ONE pending block (1 KiB per wave per poll) until it is whole, then fetch everything again
            const unsigned poff = xbase + (unsigned)__builtin_ctz(pending) * 1024u;
            bool bad;
            do {
                __builtin_amdgcn_s_sleep(1);
                const u32x4 t = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff, 0, AUX_SC1));
                bad = __any(t[0] == POISON || t[1] == POISON || t[2] == POISON || t[3] == POISON);
                if (++spins > a.spin_limit) {          // bounded: record it and stop waiting for good
                    give_up = true;
                    if (lane == 0) atomicExch(a.status, 1u + (unsigned)i);
                }
            } while (bad && !give_up);
#pragma unroll
            for (int u = 0; u < PER; ++u) xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xbase + u * 1024, 0, AUX_SC1));
            pending = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const bool b2 = xr[u][0] == POISON || xr[u][1] == POISON || xr[u][2] == POISON || xr[u][3] == POISON;
                if (__any(b2)) pending |= 1u << u;
            }
        }
        if (dg) d2 = __builtin_amdgcn_s_memrealtime();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const f32x4 xv = __builtin_bit_cast(f32x4, xr[u]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(wv[u][e], xv[e], acc);
        }
        if (PREFETCH_W && i + 1 < a.nlayers) {         // next layer's weights travel during the reduction and the wait
            const float *w = a.W + (size_t)((i + 1) % a.L) * N * K;
#pragma unroll
            for (int u = 0; u < PER; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(w + (((size_t)ntile * NB + kb0 + u) * 64 + lane) * 4);
        }
        float *r = red[i & 1];
        *reinterpret_cast<f32x4 *>(r + (wave * 64 + lane) * 4) = acc;
        __syncthreads();
        if (dg) d3 = __builtin_amdgcn_s_memrealtime();
        if (wave == 0) {
            f32x4 o;
            if (POLL == 2) {
                f32x4 sum = *reinterpret_cast<const f32x4 *>(r + lane * 4);
#pragma unroll
                for (int w = 1; w < NW; ++w) sum += *reinterpret_cast<const f32x4 *>(r + (w * 64 + lane) * 4);
                o = elu4(sum + bpre);
            } else o = reduce_tile<NW>(r, lane, a.bias, ntile);
            u32x4 ob = __builtin_bit_cast(u32x4, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (ob[j] == POISON) ob[j] = 0x7FC00000u;     // never emit the sentinel as data
            // publish, then re-arm the slot that will receive layer i+2 (it holds layer i-2, whose readers are all done)
            const u32x4 pz = {POISON, POISON, POISON, POISON};
            __builtin_amdgcn_raw_buffer_store_b128(ob, rs, (unsigned)(i % RING) * SLOT_BYTES + y_lane_off, 0, AUX_SC1);
            __builtin_amdgcn_raw_buffer_store_b128(pz, rs, (unsigned)((i + 2) % RING) * SLOT_BYTES + y_lane_off, 0, AUX_SC1);
        }
        if (dg && lane == 0) {
            d4 = __builtin_amdgcn_s_memrealtime();
            unsigned long long *q = a.diag + ((size_t)i * 2 + (wave ? 1 : 0)) * 6;
            q[0] = d0; q[1] = d1; q[2] = d2; q[3] = d3; q[4] = d4; q[5] = spins;
        }
    }
    if (tid == 0) {
        atomicMin(&a.stamps[0], t_start);
        atomicMax(&a.stamps[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}


// ---------------------------------------------------------------- (c) persistent, grouped consumer
// Each wave watches one flag dword per producer block.  Its PER blocks are consumed in G groups, in order: as soon as
// the producers of a group have published, the group is fetched, verified and multiplied, so that after the LAST
// producer only one group's work is left (the k order, hence the bits, are those of the plain chain).
// FILL adds a second, independent 1024-deep product per layer (its weights requested a layer ahead, its MFMAs
// issued after the reduction barrier, i.e. inside the wait for the next layer).
template <int NW, int G, bool FILL, bool DIAG, bool WLATE = false>
__global__ __launch_bounds__(NW * 64) void persist2(Args a) {
    __shared__ __attribute__((aligned(16))) float red[2][NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int ntile = (slot / MT) * 8 + xcd, mtile = slot % MT;
    constexpr int PER = NB / NW, GS = PER / G;
    constexpr unsigned ALL = (1u << PER) - 1u;
    const int kb0 = wave * PER;
    unsigned long long t_start = 0;
    if (tid == 0) t_start = __builtin_amdgcn_s_memrealtime();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, RING * M * K * 4, 0x00020000);
    const unsigned x_lane_off = (unsigned)((((size_t)mtile * NB + kb0) * 64 + lane) * 16);
    const unsigned y_lane_off = (unsigned)((((size_t)mtile * NT + ntile) * 64 + lane) * 16);
    const unsigned f_lane_off = (unsigned)((((size_t)mtile * NB + kb0 + (lane < PER ? lane : PER - 1)) * 64 + 63) * 16 + 12);
    constexpr unsigned SLOT_BYTES = M * K * 4;
    bool give_up = false;
    const size_t w_lane = (((size_t)ntile * NB + kb0) * 64 + lane) * 4;

    f32x4 wv[PER], wf[PER];
    f32x4 facc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        wv[u] = *reinterpret_cast<const f32x4 *>(a.W + w_lane + u * 256);
        if (FILL) wf[u] = *reinterpret_cast<const f32x4 *>(a.W + (size_t)(7 % a.L) * N * K + w_lane + u * 256);
    }
    for (int i = 0; i < a.nlayers; ++i) {
        unsigned long long d0 = 0, d2 = 0, d3 = 0, d4 = 0;
        const bool dg = DIAG && bid == 0 && (wave == 0 || wave == 3);
        if (dg) d0 = __builtin_amdgcn_s_memrealtime();
        const unsigned sbase = (unsigned)((i + RING - 1) % RING) * SLOT_BYTES;
        f32x4 bpre = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0) bpre = *reinterpret_cast<const f32x4 *>(a.bias + ntile * 16 + (lane >> 4) * 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        u32x4 xr[PER];
        unsigned spins = 0, seen = 0;                      // seen: producer blocks whose flag dword has been observed
#pragma unroll
        for (int g = 0; g < G; ++g) {
            constexpr unsigned gm0 = (1u << GS) - 1u;
            const unsigned gmask = gm0 << (g * GS);
            while ((seen & gmask) != gmask && !give_up) {
                const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(rs, sbase + f_lane_off, 0, AUX_SC1);
                seen = ~(unsigned)__ballot(t == POISON) & ALL;
                if ((seen & gmask) != gmask) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) { give_up = true; if (lane == 0) atomicExch(a.status, 1u + (unsigned)i); }
                }
            }
            bool again;
            do {                                           // a flag can be visible before the rest of its block: verify
#pragma unroll
                for (int u = g * GS; u < (g + 1) * GS; ++u)
                    xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sbase + x_lane_off + u * 1024, 0, AUX_SC1));
                if (WLATE && g == 0) {                       // weights requested only now: nothing queued ahead of the polls
                    const float *w = a.W + (size_t)(i % a.L) * N * K + w_lane;
#pragma unroll
                    for (int u = 0; u < PER; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(w + u * 256);
                }
                bool bad = false;
#pragma unroll
                for (int u = g * GS; u < (g + 1) * GS; ++u)
                    bad |= xr[u][0] == POISON || xr[u][1] == POISON || xr[u][2] == POISON || xr[u][3] == POISON;
                again = __any(bad) && !give_up;
                if (again && ++spins > a.spin_limit) { give_up = true; if (lane == 0) atomicExch(a.status, 1u + (unsigned)i); }
            } while (again);
#pragma unroll
            for (int u = g * GS; u < (g + 1) * GS; ++u) {
                const f32x4 xv = __builtin_bit_cast(f32x4, xr[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma16(wv[u][e], xv[e], acc);
            }
        }
        if (dg) d2 = __builtin_amdgcn_s_memrealtime();
        float *r = red[i & 1];
        *reinterpret_cast<f32x4 *>(r + (wave * 64 + lane) * 4) = acc;
        f32x4 wfn[PER];
        if (i + 1 < a.nlayers) {                        // next layer's weights (and the next filler's) travel during the wait
            const float *w = a.W + (size_t)((i + 1) % a.L) * N * K + w_lane;
            const float *w2 = a.W + (size_t)((i + 8) % a.L) * N * K + w_lane;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (!WLATE) wv[u] = *reinterpret_cast<const f32x4 *>(w + u * 256);
                if (FILL) wfn[u] = *reinterpret_cast<const f32x4 *>(w2 + u * 256);
            }
        }
        __syncthreads();
        if (dg) d3 = __builtin_amdgcn_s_memrealtime();
        if (wave == 0) {
            f32x4 sum = *reinterpret_cast<const f32x4 *>(r + lane * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += *reinterpret_cast<const f32x4 *>(r + (w * 64 + lane) * 4);
            const f32x4 o = elu4(sum + bpre);
            u32x4 ob = __builtin_bit_cast(u32x4, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (ob[j] == POISON) ob[j] = 0x7FC00000u;
            const u32x4 pz = {POISON, POISON, POISON, POISON};
            __builtin_amdgcn_raw_buffer_store_b128(ob, rs, (unsigned)(i % RING) * SLOT_BYTES + y_lane_off, 0, AUX_SC1);
            __builtin_amdgcn_raw_buffer_store_b128(pz, rs, (unsigned)((i + 2) % RING) * SLOT_BYTES + y_lane_off, 0, AUX_SC1);
        }
        if (FILL) {                                     // independent product on the operand blocks this wave still holds
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const f32x4 xv = __builtin_bit_cast(f32x4, xr[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) facc = mfma16(wf[u][e], xv[e], facc);
            }
            if (i + 1 < a.nlayers) {
#pragma unroll
                for (int u = 0; u < PER; ++u) wf[u] = wfn[u];
            }
        }
        if (dg && lane == 0) {
            d4 = __builtin_amdgcn_s_memrealtime();
            unsigned long long *q = a.diag + ((size_t)i * 2 + (wave ? 1 : 0)) * 6;
            q[0] = d0; q[1] = d0; q[2] = d2; q[3] = d3; q[4] = d4; q[5] = spins;
        }
    }
    if (FILL && facc[0] == 1234.5f && facc[1] == 77.f) a.status[0] = 99;      // keeps the filler product alive
    if (tid == 0) {
        atomicMin(&a.stamps[0], t_start);
        atomicMax(&a.stamps[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ---------------------------------------------------------------- (d) persistent, fetch-as-poll with an adaptive start delay
// No flag round trip: after a self-tuned delay the wave fetches all its operand blocks; if any is still poisoned it
// fetches again.  The delay shrinks while first fetches succeed and grows when they fail.
template <int NW, bool DIAG>
__global__ __launch_bounds__(NW * 64) void persist3(Args a) {
    __shared__ __attribute__((aligned(16))) float red[2][NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int ntile = (slot / MT) * 8 + xcd, mtile = slot % MT;
    constexpr int PER = NB / NW;
    const int kb0 = wave * PER;
    unsigned long long t_start = 0;
    if (tid == 0) t_start = __builtin_amdgcn_s_memrealtime();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, RING * M * K * 4, 0x00020000);
    const unsigned x_lane_off = (unsigned)((((size_t)mtile * NB + kb0) * 64 + lane) * 16);
    const unsigned y_lane_off = (unsigned)((((size_t)mtile * NT + ntile) * 64 + lane) * 16);
    constexpr unsigned SLOT_BYTES = M * K * 4;
    bool give_up = false;
    const size_t w_lane = (((size_t)ntile * NB + kb0) * 64 + lane) * 4;
    f32x4 wv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(a.W + w_lane + u * 256);
    int delay = 16;                                        // in units of s_sleep 1 (64 clocks)
    unsigned total_fail = 0;
    for (int i = 0; i < a.nlayers; ++i) {
        unsigned long long d0 = 0, d2 = 0, d3 = 0, d4 = 0;
        const bool dg = DIAG && bid == 0 && (wave == 0 || wave == 3);
        if (dg) d0 = __builtin_amdgcn_s_memrealtime();
        const unsigned sbase = (unsigned)((i + RING - 1) % RING) * SLOT_BYTES;
        f32x4 bpre = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0) bpre = *reinterpret_cast<const f32x4 *>(a.bias + ntile * 16 + (lane >> 4) * 4);
        for (int q = 0; q < delay; ++q) __builtin_amdgcn_s_sleep(1);
        u32x4 xr[PER];
        unsigned fails = 0;
        bool again;
        do {
#pragma unroll
            for (int u = 0; u < PER; ++u)
                xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sbase + x_lane_off + u * 1024, 0, AUX_SC1));
            bool bad = false;
#pragma unroll
            for (int u = 0; u < PER; ++u)
                bad |= xr[u][0] == POISON || xr[u][1] == POISON || xr[u][2] == POISON || xr[u][3] == POISON;
            again = __any(bad) && !give_up;
            if (again) {
                __builtin_amdgcn_s_sleep(4);
                if (++fails > a.spin_limit) { give_up = true; if (lane == 0) atomicExch(a.status, 1u + (unsigned)i); }
            }
        } while (again);
        if (fails == 0) delay = delay > 0 ? delay - 1 : 0;
        else            delay = delay + 4 < 200 ? delay + 4 : 200;
        total_fail += fails;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const f32x4 xv = __builtin_bit_cast(f32x4, xr[u]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(wv[u][e], xv[e], acc);
        }
        if (dg) d2 = __builtin_amdgcn_s_memrealtime();
        float *r = red[i & 1];
        *reinterpret_cast<f32x4 *>(r + (wave * 64 + lane) * 4) = acc;
        if (i + 1 < a.nlayers) {
            const float *w = a.W + (size_t)((i + 1) % a.L) * N * K + w_lane;
#pragma unroll
            for (int u = 0; u < PER; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(w + u * 256);
        }
        __syncthreads();
        if (dg) d3 = __builtin_amdgcn_s_memrealtime();
        if (wave == 0) {
            f32x4 sum = *reinterpret_cast<const f32x4 *>(r + lane * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += *reinterpret_cast<const f32x4 *>(r + (w * 64 + lane) * 4);
            const f32x4 o = elu4(sum + bpre);
            u32x4 ob = __builtin_bit_cast(u32x4, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (ob[j] == POISON) ob[j] = 0x7FC00000u;
            const u32x4 pz = {POISON, POISON, POISON, POISON};
            __builtin_amdgcn_raw_buffer_store_b128(ob, rs, (unsigned)(i % RING) * SLOT_BYTES + y_lane_off, 0, AUX_SC1);
            __builtin_amdgcn_raw_buffer_store_b128(pz, rs, (unsigned)((i + 2) % RING) * SLOT_BYTES + y_lane_off, 0, AUX_SC1);
        }
        if (dg && lane == 0) {
            d4 = __builtin_amdgcn_s_memrealtime();
            unsigned long long *q = a.diag + ((size_t)i * 2 + (wave ? 1 : 0)) * 6;
            q[0] = d0; q[1] = d0; q[2] = d2; q[3] = d3; q[4] = d4; q[5] = fails * 1000 + delay;
        }
    }
    if (tid == 0) {
        atomicMin(&a.stamps[0], t_start);
        atomicMax(&a.stamps[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

template <int NW, bool PBA = false>
double run_ref(std::vector<float *> &Wp, float *bias, float *ring, int nlayers, hipStream_t s, int replays) {
    // ring slot RING-1 holds the input; layer i reads slot (i-1)%RING and writes slot i%RING, like the persistent form
    const int L = (int)Wp.size();
    const size_t SLOT = (size_t)M * K;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nlayers; ++i)
        hipLaunchKernelGGL((layer_ref<NW, PBA>), dim3(NT * MT), dim3(NW * 64), 0, s, ring + ((i + RING - 1) % RING) * SLOT, Wp[i % L], bias,
                           ring + (i % RING) * SLOT);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double best = 1e30;
    for (int rep = 0; rep < replays; ++rep) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::high_resolution_clock::now();
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::high_resolution_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / nlayers;
        if (us < best) best = us;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main(int argc, char **argv) {
    const int L = 24;
    const int nlayers = argc > 1 ? atoi(argv[1]) : 24 * 40;      // must be a multiple of RING for the slot bookkeeping below
    const size_t SLOT = (size_t)M * K;
    std::vector<float *> Wp(L);
    std::vector<float> hp((size_t)N * K);
    float *Wall;
    CK(hipMalloc(&Wall, (size_t)L * N * K * 4));
    for (int l = 0; l < L; ++l) {
        unsigned st = 12345u + 977u * l;
        for (size_t i = 0; i < hp.size(); ++i) {
            st = st * 1664525u + 1013904223u;
            hp[i] = ((float)(st >> 8) / 16777216.0f - 0.5f) * 0.11f;         // keeps activations O(1) through ELU chains
        }
        Wp[l] = Wall + (size_t)l * N * K;
        CK(hipMemcpy(Wp[l], hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<float> hb(N), hx(SLOT);
    for (int i = 0; i < N; ++i) hb[i] = 0.01f * (float)((i * 37) % 11 - 5);
    for (size_t i = 0; i < SLOT; ++i) hx[i] = 0.5f * sinf(0.37f * (float)i);
    float *bias, *ringA, *ringB;
    CK(hipMalloc(&bias, N * 4));
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ringA, RING * SLOT * 4));
    CK(hipMalloc(&ringB, RING * SLOT * 4));
    unsigned *status; unsigned long long *stamps;
    CK(hipMalloc(&status, 4)); CK(hipMalloc(&stamps, 16));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    auto reset_ring = [&](float *ring, bool poison) {
        if (poison) {
            std::vector<unsigned> p(RING * SLOT, POISON);
            CK(hipMemcpy(ring, p.data(), p.size() * 4, hipMemcpyHostToDevice));
        } else CK(hipMemset(ring, 0, RING * SLOT * 4));
        CK(hipMemcpy(ring + (RING - 1) * SLOT, hx.data(), SLOT * 4, hipMemcpyHostToDevice));
    };

    // ---- (a) reference chain
    reset_ring(ringA, false);
    const double us_ref = run_ref<8>(Wp, bias, ringA, nlayers, s, 1);
    std::vector<float> ref(SLOT), got(SLOT);
    CK(hipMemcpy(ref.data(), ringA + ((nlayers - 1) % RING) * SLOT, SLOT * 4, hipMemcpyDeviceToHost));
    double us_ref_best = us_ref;
    for (int r = 0; r < 3; ++r) { reset_ring(ringA, false); const double u = run_ref<8>(Wp, bias, ringA, nlayers, s, 1); if (u < us_ref_best) us_ref_best = u; }
    printf("one launch per layer (hipGraph), 8 waves      : %6.2f us per layer  (%d layers)\n", us_ref_best, nlayers);
    double amax = 0; for (float v : ref) amax = fmax(amax, fabs(v));
    printf("   reference output max |y| = %.4f\n", amax);

    unsigned long long *diag;
    CK(hipMalloc(&diag, (size_t)nlayers * 12 * 8));
    std::vector<float> ref1 = ref;
    // ---- (b) persistent chain
    auto run_persist = [&](const char *name, auto kern, bool show_diag) {
        double best = 1e30, best_in = 1e30;
        bool ok = true;
        for (int rep = 0; rep < 4; ++rep) {
            reset_ring(ringB, true);
            CK(hipMemset(status, 0, 4));
            const unsigned long long init[2] = {~0ull, 0ull};
            CK(hipMemcpy(stamps, init, 16, hipMemcpyHostToDevice));
            Args a{Wall, bias, ringB, nlayers, L, status, stamps, 2000000u, diag};
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(kern, dim3(NT * MT), dim3(512), 0, s, a);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned st = 0; unsigned long long tt[2];
            CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(tt, stamps, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), ringB + ((nlayers - 1) % RING) * SLOT, SLOT * 4, hipMemcpyDeviceToHost));
            const bool same = memcmp(got.data(), ref.data(), SLOT * 4) == 0;
            if (st || !same) {
                size_t bad = 0; for (size_t i = 0; i < SLOT; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
                printf("   %s rep %d: status %u, %zu of %zu outputs differ\n", name, rep, st, bad, SLOT);
                ok = false;
            }
            best = fmin(best, ms * 1e3 / nlayers);
            best_in = fmin(best_in, (double)(tt[1] - tt[0]) * 0.01 / nlayers);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("%-46s: %6.2f us per layer (events), %6.2f in-kernel   %s\n", name, best, best_in, ok ? "bit-identical" : "MISMATCH");
        if (show_diag) {
            std::vector<unsigned long long> d((size_t)nlayers * 12);
            CK(hipMemcpy(d.data(), diag, d.size() * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 2; ++w) {
                double seg[5] = {0, 0, 0, 0, 0}, polls = 0;
                int n = 0;
                for (int i = 8; i + 1 < nlayers; ++i) {
                    const unsigned long long *q = &d[((size_t)i * 2 + w) * 6], *qn = &d[((size_t)(i + 1) * 2 + w) * 6];
                    seg[0] += (double)(q[1] - q[0]); seg[1] += (double)(q[2] - q[1]); seg[2] += (double)(q[3] - q[2]);
                    seg[3] += (double)(q[4] - q[3]); seg[4] += (double)(qn[0] - q[4]); polls += (double)q[5]; ++n;
                }
                printf("   workgroup 0 wave %d: first loads %.2f us | polling %.2f (%.1f polls) | mfma+reduce barrier %.2f | epilogue+store %.2f | loop back %.2f\n",
                       w ? 3 : 0, seg[0] / n * 0.01, seg[1] / n * 0.01, polls / n, seg[2] / n * 0.01, seg[3] / n * 0.01, seg[4] / n * 0.01);
            }
        }
    };
    if (argc > 2) {
    run_persist("persistent, poll one block", persist<8, false, 0, false>, false);
    run_persist("persistent + next W prefetched, poll one", persist<8, true, 0, false>, false);
    run_persist("persistent + prefetch, poll all blocks", persist<8, true, 1, false>, false);
    run_persist("persistent + prefetch, flag poll then fetch", persist<8, true, 2, false>, false);
    run_persist("  (diag build) flag poll then fetch", persist<8, true, 2, true>, true);
    run_persist("flag poll, buffer_inv sc1, plain fetch", persist<8, true, 2, false, 1>, false);
    run_persist("  (diag build) inv + plain fetch", persist<8, true, 2, true, 1>, true);
    run_persist("flag poll, plain fetch, NO invalidate (timing)", persist<8, true, 2, false, 2>, false);
    run_persist("  (diag build) poll one", persist<8, true, 0, true>, true);
    run_persist("  (diag build) poll all", persist<8, true, 1, true>, true);
    }
    CK(hipMemcpy(ref.data(), ref1.data(), SLOT * 4, hipMemcpyHostToHost));
    run_persist("persistent, flag poll, 1 group", persist2<8, 1, false, false>, false);
    run_persist("persistent, flag poll, 2 groups", persist2<8, 2, false, false>, false);
    run_persist("persistent, flag poll, 4 groups", persist2<8, 4, false, false>, false);
    run_persist("persistent, 2 groups + filler product", persist2<8, 2, true, false>, false);
    run_persist("persistent, fetch-as-poll, adaptive delay", persist3<8, false>, false);
    run_persist("  (diag build) fetch-as-poll", persist3<8, true>, true);
    run_persist("persistent, 1 group, W requested late", persist2<8, 1, false, false, true>, false);
    run_persist("  (diag build) 1 group", persist2<8, 1, false, true>, true);
    run_persist("  (diag build) 1 group, W late", persist2<8, 1, false, true, true>, true);
    return 0;
}
