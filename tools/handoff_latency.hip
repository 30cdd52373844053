// What does a hand-off between two workgroups cost on an MI355X?  The persistent recurrence kernel publishes a 1 KiB block with
// system-scope (sc1) buffer stores and its consumers poll / fetch it with sc1 buffer loads (k_flow.hip: flow_publish, flow_wait, flow_issue);
// the L2s of the eight XCDs are not coherent with each other, so both go through the fabric.  Here workgroup A (XCD 0) and workgroup B
// (on XCD x) play ping-pong with exactly those instructions:  A stores block i, B polls its last dword (32 bytes per poll), fetches the
// block, stores its own block i; A polls and fetches that.  One round = two hand-offs.  Workgroups land on XCD (blockIdx % 8).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/handoff_latency tools/handoff_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;

// MODE 0: poll the flag dword, then fetch the block (what the kernel does); 1: poll only (no block fetch); 2: fetch the block as the poll
template <int MODE>
__global__ __launch_bounds__(64) void pingpong(unsigned *buf, int peer, int rounds, unsigned long long *out, unsigned spin_limit, int npoll) {
    const int bid = blockIdx.x;
    if (bid != 0 && bid != peer) {
        // bystanders (npoll of them): follow workgroup 0's flag like the consumers of a layer's block that are not on the critical path here
        if (bid > npoll) return;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 2 * 1024, 0x00020000);
        unsigned spins = 0;
        for (;;) {
            const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(rs0, 63u * 16u + 12u, 0u, AUX_SC1);
            asm volatile("" ::: "memory");
            if (t >= (unsigned)rounds || ++spins > spin_limit) return;
        }
    }
    const bool is_a = bid == 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 2 * 1024, 0x00020000);
    const unsigned lane = threadIdx.x;
    const unsigned mine = is_a ? 0u : 1024u, theirs = is_a ? 1024u : 0u;
    unsigned long long t0 = 0;
    unsigned acc = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (i == 2 && is_a) t0 = wall_clock64();          // the first round warms up
        if (is_a) {
            u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16u, mine, AUX_SC1);
        }
        // wait for the peer's block i
        unsigned spins = 0;
        for (;;) {
            if (MODE == 2) {
                const u32x4 x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, theirs, AUX_SC1));
                const bool ok = x[0] == (unsigned)i && x[3] == (unsigned)i;
                if (!__any(!ok)) { acc += x[1]; break; }
            } else {
                const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(rs, 63u * 16u + 12u, theirs, AUX_SC1);
                if (t == (unsigned)i) {
                    if (MODE == 0) {
                        const u32x4 x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, theirs, AUX_SC1));
                        acc += x[1];
                    }
                    break;
                }
            }
            asm volatile("" ::: "memory");               // (the poll is a plain intrinsic: keep it inside the loop)
            if (++spins > spin_limit) { if (lane == 0) out[2] = 1; return; }
        }
        if (!is_a) {
            u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16u, mine, AUX_SC1);
        }
    }
    if (is_a && lane == 0) { out[0] = wall_clock64() - t0; out[1] = acc; }
}

int main() {
    unsigned *buf; unsigned long long *out;
    CK(hipMalloc(&buf, 2048)); CK(hipMalloc(&out, 24));
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));     // kHz
    const int rounds = 2001;
    const char *names[3] = {"flag poll, then block fetch (the kernel's protocol)", "flag poll only", "block fetch as the poll"};
    for (int mode = 0; mode < 3; ++mode)
        for (int peer : {8, 1, 2, 3, 4, 5, 6, 7}) {       // 8: the same XCD as workgroup 0; 1..7: XCD 1..7
            CK(hipMemset(buf, 0, 2048)); CK(hipMemset(out, 0, 24));
            if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(64), 0, 0, buf, peer, rounds, out, 1000000u, 0);
            if (mode == 1) hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(64), 0, 0, buf, peer, rounds, out, 1000000u, 0);
            if (mode == 2) hipLaunchKernelGGL(pingpong<2>, dim3(16), dim3(64), 0, 0, buf, peer, rounds, out, 1000000u, 0);
            CK(hipDeviceSynchronize());
            unsigned long long h[3];
            CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
            if (h[2]) { printf("mode %d peer %d: timed out\n", mode, peer); continue; }
            const double us = (double)h[0] / (double)rate * 1e3 / (rounds - 1);
            printf("%-52s  workgroup 0 (XCD 0) <-> workgroup %d (XCD %d): %.3f us per round = %.3f us per hand-off\n", names[mode], peer, peer % 8, us, us / 2);
        }
    // the same hand-off (workgroup 0 <-> workgroup 1, XCD 1) while `npoll` other workgroups poll workgroup 0's flag too
    for (int npoll : {0, 7, 15, 63, 127, 255}) {
        CK(hipMemset(buf, 0, 2048)); CK(hipMemset(out, 0, 24));
        hipLaunchKernelGGL(pingpong<0>, dim3(256), dim3(64), 0, 0, buf, 1, rounds, out, 4000000u, npoll);
        CK(hipDeviceSynchronize());
        unsigned long long h[3];
        CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        if (h[2]) { printf("npoll %d: timed out\n", npoll); continue; }
        const double us = (double)h[0] / (double)rate * 1e3 / (rounds - 1);
        printf("flag poll, then block fetch, %3d bystanders polling the same flag: %.3f us per round = %.3f us per hand-off\n", npoll, us, us / 2);
    }
    return 0;
}
