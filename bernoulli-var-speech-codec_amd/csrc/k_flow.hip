// Persistent BVRNN recurrence for gfx950: ALL frames of BVRNN.encode / BVRNN.decode (bvrnn.py:186-206,
// 222-227) in ONE kernel launch.
//
// One workgroup (8 waves) owns one 16x16 output tile (16 utterances x 16 features) of EVERY layer of the
// step, for all frames; the 64 feature tiles of an utterance group form a chain of all-to-all hand-offs
// (a layer needs the whole output of the previous one), the utterance groups are independent chains.
// There is no grid barrier and no flag: the hand-off is "poisoned buffer" dataflow.
//   * every activation buffer exists twice (frame parity) and starts filled with a NaN sentinel;
//   * a producer stores its 1 KiB output block write-through (sc1) and re-arms its block of the other
//     parity with the sentinel (its readers finished a frame ago);
//   * a consumer wave watches the last dword of each producer block it needs (one sc1 load per poll),
//     then fetches the blocks with sc1 loads (they bypass the non-coherent L1) and verifies that no
//     dword is the sentinel before it multiplies.  A value equal to the sentinel is never published.
// K is split over the 8 waves, partial tiles are summed in fixed order through LDS, wave 0 applies the
// layer's epilogue (bias, ELU, sigmoid/round/bit-mask, mel normalisation, GRU cell).  The MFMA computes
// Y^T = W X^T (A = weight fragment, B = activation fragment) so that a lane's four results are four
// consecutive features of one utterance: exactly the 16-byte granule of the next layer's operand block.
// The next layer's first weight blocks are requested before the wait, so they travel meanwhile.  The layer sequence
// is a compile-time program (bvrnn_flow_kernel<PERH, ENCODE>), its operands come from a device-resident FlowArgs.
// Every wait is bounded: a wave that waits too long records it in the model's status word and stops
// waiting for good (results are then garbage, the kernel still ends) - bvc_model_status() reports it.
//
// What sits in a compute unit for the whole launch (filler form, h_dim 1024): the weights of two wide layers (decode: three) in registers -
// the kernel needs 165 of the 256 VGPRs a wave may have - and the first layer's in the 64 KiB of LDS the kernel had left; the vector
// instructions on a hop's critical path are few on purpose (fp32 MFMAs and vector instructions share the ALUs, also within one wave:
// profiles/r04_mfma_valu_overlap.txt), and scheduling fences keep loads and checks where the source puts them (hipcc sinks loads to their
// first use and hoists checks in front of the products when left alone: profiles/r04_flow_variants.txt).
//
// Measured form of the hand-off: MI355X_MICROARCH.md (valid forms: sc1 stores, sc1 loads, data-tagged
// granules), tools/persist_bench.hip (3.3-3.8 us per 1024x1024 layer against 4.25 launch-per-layer).
#include "bvc_internal.h"

namespace bvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifndef BVC_GRU_ROUNDS_FILL
#define BVC_GRU_ROUNDS_FILL 4      // rounds in which the GRU layer's remaining weights (phi_x third) pass through the registers
#endif
// Compile-time switches (tools/flow_variants.py builds and times variants):
//   BVC_FLOW_EARLYW       1: the next layer's weights are requested right behind this layer's operand requests instead of in front of
//                         the reduction barrier (the default place).  Worth 0.6 ms per step while spill reloads sat in the epilogues
//                         (they wait, in order, for everything in flight); costs 1.4 ms without them: 54.3 against 52.8 ms per step
//   BVC_FLOW_LATEW        1: ... or behind the reduction barrier (nothing but the quantum's weights in front of it)
//   BVC_FLOW_STASH        1 (default): a wave keeps the operand blocks its filler quanta multiply (h, phi_z: fetched and verified for the
//                         layer that consumes them first) in LDS, so a quantum requests its weights only
//   BVC_FLOW_PARTNER_WAIT 1 (default): the wave that shares wave 0's SIMD (wave 4) starts its quantum's products only once wave 0 has
//                         issued the publishing store (its MFMAs take issue slots from the epilogue everybody waits for)
//   BVC_FILL_EARLY        1: a quantum's weights are requested inside the layer's segment too (behind the next layer's weights) instead of
//                         in front of the reduction barrier
//   BVC_FLOW_DIAG         1: the probing lane also stamps flags-seen / products-done / barrier-passed (tools/flow_variants.py run --diag)
// What these are about: a compute unit takes vector-memory requests in order, 64 B per clock (1 KiB per wave-instruction = 16
// clocks), and a wave that issues a request into a full queue stalls.  Everything requested in front of the reduction barrier
// therefore holds the barrier - and the publishing store behind it - back: 24 KiB per wave there (next weights, quantum weights,
// quantum operands) cost every filler layer more than a microsecond (stamps: products -> barrier 1.2-1.8 us, 0.2-0.5 without).
// Measured and dropped: two or three flag polls in flight per wave (57.9 / 59.2 against 55.7 ms per step: the polls
// themselves load the hand-off path); quanta requested behind the publishing store and multiplied a layer later, with and without a
// pre-issued poll of the next layer's flags (shorter layer spans, 43.8 against 45.2 us per encode frame, but wave 0 leaves each layer
// later: 55.0-55.9 against 54.1 ms per step); the ELU epilogue split over four waves (wave j reduces and activates output j of every
// lane, wave 0 gathers through LDS and publishes: barrier -> published stays at 0.45 us, 53.8 against 52.8 ms per step - unlike the
// GRU cell's, which did shrink from 1.5 to 0.8 us that way, the linear layers' epilogue is not bound by its instruction count).
#ifndef BVC_FLOW_EARLYW
#define BVC_FLOW_EARLYW 0
#endif
#ifndef BVC_FLOW_LATEW
#define BVC_FLOW_LATEW 0
#endif
#ifndef BVC_FLOW_STASH
#define BVC_FLOW_STASH 1
#endif
#ifndef BVC_FLOW_PARTNER_WAIT
#define BVC_FLOW_PARTNER_WAIT 1
#endif
#ifndef BVC_FILL_EARLY
#define BVC_FILL_EARLY 0
#endif
//   BVC_GRU_FAST          1: the GRU layer's remaining product, W_ih[:, :H] phi_x(d_t) (24 KiB of weights per wave, the layer
//                         was bound by that stream), gets half its weights requested a layer early (inside phi_x.4's segment), a quarter
//                         parked in LDS for the whole launch, and only the last quarter streamed inside the layer (h_dim 1024, filler form).
//                         Measured without effect (52.76 ms per step either way): with its h and phi_z parts taken out by the quanta the layer
//                         is bound by the 192 MFMAs its two waves per SIMD issue, not by the weight stream any more.  Off by default.
#ifndef BVC_GRU_FAST
#define BVC_GRU_FAST 0
#endif
//   (what round 4 measured and did not keep - resident rounds of the GRU stream, other request orders, fences in the chain kernels - is in
//   profiles/r04_flow_variants.txt)
//   BVC_GRU_FENCE         1 (default): scheduling fences around the rounds of the GRU layer's weight stream (see flow_gru)
//   BVC_GRU_DEPTH         register sets the stream runs through (default 3; 2: the request for round i + 1 in front of round i's products; up to all four rounds)
//   BVC_FLOW_PARKV        2 (default; 0 off, 1 two layers): (filler form) the weights of two wide layers (three in decode) - this wave's 8 blocks each - stay in registers for the whole launch
//                         (the filler kernels use 165 of the 256 VGPRs a wave may have): 128 KiB (decode 192) per compute unit and frame less to pull from
//                         the L2, whose fill path into the compute units is what the kernel is bound by
#ifndef BVC_FLOW_PARKV
#define BVC_FLOW_PARKV 2
#endif
//   BVC_FLOW_PARKL        1 (default): (filler form) the FIRST layer's weights (enc.0 / dec.0, the half that multiplies h) stay in LDS for the whole launch - the
//                         64 KiB that the kernel's 128 KiB left of a compute unit's 160 - instead of being requested by every frame's GRU layer,
//                         at the most crowded point of the frame
#ifndef BVC_FLOW_PARKL
#define BVC_FLOW_PARKL 1
#endif
#ifndef BVC_FLOW_PARK_SET
#define BVC_FLOW_PARK_SET 1                          // which two: 1 dec.2 + dec.4, 0 phi_x.2 + phi_x.4
#endif
#ifndef BVC_GRU_FENCE
#define BVC_GRU_FENCE 1
#endif
#ifndef BVC_GRU_DEPTH
#define BVC_GRU_DEPTH 3
#endif
#ifndef BVC_FLOW_POLL_SLEEP
#define BVC_FLOW_POLL_SLEEP 1                        // s_sleep units (64 clocks) between two polls of a wave
#endif
#ifndef BVC_FLOW_DIAG
#define BVC_FLOW_DIAG 0
#endif
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float elu1(float v) { return v > 0.0f ? v : expf(v) - 1.0f; }
__device__ __forceinline__ float sigmoid1(float v) { return 1.0f / (1.0f + expf(-v)); }

// the status word lives in host-mapped pinned memory (the host reads it without synchronising): a plain system-scope store
__device__ __forceinline__ void flow_report(unsigned *status, unsigned code) {
    __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct FlowWg {
    __amdgpu_buffer_rsrc_t rs;       // the flow region
    int lane, wave, mtile, ntile;
    int wmul;                        // 1 (0 in the hot-weights timing experiment)
    unsigned spin_limit;
    unsigned *status;
};

//   BVC_FLOW_MAXCHK       1 (default): "does a fetched block still hold the sentinel" as an unsigned maximum over the block's dwords (two v_max3_u32 and a
//                         compare per 16 bytes instead of four compares and three ors); nothing but the sentinel itself may then lie at or above
//                         it: publishable() maps every such bit pattern (negative NaNs with an all-ones payload top) to the canonical NaN
#ifndef BVC_FLOW_SPEC
#define BVC_FLOW_SPEC 0         // lin_segment: 1 = operand blocks requested without polling their flags first (verified and requested again while one holds the sentinel)
#endif
#ifndef BVC_CHAIN_SPEC
#define BVC_CHAIN_SPEC 0        // flow_layer_chains: the first chain's operand blocks are requested without polling their flags first
#endif
#ifndef BVC_CHAIN_G
#define BVC_CHAIN_G 4           // chains per reduction group of flow_layer_chains (three and more chains per workgroup)
#endif
#ifndef BVC_FLOW_MAXCHK
#define BVC_FLOW_MAXCHK 1
#endif
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ bool is_poison4(const u32x4 v) {
    if (BVC_FLOW_MAXCHK) return umax3(umax3(v[0], v[1], v[2]), v[3], 0u) >= FLOW_POISON;
    return v[0] == FLOW_POISON || v[1] == FLOW_POISON || v[2] == FLOW_POISON || v[3] == FLOW_POISON;
}

// Operand blocks [kb0, kb0 + PER) of utterance group g.mtile from the flow buffer at byte offset `buf`
// (nbdim blocks per utterance group), waiting for their producers.
// (Tried and dropped: a self-tuned delay followed by fetching the blocks directly, without the flag poll - 0.4 us per
// layer faster in tools/persist_bench.hip, no faster in the real step, whose layers are bound by the operand stream.)
struct FlowSrc { unsigned base, vl; };     // scalar byte offset of this wave's first operand block, lane offset

// Waits until the producers of blocks [kb0, kb0 + PER) of utterance group g.mtile have published them (one dword per
// producer block and poll).  A flag can be visible before the rest of its block: the consumer still verifies what it fetches.
template <int PER>
__device__ __forceinline__ FlowSrc flow_wait(const FlowWg &g, unsigned buf, int nbdim, int kb0, bool &give_up, unsigned code,
                                             unsigned &spins) {
    FlowSrc s;
    // uniform part of every address in the scalar offset, lane part in one shared VGPR
    s.base = __builtin_amdgcn_readfirstlane(buf + (unsigned)(g.mtile * nbdim + kb0) * 1024u);
    s.vl = (unsigned)g.lane * 16u;
    const unsigned fl = (unsigned)(g.lane < PER ? g.lane : PER - 1) * 1024u + 63u * 16u + 12u;
    while (!give_up) {
        const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(g.rs, fl, s.base, AUX_SC1);
        if (!__any(t == FLOW_POISON)) break;
        __builtin_amdgcn_s_sleep(BVC_FLOW_POLL_SLEEP);
        if (++spins > g.spin_limit) {
            give_up = true;
            if (g.lane == 0) flow_report(g.status, code);
        }
    }
    return s;
}

//   BVC_FLOW_INORDER      1 (default): a segment's operand blocks are requested strictly in k order and multiplied in that order, each as soon as IT has
//                         arrived (a wave's loads return in order), with the sentinel checks behind the last product.  hipcc otherwise
//                         schedules the checks - which need every block - in front of the first product and requests block 0 second to
//                         last: all eight blocks then have to be there before the first MFMA issues
#ifndef BVC_FLOW_INORDER
#define BVC_FLOW_INORDER 1
#endif
template <int PER>
__device__ __forceinline__ void flow_issue(const FlowWg &g, const FlowSrc &s, u32x4 (&xr)[PER]) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(g.rs, s.vl, s.base + (unsigned)u * 1024u, AUX_SC1));
        if (BVC_FLOW_INORDER) __builtin_amdgcn_sched_barrier(0);
    }
}

// Lane's view of a run of weight blocks: a wave-uniform byte pointer (kept in SGPRs, re-derived every frame so that the
// compiler does not hoist fourteen layers' worth of per-lane addresses out of the frame loop) plus lane * 16.
typedef const char __attribute__((address_space(1))) *GPtr;        // global (not flat) loads
__device__ __forceinline__ GPtr uniform_ptr(const float *w, size_t block) {
    unsigned long long p = reinterpret_cast<unsigned long long>(w) + (block << 10);
    asm volatile("" : "+s"(p));
    return (GPtr)p;
}
__device__ __forceinline__ f32x4 wload(GPtr ub, unsigned lane16, int blk) {
    // (Measured and not kept: the part of blk * 1024 beyond the instruction's 4095-byte immediate moved into an opaque SCALAR base - a
    // third fewer 64-bit vector adds in front of the reduction barrier, and 0.15 ms per step slower: profiles/r04_flow_variants.txt)
    return *reinterpret_cast<const f32x4 __attribute__((address_space(1))) *>(ub + lane16 + (unsigned)blk * 1024u);
}

// What a layer wants done between a segment's operand requests and its products: the next layer's weights requested
// (BVC_FLOW_EARLYW), diagnostic stamps taken.
typedef u32x4 __attribute__((address_space(3))) *LdsX;      // this wave's stash of operand blocks in LDS: [k-block][lane]
struct SegHook {
    const float *nw; int nwnb; bool pre;           // next layer's packed weights, k-blocks per row; request them here?
    unsigned long long *st_flags, *st_done;       // BVC_FLOW_DIAG: where to stamp "flags seen" / "products done" (or null)
    LdsX stash;                                    // BVC_FLOW_STASH: keep the verified operand blocks of this segment there (or null)
    const float *fw; size_t fblock; int fstride;   // BVC_FILL_EARLY: the layer's filler quantum - weights, first block, blocks between k-blocks (fw null: none)
    const float *gw; size_t gblock;                // BVC_GRU_FAST: the GRU layer's first GRU_EARLY_BLOCKS weight blocks are requested here (gw null: none)
};
constexpr int GRU_EARLY_BLOCKS = 12;               // rounds 0 and 1 of four: 4 k-blocks x 3 gates (gate-interleaved: consecutive 1 KiB blocks)
__device__ __forceinline__ SegHook no_hook() { return SegHook{nullptr, 0, false, nullptr, nullptr, (LdsX)0, nullptr, 0, 1, nullptr, 0}; }

// acc += W[ntile rows][segment] . X[segment]   for this wave's share of the segment's k-blocks
// GRUPRE / FEARLY: (compile time) the hook carries GRU weights / a filler quantum's weights to request behind the operands.
template <int PER, int PERN, bool GRUPRE = false, bool FEARLY = false>
__device__ __forceinline__ void lin_segment(const FlowWg &g, const float *w, int wnb, int nb, unsigned buf, bool w_ready,
                                            f32x4 (&wv)[PER], f32x4 &acc, bool &give_up, unsigned code, int kb_off,
                                            const SegHook &hk, f32x4 (&wn)[PERN], f32x4 (&fwv)[PERN], f32x4 (&gqv)[GRU_EARLY_BLOCKS]) {
    const int kb0 = kb_off + g.wave * PER;
    if (PER == 1 && kb0 >= nb) return;                     // wave-uniform: fewer k-blocks than waves
    if (!w_ready) {
        const GPtr ub = uniform_ptr(w, ((size_t)g.ntile * wnb + kb0) * g.wmul);
#pragma unroll
        for (int u = 0; u < PER; ++u) wv[u] = wload(ub, (unsigned)g.lane * 16u, u);
    }
    unsigned spins = 0;
    FlowSrc src;
    if (BVC_FLOW_SPEC) {                                   // no flag poll: the blocks themselves are requested until none holds the sentinel
        src.base = __builtin_amdgcn_readfirstlane(buf + (unsigned)(g.mtile * nb + kb0) * 1024u);
        src.vl = (unsigned)g.lane * 16u;
    } else {
        src = flow_wait<PER>(g, buf, nb, kb0, give_up, code, spins);
    }
    if (BVC_FLOW_DIAG && hk.st_flags) *hk.st_flags = __builtin_amdgcn_s_memrealtime();
    const f32x4 acc_in = acc;
    u32x4 xr[PER];
    flow_issue<PER>(g, src, xr);
    // behind the operand requests (nothing this layer waits for queues behind them; a wave's loads return in order):
    if (hk.pre) {                                          // the next layer's weights
        const GPtr ub = uniform_ptr(hk.nw, ((size_t)g.ntile * hk.nwnb + g.wave * PERN) * g.wmul);
#pragma unroll
        for (int u = 0; u < PERN; ++u) wn[u] = wload(ub, (unsigned)g.lane * 16u, u);
    }
    if (GRUPRE) {                                          // the first half of the GRU layer's weights
        const GPtr ug = uniform_ptr(hk.gw, hk.gblock * g.wmul);
#pragma unroll
        for (int i = 0; i < GRU_EARLY_BLOCKS; ++i) gqv[i] = wload(ug, (unsigned)g.lane * 16u, i);
    }
    if (FEARLY) {                                          // the layer's filler quantum's weights (zero blocks if it has none)
        const GPtr uf = uniform_ptr(hk.fw, hk.fblock * g.wmul);
#pragma unroll
        for (int u = 0; u < PERN; ++u) fwv[u] = hk.fw ? wload(uf, (unsigned)g.lane * 16u, u * hk.fstride) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (;;) {
        // The blocks are multiplied as they arrive (the loads return in order); whether one of them still held the
        // sentinel is only known at the end: then the products are thrown away and everything is fetched again.
        f32x4 a2 = acc_in;
        bool bad = false;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (!BVC_FLOW_INORDER) bad |= is_poison4(xr[u]);
            const f32x4 xv = __builtin_bit_cast(f32x4, xr[u]);
#pragma unroll
            for (int e = 0; e < 4; ++e) a2 = mfma16(wv[u][e], xv[e], a2);
            if (BVC_FLOW_INORDER) __builtin_amdgcn_sched_barrier(0);       // block u's products before block u + 1 is waited for
        }
        if (BVC_FLOW_INORDER) {
#pragma unroll
            for (int u = 0; u < PER; ++u) bad |= is_poison4(xr[u]);
        }
        if (!(__any(bad) && !give_up)) {
            acc = a2;
            if (BVC_FLOW_STASH && hk.stash) {
#pragma unroll
                for (int u = 0; u < PER; ++u) hk.stash[(kb_off / 8 * PER + u) * 64 + g.lane] = xr[u];
            }
            break;
        }
        if (++spins > g.spin_limit) {
            give_up = true;
            if (g.lane == 0) flow_report(g.status, code);
        }
        flow_issue<PER>(g, src, xr);
    }
    if (BVC_FLOW_DIAG && hk.st_done) *hk.st_done = __builtin_amdgcn_s_memrealtime();
}
template <int PER>
__device__ __forceinline__ void lin_segment(const FlowWg &g, const float *w, int wnb, int nb, unsigned buf, bool w_ready,
                                            f32x4 (&wv)[PER], f32x4 &acc, bool &give_up, unsigned code, int kb_off = 0) {
    f32x4 none[1], nof[1], nog[GRU_EARLY_BLOCKS];
    lin_segment<PER, 1>(g, w, wnb, nb, buf, w_ready, wv, acc, give_up, code, kb_off, no_hook(), none, nof, nog);
}

// One round of a GRU segment: HALF k-blocks x 3 gates of weights (gate-interleaved [n/16][k/16][gate][lane][4]).
template <int HALF>
__device__ __forceinline__ void gru_issue_w(GPtr ub, unsigned l16, int h0, f32x4 (&w3)[HALF][3]) {
#pragma unroll
    for (int u = 0; u < HALF; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) w3[u][q] = wload(ub, l16, (h0 + u) * 3 + q);
}
template <int PER, int HALF>
__device__ __forceinline__ void gru_round(const f32x4 (&w3)[HALF][3], const u32x4 (&xr)[PER], int h0, f32x4 (&acc)[3]) {
#pragma unroll
    for (int u = 0; u < HALF; ++u) {
        const f32x4 xv = __builtin_bit_cast(f32x4, xr[h0 + u]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = mfma16(w3[u][q][e], xv[e], acc[q]);
    }
}

// the same round on six consecutive weight blocks [k-block][gate] (BVC_GRU_FAST)
template <int PER>
__device__ __forceinline__ void gru_round6(const f32x4 *w6, const u32x4 (&xr)[PER], int h0, f32x4 (&acc)[3]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f32x4 xv = __builtin_bit_cast(f32x4, xr[h0 + u]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = mfma16(w6[u * 3 + q][e], xv[e], acc[q]);
    }
}

// operand blocks of an input this wave has already fetched and verified earlier in the frame (h at the first layer, phi_z
// at dec.0): complete, and not re-armed before the next frame, so neither a flag wait nor a check is needed
template <int PER>
__device__ __forceinline__ void gru_issue_known(const FlowWg &g, unsigned buf, int nb, u32x4 (&xr)[PER]) {
    const int kb0 = g.wave * PER;
    FlowSrc s;
    s.base = __builtin_amdgcn_readfirstlane(buf + (unsigned)(g.mtile * nb + kb0) * 1024u);
    s.vl = (unsigned)g.lane * 16u;
    flow_issue<PER>(g, s, xr);
}

// operand blocks of the input that is being produced right now: wait, fetch, verify
template <int PER>
__device__ __forceinline__ void gru_fetch_fresh(const FlowWg &g, unsigned buf, int nb, u32x4 (&xr)[PER], bool &give_up, unsigned code) {
    const int kb0 = g.wave * PER;
    unsigned spins = 0;
    const FlowSrc src = flow_wait<PER>(g, buf, nb, kb0, give_up, code, spins);
    bool again;
    do {
        flow_issue<PER>(g, src, xr);
        bool bad = false;
#pragma unroll
        for (int u = 0; u < PER; ++u) bad |= is_poison4(xr[u]);
        again = __any(bad) && !give_up;
        if (again && ++spins > g.spin_limit) {
            give_up = true;
            if (g.lane == 0) flow_report(g.status, code);
        }
    } while (again);
}

// One filler quantum: acc += W[ntile rows][this wave's k-blocks] . X for an input that is complete and was verified by this very
// wave earlier in the frame (no wait, no check).  GATE >= 0: gate-interleaved GRU weights, that gate only; GATE == -1: plain
// packed weights; GATE == -2: no filler.  The operands are REQUESTED before a layer's reduction barrier and MULTIPLIED behind its
// epilogue, i.e. while the inputs of the next layer are still being produced.
struct FlowFill { const float *w; int wnb; int nb; unsigned buf; };
template <int PER, int GATE>
__device__ __forceinline__ bool fill_active(const FlowWg &g, const FlowFill &f) {
    return GATE != -2 && g.ntile < f.nb && !(PER == 1 && g.wave >= f.nb);       // (these products have h_dim outputs and inputs)
}
template <int PER, int GATE>
__device__ __forceinline__ void fill_issue(const FlowWg &g, const FlowFill &f, f32x4 (&wv)[PER], u32x4 (&xr)[PER]) {
    if (!fill_active<PER, GATE>(g, f)) {                   // (defined on every path: no stale value stays live across the frame loop)
#pragma unroll
        for (int u = 0; u < PER; ++u) { wv[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; xr[u] = (u32x4){0u, 0u, 0u, 0u}; }
        return;
    }
    const int kb0 = g.wave * PER;
    const unsigned l16 = (unsigned)g.lane * 16u;
    const GPtr ub = uniform_ptr(f.w, (GATE >= 0 ? ((size_t)g.ntile * f.wnb + kb0) * 3 + GATE : (size_t)g.ntile * f.wnb + kb0) * g.wmul);
#pragma unroll
    for (int u = 0; u < PER; ++u) wv[u] = wload(ub, l16, GATE >= 0 ? u * 3 : u);
    if (!BVC_FLOW_STASH) gru_issue_known<PER>(g, f.buf, f.nb, xr);
}
template <int PER, int GATE>
__device__ __forceinline__ void fill_multiply(const FlowWg &g, const FlowFill &f, const f32x4 (&wv)[PER], const u32x4 (&xr)[PER], f32x4 &acc,
                                              LdsX stash) {
    if (!fill_active<PER, GATE>(g, f)) return;
    u32x4 xs[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) xs[u] = BVC_FLOW_STASH ? stash[u * 64 + g.lane] : xr[u];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const f32x4 xv = __builtin_bit_cast(f32x4, xs[u]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma16(wv[u][e], xv[e], acc);
    }
}

__device__ __forceinline__ u32x4 publishable(f32x4 o, bool rowok) {
    u32x4 b = __builtin_bit_cast(u32x4, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (BVC_FLOW_MAXCHK ? b[j] >= FLOW_POISON : b[j] == FLOW_POISON) b[j] = 0x7FC00000u;       // never publish the sentinel as data
        if (!rowok) b[j] = 0u;                             // padding rows of the last utterance group stay zero
    }
    return b;
}

template <bool C, typename A, typename B>
__device__ __forceinline__ decltype(auto) pick(A &a, B &b) {
    if constexpr (C) return (a);
    else return (b);
}

// per-workgroup state of the frame loop
// The kernel arguments are read through a constant-address-space pointer that is made opaque once per frame: their
// fields then come in by scalar loads where they are used, instead of ~100 SGPRs being filled (and spilled) up front.
typedef const FlowArgs __attribute__((address_space(4))) *FlowArgsC;

template <typename P>
__device__ __forceinline__ FlowLin L(const P &l) {        // member-wise copy out of the constant address space
    FlowLin r;
    r.w = l.w; r.bias = l.bias; r.wnb = l.wnb; r.pad_ = 0;
    return r;
}

struct FlowCtx {
    FlowWg g;
    FlowArgsC a;
    float *red_lin, *red_gru;
    float *red_chain;        // MULTI: [2][4][NW][256] partial tiles of a group of chains (flow_layer_chains)
    unsigned sb;             // bytes per flow buffer slot: kept in a register (every layer needs it before its first request; a scalar
                             // load there is a cache round trip on the critical path)
    LdsX stash;              // BVC_FLOW_STASH: this wave's operand blocks of the quanta's input
    LdsX gpark;              // BVC_GRU_FAST: this wave's parked quarter of the GRU layer's weights: [6 blocks][lane]
    LdsX lpark;              // BVC_FLOW_PARKL: this wave's eight blocks of the first layer's weights (same LDS as gpark: one or the other)
    volatile unsigned __attribute__((address_space(3))) *pubflag;      // BVC_FLOW_PARTNER_WAIT: hop count of wave 0's last publishing store
    unsigned par;            // frame parity
    long long t, fr;         // frame; (utterance, frame) index of this lane's row
    int row;
    bool rowok, give_up;
    bool pre_now;            // MULTI: the next layer's weights are requested by the LAST chain of a layer only (true otherwise)
    bool probe;              // this lane records layer entry / exit times (bench instrumentation)
    unsigned hopctr;
    // FILL: this wave's partial sums of the products whose inputs exist long before their layer - W_hh h, W_ih[:, H:] phi_z, the
    // h half of dec.0 - computed one quantum at a time in the waits behind other layers
    f32x4 fgh[3], fgi[3], fd0;
};

__device__ __forceinline__ unsigned long long *flow_stamp_slot(const FlowCtx &c, int hopid, int which) {
    if (!c.probe) return nullptr;
    const auto &a = *c.a;
    return a.probe + ((long long)which * a.T + c.t) * a.probe_nodes + (hopid - a.probe_first);
}
// which: 0 layer entered, 1 output published (wave 0); BVC_FLOW_DIAG: 2 flags of the (last) segment seen, 3 its products done,
// 4 barrier passed, 5 layer left (behind the filler quantum)
__device__ __forceinline__ void flow_stamp(const FlowCtx &c, int hopid, int which) {
    unsigned long long *p = flow_stamp_slot(c, hopid, which);
    if (p) *p = __builtin_amdgcn_s_memrealtime();
}

// Wave 0 of a workgroup, after the barrier: sum the waves' partial tiles (fixed order), apply the layer's epilogue and publish the
// 1 KiB block; re-arm (poison) this workgroup's block of the other frame parity.
template <int EPI, bool ADD, bool REARM_H, int NW>
__device__ __forceinline__ void flow_publish(const FlowCtx &c, int hopid, const float *r, int n0, unsigned ytile, int ntiles, int out,
                                             const f32x4 bias4, const f32x4 add4, const f32x4 mean4, const f32x4 std4, float bitsv) {
    const FlowWg &g = c.g;
    const auto &a = *c.a;
    const int lane = g.lane;
    __builtin_amdgcn_s_setprio(3);                     // the publishing wave goes first: its SIMD partner may be multiplying a filler
    f32x4 v = *reinterpret_cast<const f32x4 *>(r + lane * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4 *>(r + (w * 64 + lane) * 4);
    v += bias4;
    f32x4 o, pr = {0.f, 0.f, 0.f, 0.f};
    if (EPI == FE_ELU || EPI == FE_ELU_KEEP) {
        if (ADD) v += add4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = elu1(v[j]);
    } else if (EPI == FE_CODE) {                       // bvrnn.py:189-194
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pr[j] = sigmoid1(v[j]);
            float z = rintf(pr[j]);                    // round half to even (torch.round)
            if (a.var_bit) z = (bitsv > (float)(n0 + j)) ? z : (z != z ? z : 0.5f);    // z*m + 0.5*(1-m): NaN * 0 is NaN (bvrnn.py:193-194)
            o[j] = z;
        }
    } else {                                           // FE_MEL, bvrnn.py:202-204
        if (a.mel && c.rowok) *reinterpret_cast<f32x4 *>(a.mel + c.fr * (ntiles * 16) + n0) = v;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean4[j]) / std4[j];
    }
    unsigned pv = FLOW_POISON;
    asm volatile("" : "+v"(pv));                       // re-materialised here (hoisted out of the frame loop it gets spilled)
    const u32x4 poison4 = {pv, pv, pv, pv};
    const unsigned ob = (unsigned)(out * 2) * c.sb;
    __builtin_amdgcn_raw_buffer_store_b128(publishable(o, c.rowok), g.rs, ob + c.par * c.sb + ytile, 0, AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(poison4, g.rs, ob + (c.par ^ 1u) * c.sb + ytile, 0, AUX_SC1);
    // The slot that will receive h(t+1) still holds h(t-1).  This layer's input was produced by workgroups that had all
    // consumed h(t), i.e. had all finished frame t-1: nobody reads h(t-1) any more (same tile shape as this layer's).
    if (REARM_H) __builtin_amdgcn_raw_buffer_store_b128(poison4, g.rs, (unsigned)(FB_H * 2 + (c.par ^ 1u)) * c.sb + ytile, 0, AUX_SC1);
    if (EPI == FE_ELU_KEEP && c.rowok && a.keep) *reinterpret_cast<f32x4 *>(a.keep + c.fr * (ntiles * 16) + n0) = o;     // (behind the hand-off stores; encode keeps u only for the fused forward)
    if (EPI == FE_CODE && c.rowok) {                    // the call's outputs: behind the hand-off stores as well (-0.06 ms per step)
        *reinterpret_cast<f32x4 *>(a.codes + c.fr * (ntiles * 16) + n0) = o;
        if (a.prob) *reinterpret_cast<f32x4 *>(a.prob + c.fr * (ntiles * 16) + n0) = pr;
    }
    __builtin_amdgcn_s_setprio(0);
    flow_stamp(c, hopid, 1);
}

// One layer: y = epi( sum_s W_s . x_s + bias [+ addend] ).  PER k-blocks per wave and segment (compile time), one or
// two segments (the one whose input is produced last comes last), PRE_IN: wv already holds segment 0's weights,
// PRE_OUT: request `nxt`'s weights (PERN blocks per wave) into wn (inside the last segment, or before the reduction).
// FGATE != -2: a filler quantum rides on this layer - its weights are requested in front of the reduction barrier (or inside the
// segment: BVC_FILL_EARLY) and multiplied behind the barrier, in the shadow of the epilogue and the hand-off.
// GRUPRE: (BVC_GRU_FAST) this is the layer in front of the GRU layer - the first half of the GRU's weights is requested inside its segment (gq).
template <int PER, int EPI, bool TWO, bool ADD, bool PRE_IN, bool PRE_OUT, int PERN, bool REARM_H = false, int FGATE = -2, int NW = 8, bool GRUPRE = false>
__device__ __forceinline__ void flow_layer(FlowCtx &c, int hopid, const FlowLin l0, int src0, const FlowLin l1, int src1,
                                           int nb, int ntiles, int out, f32x4 (&wv)[PER], const FlowLin nxt, f32x4 (&wn)[PERN],
                                           f32x4 (&gq)[GRU_EARLY_BLOCKS],
                                           const f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, const FlowFill fill = FlowFill{nullptr, 0, 0, 0u},
                                           f32x4 *facc = nullptr) {
    const FlowWg &g = c.g;
    const auto &a = *c.a;
    f32x4 fw[PERN];
    u32x4 fx[PERN];
    if (g.ntile >= ntiles) {                               // uniform per workgroup (layers narrower than h_dim)
        if (PRE_OUT && c.pre_now) {
#pragma unroll
            for (int u = 0; u < PERN; ++u) wn[u] = (f32x4){0.f, 0.f, 0.f, 0.f};      // defined on every path: no value lives across the layer
        }
        if (FGATE != -2) {                                 // nothing else to do in this layer: the whole quantum right away
            fill_issue<PERN, FGATE>(g, fill, fw, fx);
            fill_multiply<PERN, FGATE>(g, fill, fw, fx, *facc, c.stash);
        }
        if (BVC_GRU_FAST && GRUPRE) {
#pragma unroll
            for (int i = 0; i < GRU_EARLY_BLOCKS; ++i) gq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};     // defined on every path (see wn)
        }
        return;
    }
    int lane = g.lane;
    const int wave = g.wave;
    // (opaque per layer: what is derived from the lane - tile offsets, feature indices, their float forms - is then recomputed here
    // instead of being hoisted out of the frame loop for all fourteen layers and SPILLED; a spill reload in an epilogue waits, in
    // order, for every prefetch the wave has in flight)
    asm volatile("" : "+v"(lane));
    const unsigned code = (unsigned)((c.t << 4) | (unsigned)hopid) | 0x80000000u;
    const int n0 = g.ntile * 16 + (lane >> 4) * 4;
    const unsigned ytile = (unsigned)((g.mtile * ntiles + g.ntile) * 1024 + lane * 16);
    flow_stamp(c, hopid, 0);
    // ---- epilogue operands of wave 0, requested up front
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, add4 = {0.f, 0.f, 0.f, 0.f}, mean4 = {0.f, 0.f, 0.f, 0.f}, std4 = {1.f, 1.f, 1.f, 1.f};
    float bitsv = 0.0f;
    if (wave == 0) {
        if (l0.bias) bias4 = *reinterpret_cast<const f32x4 *>(l0.bias + n0);
        if (ADD && c.rowok) add4 = *reinterpret_cast<const f32x4 *>(a.part0 + c.fr * (ntiles * 16) + n0);
        if (EPI == FE_CODE && a.var_bit && c.rowok) bitsv = a.bits[c.fr];
        if (EPI == FE_MEL) {
            mean4 = *reinterpret_cast<const f32x4 *>(a.mean + n0);
            std4 = *reinterpret_cast<const f32x4 *>(a.stdv + n0);
        }
    }
    f32x4 acc = acc0;
    // EARLY: the next layer's weights are requested inside the (last) segment, right behind its operand requests
    constexpr bool EARLY = BVC_FLOW_EARLYW && PRE_OUT && PER > 1;
    constexpr bool FEARLY = BVC_FILL_EARLY && FGATE != -2 && PER > 1;
    SegHook hk = no_hook();
    if (BVC_FLOW_DIAG) { hk.st_flags = flow_stamp_slot(c, hopid, 2); hk.st_done = flow_stamp_slot(c, hopid, 3); }
    // the layer behind which the first quantum of a product is requested is the one that fetches that product's input
    if (BVC_FLOW_STASH && FGATE == 0) hk.stash = c.stash;
    if (PER == 1) {                                        // a narrow input (<= 8 k-blocks): one block per wave and pass
        for (int off = 0; off < nb; off += NW)
            lin_segment<PER, PERN>(g, l0.w, l0.wnb, nb, (unsigned)(src0 * 2 + c.par) * c.sb, false, wv, acc, c.give_up, code, off, hk, wn, fw, gq);
        if (TWO)                                           // (h_dim <= 128 without filler quanta: dec.0 of encode has both halves here)
            for (int off = 0; off < nb; off += NW)
                lin_segment<PER, PERN>(g, l1.w, l1.wnb, nb, (unsigned)(src1 * 2 + c.par) * c.sb, false, wv, acc, c.give_up, code, off, hk, wn, fw, gq);
    } else {
        SegHook hl = hk;
        if (EARLY) { hl.nw = nxt.w; hl.nwnb = nxt.wnb; hl.pre = c.pre_now; }
        constexpr bool GP = BVC_GRU_FAST && GRUPRE;
        if (GP) { hl.gw = a.w_ihx; hl.gblock = ((size_t)g.ntile * 2 * a.hb + wave * PER) * 3; }
        if (FEARLY && fill_active<PERN, FGATE>(g, fill)) {
            static_assert(!FEARLY || BVC_FLOW_STASH, "BVC_FILL_EARLY requests the quantum's weights only: its input must come from the stash");
            hl.fw = fill.w;
            hl.fblock = FGATE >= 0 ? ((size_t)g.ntile * fill.wnb + wave * PERN) * 3 + FGATE : (size_t)g.ntile * fill.wnb + wave * PERN;
            hl.fstride = FGATE >= 0 ? 3 : 1;
        }
        if (TWO) {
            lin_segment<PER>(g, l0.w, l0.wnb, nb, (unsigned)(src0 * 2 + c.par) * c.sb, PRE_IN, wv, acc, c.give_up, code);
            lin_segment<PER, PERN, GP, FEARLY>(g, l1.w, l1.wnb, nb, (unsigned)(src1 * 2 + c.par) * c.sb, false, wv, acc, c.give_up, code, 0, hl, wn, fw, gq);
        } else {
            lin_segment<PER, PERN, GP, FEARLY>(g, l0.w, l0.wnb, nb, (unsigned)(src0 * 2 + c.par) * c.sb, PRE_IN, wv, acc, c.give_up, code, 0, hl, wn, fw, gq);
        }
    }
    constexpr bool LATE = BVC_FLOW_LATEW && PRE_OUT && !EARLY;
    if (PRE_OUT && !EARLY && !LATE && c.pre_now) {         // the next layer's weights travel during the reduction and the wait
        const GPtr ub = uniform_ptr(nxt.w, ((size_t)g.ntile * nxt.wnb + wave * PERN) * g.wmul);
#pragma unroll
        for (int u = 0; u < PERN; ++u) wn[u] = wload(ub, (unsigned)lane * 16u, u);
    }
    // The quantum's operands are requested in front of the barrier and multiplied behind it.  (Round 2 measured the other orders with
    // the quantum's input re-fetched from memory: the publishing wave requesting its own behind its store, or every wave behind the
    // store: 56.8 / 59.1 against 56.4 ms per step - the quantum's own fetch is then exposed behind the layer.)
    if (FGATE != -2 && !FEARLY) fill_issue<PERN, FGATE>(g, fill, fw, fx);
    float *r = c.red_lin + (c.hopctr & 1u) * (NW * 256);
    ++c.hopctr;
    *reinterpret_cast<f32x4 *>(r + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 4);
    if (LATE && wave != 0 && c.pre_now) {                  // (wave 0: behind its store)
        const GPtr ub = uniform_ptr(nxt.w, ((size_t)g.ntile * nxt.wnb + wave * PERN) * g.wmul);
#pragma unroll
        for (int u = 0; u < PERN; ++u) wn[u] = wload(ub, (unsigned)lane * 16u, u);
    }
    if (wave == 0) {
        flow_publish<EPI, ADD, REARM_H, NW>(c, hopid, r, n0, ytile, ntiles, out, bias4, add4, mean4, std4, bitsv);
        if (BVC_FLOW_PARTNER_WAIT && NW == 8 && FGATE != -2) *c.pubflag = c.hopctr;
        if (LATE && c.pre_now) {
            const GPtr ub = uniform_ptr(nxt.w, ((size_t)g.ntile * nxt.wnb + wave * PERN) * g.wmul);
#pragma unroll
            for (int u = 0; u < PERN; ++u) wn[u] = wload(ub, (unsigned)lane * 16u, u);
        }
    } else if (BVC_FLOW_PARTNER_WAIT && NW == 8 && wave == 4 && FGATE != -2) {
        // waves 0 and 4 share a SIMD: this wave's MFMAs would take issue slots from the epilogue everybody is waiting for
        // (stamps: barrier -> published 0.87 us with the partner multiplying, 0.46 with it waiting, 0.42 in layers without a quantum)
        for (int i = 0; i < 4096 && *c.pubflag != c.hopctr; ++i) __builtin_amdgcn_s_sleep(2);
    }
    if (FGATE != -2) fill_multiply<PERN, FGATE>(g, fill, fw, fx, *facc, c.stash);
    if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 5);
}

// MULTI: one single-segment layer for ALL chains of this workgroup.  The chains' products are software-pipelined (while chain ci is
// multiplied, the operand blocks of chain ci+1 are on their way - fetched without a flag poll: their producers published them while
// this workgroup was busy with other chains - and verified like every fetch; a block that still holds the sentinel sends the wave
// through the ordinary wait-and-fetch path) and reduced in GROUPS of up to four chains: four accumulators per wave, ONE reduction
// barrier per group, then wave k of the group's four publishing waves (0-3 for even groups, 4-7 for odd ones) runs chain k's
// epilogue while everybody else goes on.  (One barrier and one single-wave epilogue per CHAIN - round 2 - made wave 0 the pole of
// every chain: its own products, then the epilogue, with seven waves waiting at the next chain's barrier.)  The partial tiles of a
// group lie in LDS slot (group & 1): the next writers of a slot are two groups away, and a group's partials are only written
// after ALL its products, whose operands - the previous layer's outputs of those chains from every workgroup, this one included -
// exist only once the publishing waves have read what they publish.  Same arithmetic, same order per output as flow_layer.
// TWO: y = epi(W0 x0 + W1 x1 + bias) - the group's chains are first multiplied with segment 0 (its weights are fetched here, into
// registers of their own), then with segment 1 (wv), into the same accumulators: per output the order of flow_layer's two segments.
template <int PER, int EPI, bool ADD, bool PRE_IN, bool PRE_OUT, int PERN, bool REARM_H, int NW, bool TWO = false>
__device__ __forceinline__ void flow_layer_chains(FlowCtx &c, int hopid, const FlowLin l0, int src0, int nb, int ntiles, int out,
                                                  f32x4 (&wv)[PER], const FlowLin nxt, f32x4 (&wn)[PERN], int mt0, int nch, long long T,
                                                  const FlowLin lfirst = FlowLin{nullptr, nullptr, 0, 0}, int srcfirst = 0) {
    static_assert(PER > 1, "narrow inputs take the per-chain path");
    static_assert(NW == 8, "two sets of four publishing waves");
    constexpr int GM = 4;                                  // at most four chains per reduction group ...
    const int G = nch >= 3 ? BVC_CHAIN_G : 1;              // ... and one with two chains per workgroup (measured: 128 x 5 s 6,390 audio-s/s with
                                                           // one chain per barrier against 6,250 with both behind one; 256 x 5 s 6,790 / 6,960)
    FlowWg &g = c.g;
    const auto &a = *c.a;
    if (g.ntile >= ntiles) {
        if (PRE_OUT) {
#pragma unroll
            for (int u = 0; u < PERN; ++u) wn[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    int lane = g.lane;
    const int wave = g.wave;
    asm volatile("" : "+v"(lane));                         // (see flow_layer)
    const unsigned code = (unsigned)((c.t << 4) | (unsigned)hopid) | 0x80000000u;
    const int n0 = g.ntile * 16 + (lane >> 4) * 4;
    const int kb0 = wave * PER;
    const unsigned bufb = (unsigned)(src0 * 2 + c.par) * c.sb;
    if (!PRE_IN) {
        const GPtr ub = uniform_ptr(l0.w, ((size_t)g.ntile * l0.wnb + kb0) * g.wmul);
#pragma unroll
        for (int u = 0; u < PER; ++u) wv[u] = wload(ub, (unsigned)lane * 16u, u);
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, mean4 = {0.f, 0.f, 0.f, 0.f}, std4 = {1.f, 1.f, 1.f, 1.f};
    if (l0.bias) bias4 = *reinterpret_cast<const f32x4 *>(l0.bias + n0);         // (every wave may publish a chain)
    if (EPI == FE_MEL) {
        mean4 = *reinterpret_cast<const f32x4 *>(a.mean + n0);
        std4 = *reinterpret_cast<const f32x4 *>(a.stdv + n0);
    }
    unsigned spins = 0;
    flow_stamp(c, hopid, 0);
    f32x4 wfirst[TWO ? PER : 1];
    if (TWO) {
        const GPtr ub = uniform_ptr(lfirst.w, ((size_t)g.ntile * lfirst.wnb + kb0) * g.wmul);
#pragma unroll
        for (int u = 0; u < PER; ++u) wfirst[u] = wload(ub, (unsigned)lane * 16u, u);
    }
    u32x4 xc[PER], xn[PER];
    for (int g0 = 0, grp = 0; g0 < nch; g0 += G, ++grp) {
        const int gn = nch - g0 < G ? nch - g0 : G;
        f32x4 accs[GM];
#pragma unroll
        for (int k = 0; k < GM; ++k) accs[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pass = TWO ? 0 : 1; pass < 2; ++pass) {
        const unsigned bufp = (TWO && pass == 0) ? (unsigned)(srcfirst * 2 + c.par) * c.sb : bufb;
        const f32x4 (&wp)[TWO ? PER : 1] = wfirst;         // (pass 0 only)
        // the first chain of this pass (of this group): wait for its producers, request its blocks
        if (TWO || g0 == 0) {
            g.mtile = mt0 + g0;
            FlowSrc s0;
            if (BVC_CHAIN_SPEC) {                          // no flag poll (a round trip to the fabric): request the blocks, verify them below
                s0.base = __builtin_amdgcn_readfirstlane(bufp + (unsigned)(g.mtile * nb + kb0) * 1024u);
                s0.vl = (unsigned)lane * 16u;
            } else {
                s0 = flow_wait<PER>(g, bufp, nb, kb0, c.give_up, code, spins);
            }
            flow_issue<PER>(g, s0, xc);
        }
#pragma unroll
        for (int k = 0; k < GM; ++k) {
            if (k < gn) {                                  // (uniform)
                const int ci = g0 + k;
                g.mtile = mt0 + ci;
                if (TWO ? k + 1 < gn : ci + 1 < nch) {     // the next chain's blocks: requested before this chain's are waited for
                    FlowSrc sn;                            // (TWO: within the group's pass; else on across the groups)
                    sn.base = __builtin_amdgcn_readfirstlane(bufp + (unsigned)((g.mtile + 1) * nb + kb0) * 1024u);
                    sn.vl = (unsigned)lane * 16u;
                    flow_issue<PER>(g, sn, xn);
                } else {
#pragma unroll
                    for (int u = 0; u < PER; ++u) xn[u] = (u32x4){0u, 0u, 0u, 0u};
                }
                bool bad = false;
#pragma unroll
                for (int u = 0; u < PER; ++u) bad |= is_poison4(xc[u]);
                while (__any(bad) && !c.give_up) {         // (rare) not published yet, or a flag ahead of its block: wait, fetch again
                    const FlowSrc sr = flow_wait<PER>(g, bufp, nb, kb0, c.give_up, code, spins);
                    flow_issue<PER>(g, sr, xc);
                    bad = false;
#pragma unroll
                    for (int u = 0; u < PER; ++u) bad |= is_poison4(xc[u]);
                    if (++spins > g.spin_limit) {
                        c.give_up = true;
                        if (lane == 0) flow_report(g.status, code);
                    }
                }
                if (BVC_FLOW_DIAG && k == 0 && g0 == 0 && pass == 1) flow_stamp(c, hopid, 2);     // first chain's operands here
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const f32x4 xv = __builtin_bit_cast(f32x4, xc[u]);
                    const f32x4 wu = (TWO && pass == 0) ? wp[TWO ? u : 0] : wv[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[k] = mfma16(wu[e], xv[e], accs[k]);
                }
#pragma unroll
                for (int u = 0; u < PER; ++u) xc[u] = xn[u];
            }
        }
      }
        if (BVC_FLOW_DIAG && g0 == 0) flow_stamp(c, hopid, 3);
        if (PRE_OUT && g0 + G >= nch) {                    // the next layer's weights travel during the last group's reduction
            const GPtr ub = uniform_ptr(nxt.w, ((size_t)g.ntile * nxt.wnb + wave * PERN) * g.wmul);
#pragma unroll
            for (int u = 0; u < PERN; ++u) wn[u] = wload(ub, (unsigned)lane * 16u, u);
        }
        // this wave's chain to publish (if any) and its epilogue operands, requested in front of the barrier
        const int pk = wave - (grp & 1) * GM;              // publishing waves of this group: (grp & 1) * 4 + k
        const bool pub = pk >= 0 && pk < gn;
        FlowCtx ce = c;
        ce.g.mtile = mt0 + g0 + (pub ? pk : 0);
        ce.row = ce.g.mtile * 16 + (lane & 15);
        ce.rowok = ce.row < a.B;
        ce.fr = (long long)ce.row * T + c.t;
        ce.probe = c.probe && pk == 0 && g0 == 0;
        f32x4 add4 = {0.f, 0.f, 0.f, 0.f};
        float bitsv = 0.0f;
        if (pub) {
            if (ADD && ce.rowok) add4 = *reinterpret_cast<const f32x4 *>(a.part0 + ce.fr * (ntiles * 16) + n0);
            if (EPI == FE_CODE && a.var_bit && ce.rowok) bitsv = a.bits[ce.fr];
        }
        float *rg = c.red_chain + (grp & 1) * (GM * NW * 256);
#pragma unroll
        for (int k = 0; k < GM; ++k)
            if (k < gn) *reinterpret_cast<f32x4 *>(rg + ((k * NW + wave) * 64 + lane) * 4) = accs[k];
        __syncthreads();
        if (BVC_FLOW_DIAG && g0 == 0) flow_stamp(c, hopid, 4);
        if (pub) {
            const unsigned ytile = (unsigned)((ce.g.mtile * ntiles + g.ntile) * 1024 + lane * 16);
            flow_publish<EPI, ADD, REARM_H, NW>(ce, hopid, rg + pk * (NW * 256), n0, ytile, ntiles, out, bias4, add4, mean4, std4, bitsv);
        }
    }
    if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 5);
}

// The GRU cell's reduction and epilogue for ONE chain (c.g.mtile / c.row / c.rowok / c.fr say which): gi / gh are this wave's partial
// gate sums; prefetch: request the next frame's first weights (wn) in front of the first barrier.
template <bool ENCODE, int PERN, int NW>
__device__ __forceinline__ void gru_epilogue(const FlowCtx &c, int hopid, int hb, int n0, unsigned ytile, const f32x4 (&gi)[3], const f32x4 (&gh)[3],
                                             const FlowLin nxt, f32x4 (&wn)[PERN], bool prefetch) {
    const FlowWg &g = c.g;
    const auto &a = *c.a;
    const int lane = g.lane, wave = g.wave;
    const long long H = (long long)hb * 16;
    const unsigned hbuf = (unsigned)(FB_H * 2) * c.sb;
    if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 3);
    // The cell's epilogue is spread over the waves (one wave doing all of it - 48 partial tiles to sum, two sigmoids and a tanh for
    // each of its four outputs per lane - took 1.5 us with seven waves idle): wave k < 6 sums quantity k (gi_r, gi_z, gi_n, gh_r,
    // gh_z, gh_n) over the eight waves and adds its bias, wave j < 4 then evaluates the cell for output j of every lane, wave 0
    // gathers the four and publishes.  Same operations in the same order per output as the one-wave form.
    // Their operands are requested now: they arrive while the other waves reach the barrier.
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, p4 = {0.f, 0.f, 0.f, 0.f};
    float hp = 0.0f;
    if (wave < 3) {
        if (ENCODE) b4 = *reinterpret_cast<const f32x4 *>(a.b_ih + wave * H + n0);
        if (!ENCODE && c.rowok) p4 = *reinterpret_cast<const f32x4 *>(a.part_gru + c.fr * 3 * H + wave * H + n0);
    } else if (wave < 6) {
        b4 = *reinterpret_cast<const f32x4 *>(a.b_hh + (wave - 3) * H + n0);
    }
    // this workgroup's own block of h(t): written by itself a frame ago (or the initial state); wave j takes element j of each lane
    if (wave < 4) hp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g.rs, hbuf + c.par * c.sb + ytile + (unsigned)wave * 4u, 0, AUX_SC1));
    if (prefetch) {    // the first layer of the next frame (after the last frame a harmless extra request: no value is carried across)
        const GPtr ub = uniform_ptr(nxt.w, ((size_t)g.ntile * nxt.wnb + wave * PERN) * g.wmul);
#pragma unroll
        for (int u = 0; u < PERN; ++u) wn[u] = wload(ub, (unsigned)lane * 16u, u);
    }
    float *red_gru = c.red_gru;
    float *red2 = c.red_lin + (c.hopctr & 1u) * (NW * 256);            // [6][256] sums | [256] outputs: the layer partials' free slot
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        *reinterpret_cast<f32x4 *>(red_gru + ((wave * 6 + q) * 64 + lane) * 4) = gi[q];
        *reinterpret_cast<f32x4 *>(red_gru + ((wave * 6 + 3 + q) * 64 + lane) * 4) = gh[q];
    }
    __syncthreads();
    if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 4);
    if (wave < 6) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(red_gru + (wave * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4 *>(red_gru + ((w * 6 + wave) * 64 + lane) * 4);
        v += b4;                                           // gi: (sum + b_ih) + the pre-computed phi_z part; gh: sum + b_hh
        if (wave < 3) v += p4;
        *reinterpret_cast<f32x4 *>(red2 + (wave * 64 + lane) * 4) = v;
    }
    __syncthreads();
    if (wave < 4) {
        float sgm[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) sgm[k] = red2[(k * 64 + lane) * 4 + wave];
        const float rg = sigmoid1(sgm[3] + sgm[0]);
        const float zg = sigmoid1(sgm[4] + sgm[1]);
        const float ng = tanhf(sgm[2] + rg * sgm[5]);
        red2[6 * 256 + lane * 4 + wave] = (hp - ng) * zg + ng;
    }
    __syncthreads();
    if (wave == 0) {
        const f32x4 hn = *reinterpret_cast<const f32x4 *>(red2 + 6 * 256 + lane * 4);
        if (a.all_h && c.rowok && c.t + 1 < a.T) *reinterpret_cast<f32x4 *>(a.all_h + (c.fr + 1) * H + n0) = hn;   // all_h[:, t+1], bvrnn.py:205
        // h(t)'s slot is NOT re-armed here: other workgroups may still be reading h(t) in their own GRU layer.  It is re-armed
        // by the second layer of frame t+1 (REARM_H), whose inputs prove that every workgroup has left frame t.
        __builtin_amdgcn_raw_buffer_store_b128(publishable(hn, c.rowok), g.rs, hbuf + (c.par ^ 1u) * c.sb + ytile, 0, AUX_SC1);
        flow_stamp(c, hopid, 1);
    }
    // red_gru is single-buffered, also when chains are interleaved (MULTI): its next writers are past this call's three barriers,
    // the partials' readers between the first and the second
}

// GRU cell (PyTorch gate order r, z, n; bvrnn.py:206,227): gh = W_hh h, gi = W_ih [phi_x_gen ; phi_z]; in decode the
// phi_z half of gi (+ b_ih) arrives pre-computed (a.part_gru).  Segments in the order their inputs become ready.
template <int PER, bool ENCODE, int PERN, bool FILL, int NW = 8>
__device__ __forceinline__ void flow_gru(FlowCtx &c, int hopid, int hb, const FlowLin nxt, f32x4 (&wn)[PERN], f32x4 (&gq)[GRU_EARLY_BLOCKS]) {
    const FlowWg &g = c.g;
    const auto &a = *c.a;
    if (g.ntile >= hb) {
        if (c.pre_now) {
#pragma unroll
            for (int u = 0; u < PERN; ++u) wn[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    int lane = g.lane;
    const int wave = g.wave;
    // (opaque per layer: what is derived from the lane - tile offsets, feature indices, their float forms - is then recomputed here
    // instead of being hoisted out of the frame loop for all fourteen layers and SPILLED; a spill reload in an epilogue waits, in
    // order, for every prefetch the wave has in flight)
    asm volatile("" : "+v"(lane));
    const unsigned code = (unsigned)((c.t << 4) | (unsigned)hopid) | 0x80000000u;
    const int n0 = g.ntile * 16 + (lane >> 4) * 4;
    const unsigned ytile = (unsigned)((g.mtile * hb + g.ntile) * 1024 + lane * 16);
    const unsigned hbuf = (unsigned)(FB_H * 2) * c.sb;
    flow_stamp(c, hopid, 0);
    f32x4 gi[3], gh[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        gi[q] = FILL ? c.fgi[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
        gh[q] = FILL ? c.fgh[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // h(t) and (encode) phi_z(z_t) are complete and were verified by this very wave earlier in the frame: their blocks are
    // requested at once and multiplied while phi_x(d_t), the input produced last, is still on its way
    // The weights stream through two register sets: the request for round i+1 is issued before round i is multiplied.
    if constexpr (FILL && PER == 8 && BVC_GRU_FAST) {
        // Four rounds of two k-blocks x three gates, in k order (the same sums as the generic form below).  Rounds 0 and 1 were requested
        // a layer ago (gq), round 2 has been in LDS since the launch began, round 3 is requested into the registers round 0 leaves.
        const unsigned l16 = (unsigned)lane * 16u;
        const GPtr ux = uniform_ptr(a.w_ihx, ((size_t)g.ntile * 2 * hb + wave * PER) * 3 * g.wmul);
        u32x4 xa[PER];
        gru_fetch_fresh<PER>(g, (unsigned)(FB_G3 * 2 + c.par) * c.sb, hb, xa, c.give_up, code);
        if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 2);        // (here: flags seen AND operands fetched)
        // (scheduling fences: hoisting the later rounds' loads above the earlier rounds' products would need registers that are not there)
        gru_round6<PER>(gq, xa, 0, gi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) gq[i] = wload(ux, l16, 18 + i);                       // round 3: k-blocks 6, 7
        __builtin_amdgcn_sched_barrier(0);
        gru_round6<PER>(gq + 6, xa, 2, gi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) gq[6 + i] = __builtin_bit_cast(f32x4, c.gpark[i * 64 + lane]);    // round 2: k-blocks 4, 5
        __builtin_amdgcn_sched_barrier(0);
        gru_round6<PER>(gq + 6, xa, 4, gi);
        gru_round6<PER>(gq, xa, 6, gi);
        __builtin_amdgcn_sched_barrier(0);
    } else if (!(PER == 1 && wave >= hb)) {                // (wave-uniform) a wave without a k-block of its own contributes zeros
        constexpr int HALF = FILL ? (PER >= BVC_GRU_ROUNDS_FILL ? PER / BVC_GRU_ROUNDS_FILL : 1) : (PER >= 4 ? PER / 4 : 1);       // k-blocks per round
        constexpr int RPS = PER / HALF;                    // rounds per segment
        constexpr int NSEG = FILL ? 1 : (ENCODE ? 3 : 2);  // FILL: the h and phi_z products were accumulated earlier in the frame
        constexpr int NR = NSEG * RPS;
        const int kb0 = wave * PER;
        const unsigned l16 = (unsigned)lane * 16u;
        const GPtr uh = uniform_ptr(a.w_hh, ((size_t)g.ntile * hb + kb0) * 3 * g.wmul);
        const GPtr uz = uniform_ptr(a.w_ihz, ((size_t)g.ntile * 2 * hb + kb0) * 3 * g.wmul);
        const GPtr ux = uniform_ptr(a.w_ihx, ((size_t)g.ntile * 2 * hb + kb0) * 3 * g.wmul);
        u32x4 xa[PER], xb[PER];
        // The weights stream through DEPTH register sets: the request for round i + DEPTH - 1 is issued before round i is multiplied.
        // The scheduling fences make that true: hipcc sinks a load to just in front of its first use when left alone (it schedules
        // for few live registers although this kernel may use all 256), and every round then waited a full cache round trip for
        // its own weights - the GRU hop measured 6 us against the 2.6 us its 192 MFMAs per SIMD take.
        constexpr int DEPTH = (FILL && BVC_GRU_DEPTH > 2) ? (BVC_GRU_DEPTH < NR ? BVC_GRU_DEPTH : NR) : 2;
        f32x4 wr[DEPTH][HALF][3];
        auto wsrc = [&](int i) { const int sn = i / RPS; return FILL ? ux : (sn == 0 ? uh : ((ENCODE && sn == 1) ? uz : ux)); };
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i)
            if (i < NR) gru_issue_w<HALF>(wsrc(i), l16, (i % RPS) * HALF, wr[i]);
        if (!FILL) gru_issue_known<PER>(g, hbuf + c.par * c.sb, hb, xa);
        if (!FILL && ENCODE) gru_issue_known<PER>(g, (unsigned)(FB_Q3 * 2 + c.par) * c.sb, hb, xb);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int sg = i / RPS, h0 = (i % RPS) * HALF;
            if (i + DEPTH - 1 < NR) gru_issue_w<HALF>(wsrc(i + DEPTH - 1), l16, ((i + DEPTH - 1) % RPS) * HALF, wr[(i + DEPTH - 1) % DEPTH]);
            if (BVC_GRU_FENCE) __builtin_amdgcn_sched_barrier(0);
            if (i == (NSEG - 1) * RPS) {                   // the last segment's input, phi_x(d_t): wait, fetch, verify.  (What was requested in front
                gru_fetch_fresh<PER>(g, (unsigned)(FB_G3 * 2 + c.par) * c.sb, hb, xa, c.give_up, code);     // of the poll travels during the wait.)
                if (BVC_FLOW_DIAG) flow_stamp(c, hopid, 2);        // (here: flags seen AND operands fetched)
            }
            if (FILL)                   gru_round<PER, HALF>(wr[i % DEPTH], xa, h0, gi);
            else if (sg == 0)           gru_round<PER, HALF>(wr[i % DEPTH], xa, h0, gh);
            else if (ENCODE && sg == 1) gru_round<PER, HALF>(wr[i % DEPTH], xb, h0, gi);
            else                        gru_round<PER, HALF>(wr[i % DEPTH], xa, h0, gi);
            if (BVC_GRU_FENCE) __builtin_amdgcn_sched_barrier(0);
        }
    }
    gru_epilogue<ENCODE, PERN, NW>(c, hopid, hb, n0, ytile, gi, gh, nxt, wn, c.pre_now && !(BVC_FLOW_PARKL && FILL && PER == 8));
}

// MULTI: the GRU cell for the chains of this workgroup, two chains at a time: a round's weights (two k-blocks x three gates, streamed
// through two register sets as in flow_gru) are multiplied with BOTH chains' operand blocks, so the 576 KiB (encode; decode 384) of
// gate weights per workgroup cross the compute unit's memory path once per pair instead of once per chain - they were a third of
// everything a workgroup requested per frame.  Per chain and wave the same products in the same order as flow_gru (segments h,
// phi_z, phi_x; k ascending), hence the same bits; the cells' epilogues run one after the other (gru_epilogue).
template <int PER, bool ENCODE, int PERN, int NW>
__device__ __forceinline__ void flow_gru_chains(FlowCtx &c, int hopid, int hb, const FlowLin nxt, f32x4 (&wn)[PERN], int mt0, int nch, long long T) {
    static_assert(PER == 8, "interleaved chains are built for h_dim 1024");
    FlowWg &g = c.g;
    const auto &a = *c.a;
    if (g.ntile >= hb) {
#pragma unroll
        for (int u = 0; u < PERN; ++u) wn[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    int lane = g.lane;
    const int wave = g.wave;
    asm volatile("" : "+v"(lane));                         // (see flow_layer)
    const unsigned code = (unsigned)((c.t << 4) | (unsigned)hopid) | 0x80000000u;
    const int n0 = g.ntile * 16 + (lane >> 4) * 4;
    const unsigned hbuf = (unsigned)(FB_H * 2) * c.sb;
    constexpr int HALF = 2, RPS = PER / HALF, NSEG = ENCODE ? 3 : 2, NR = NSEG * RPS;
    const int kb0 = wave * PER;
    const unsigned l16 = (unsigned)lane * 16u;
    flow_stamp(c, hopid, 0);
    for (int p0 = 0; p0 < nch; p0 += 2) {
        const bool two = p0 + 1 < nch;                     // (uniform) an odd last chain goes alone: its partner's products are skipped
        FlowWg gA = g, gB = g;
        gA.mtile = mt0 + p0;
        gB.mtile = mt0 + p0 + (two ? 1 : 0);
        f32x4 giA[3], ghA[3], giB[3], ghB[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) giA[q] = ghA[q] = giB[q] = ghB[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const GPtr uh = uniform_ptr(a.w_hh, ((size_t)g.ntile * hb + kb0) * 3 * g.wmul);
        const GPtr uz = uniform_ptr(a.w_ihz, ((size_t)g.ntile * 2 * hb + kb0) * 3 * g.wmul);
        const GPtr ux = uniform_ptr(a.w_ihx, ((size_t)g.ntile * 2 * hb + kb0) * 3 * g.wmul);
        u32x4 xA[PER], xB[PER];
        f32x4 wr[2][HALF][3];
        gru_issue_w<HALF>(uh, l16, 0, wr[0]);
        gru_issue_known<PER>(gA, hbuf + c.par * c.sb, hb, xA);      // h(t) of both chains: complete since the frame began
        gru_issue_known<PER>(gB, hbuf + c.par * c.sb, hb, xB);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int sg = i / RPS, h0 = (i % RPS) * HALF;
            if (i + 1 < NR) {
                const int sn = (i + 1) / RPS, hn0 = ((i + 1) % RPS) * HALF;
                gru_issue_w<HALF>(sn == 0 ? uh : ((ENCODE && sn == 1) ? uz : ux), l16, hn0, wr[(i + 1) & 1]);
            }
            // a chain's operand blocks of the NEXT segment are requested as soon as its last round of this one is multiplied: they
            // travel under the partner's round (phi_x(d_t), produced last, is waited for, fetched and verified; h and phi_z are known)
            const bool seg_end = (i % RPS == RPS - 1) && i + 1 < NR;
            const bool next_fresh = (i + 1) / RPS == NSEG - 1;
            if (sg == 0) gru_round<PER, HALF>(wr[i & 1], xA, h0, ghA);
            else         gru_round<PER, HALF>(wr[i & 1], xA, h0, giA);
            if (seg_end) {
                if (next_fresh) gru_fetch_fresh<PER>(gA, (unsigned)(FB_G3 * 2 + c.par) * c.sb, hb, xA, c.give_up, code);
                else            gru_issue_known<PER>(gA, (unsigned)(FB_Q3 * 2 + c.par) * c.sb, hb, xA);
            }
            if (two) {
                if (sg == 0) gru_round<PER, HALF>(wr[i & 1], xB, h0, ghB);
                else         gru_round<PER, HALF>(wr[i & 1], xB, h0, giB);
            }
            if (seg_end) {
                if (next_fresh) { if (two) gru_fetch_fresh<PER>(gB, (unsigned)(FB_G3 * 2 + c.par) * c.sb, hb, xB, c.give_up, code); }
                else            gru_issue_known<PER>(gB, (unsigned)(FB_Q3 * 2 + c.par) * c.sb, hb, xB);
            }
        }
        // the two cells' epilogues, one after the other (one set of partial tiles in LDS)
        for (int k = 0; k < (two ? 2 : 1); ++k) {
            FlowCtx ce = c;
            ce.g.mtile = mt0 + p0 + k;
            ce.row = ce.g.mtile * 16 + (lane & 15);
            ce.rowok = ce.row < a.B;
            ce.fr = (long long)ce.row * T + c.t;
            ce.probe = c.probe && p0 == 0 && k == 0;
            const unsigned ytile = (unsigned)((ce.g.mtile * hb + g.ntile) * 1024 + lane * 16);
            const bool lastc = p0 + k == nch - 1;
            if (k == 0) gru_epilogue<ENCODE, PERN, NW>(ce, hopid, hb, n0, ytile, giA, ghA, nxt, wn, lastc);
            else        gru_epilogue<ENCODE, PERN, NW>(ce, hopid, hb, n0, ytile, giB, ghB, nxt, wn, lastc);
        }
    }
}

}  // namespace

// PERH: k-blocks per wave of an h_dim-sized operand (h_dim = 128 * PERH, or h_dim <= 128 for PERH = 1: then a wave owns at
// most one k-block); the z_dim- and num_mels-sized operands (<= 128) always have one k-block per wave.
#ifndef BVC_FLOW_WAVES_PER_SIMD
#define BVC_FLOW_WAVES_PER_SIMD 2
#endif
// MULTI: one layer of the static program, chain by chain
#define FLOW_EACH_CHAIN(...)                                                                                            \
    for (int ci_ = 0; ci_ < nch; ++ci_) {                                                                               \
        c.g.mtile = mt0 + ci_;                                                                                          \
        c.row = c.g.mtile * 16 + (c.g.lane & 15);                                                                       \
        c.rowok = c.row < a.B;                                                                                          \
        c.fr = (long long)c.row * T + t;                                                                                \
        c.pre_now = ci_ == nch - 1;                                                                                     \
        __VA_ARGS__;                                                                                                    \
    }

// MULTI: a workgroup owns its feature tile for MG utterance groups ("chains": utterances never interact) and works through a
// layer chain by chain with the same weight registers; by the time it returns to a chain for the next layer, that chain's
// inputs - produced by the other workgroups in the same order - have long arrived, so the hand-off latency of one chain lies
// under the products of the others.  Batches of more than 16 * (CUs / feature tiles) utterances (64 at h_dim 1024) run this way.
// FOLD: 1 / 0: the folded hop (FlowArgs::pxc) is / is not compiled in; -1: both, chosen at run time.  (The interleaved-chain
// kernels with both programs in them spill: they are instantiated once per program.)
template <int PERH, bool ENCODE, bool FILL, int NW = 8, bool MULTI = false, int FOLD = -1>
__global__ __launch_bounds__(NW * 64, NW == 8 ? BVC_FLOW_WAVES_PER_SIMD : 1) void bvrnn_flow_kernel(const FlowArgs *a0) {
    static_assert(!(MULTI && FILL), "the filler quanta are per chain: not built for interleaved chains");
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][8][256] layer partials | [1 or 2][8][6][256] GRU partials
    const int tid = threadIdx.x;
    FlowCtx c;
    FlowArgsC ap = (FlowArgsC)(unsigned long long)a0;      // device-resident copy of the arguments (flow_set_args_kernel)
    c.a = ap;
    // LDS: [2][NW][256] layer partials | [NW][6][256] GRU partials | MULTI: [2][4][NW][256] partials of a group of chains.  Filler kernels: the waves' operand stashes
    // (NW x 8 KiB) lie over the GRU partials - the stashes are dead from the last quantum (layer 9) to the next frame's first layer,
    // whose operand, h(t+1), exists only after wave 0 has read the partials -, then the parked weights (NW x 8 KiB: the first layer's, BVC_FLOW_PARKL;
    // or a quarter of the GRU layer's, BVC_GRU_FAST), then the flag.
    c.red_lin = lds;
    c.red_gru = lds + 2 * NW * 256;
    c.red_chain = lds + 2 * NW * 256 + NW * 6 * 256;      // (MULTI only: FLOW_LDS_MULTI)
    c.stash = (LdsX)(lds + 2 * NW * 256) + (tid >> 6) * (PERH * 64);
    c.gpark = (LdsX)(lds + 2 * NW * 256 + NW * 8 * 256) + (tid >> 6) * (6 * 64);
    c.lpark = (LdsX)(lds + 2 * NW * 256 + NW * 8 * 256) + (tid >> 6) * (8 * 64);
    static_assert(!(BVC_FLOW_PARKL && BVC_GRU_FAST), "one use of the parking region");
    c.pubflag = (volatile unsigned __attribute__((address_space(3))) *)(lds + 2 * NW * 256 + NW * 8 * 256 + NW * 8 * 256);
    if (FILL && tid == 0) c.pubflag[0] = 0xFFFFFFFFu;
    c.g.lane = tid & 63;
    c.g.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int MT = ap->MT;
    const int MG = MULTI ? ap->MG : 1;                     // chains per workgroup
    const int MTG = MULTI ? (MT + MG - 1) / MG : MT;       // workgroups per feature tile
    c.g.ntile = (slot / MTG) * 8 + xcd;                   // the utterance groups of a feature tile share an XCD (weights cross the fabric once)
    const int mt0 = (slot % MTG) * MG;
    const int nch = MULTI ? (MT - mt0 < MG ? MT - mt0 : MG) : 1;
    c.g.mtile = mt0;
    if (ap->dbg_withhold && bid == 0) return;              // tests: a workgroup that never publishes (its consumers time out)
    if (c.g.ntile >= ap->NTG) return;                      // grid is rounded up to a multiple of 8 feature tiles
    c.g.rs = __builtin_amdgcn_make_buffer_rsrc(ap->flow, 0, (int)(FB_COUNT * 2u * ap->slot_bytes), 0x00020000);
    c.sb = ap->slot_bytes;
    c.g.spin_limit = ap->spin_limit;
    c.g.wmul = ap->dbg_hot_w ? 0 : 1;
    c.g.status = ap->status;
    c.row = c.g.mtile * 16 + (c.g.lane & 15);
    c.rowok = c.row < ap->B;
    c.give_up = false;
    c.pre_now = true;
    c.probe = ap->probe != nullptr && bid == ap->probe_wg && tid == ap->probe_wave * 64;
    c.hopctr = 0;
    const int hb = ap->hb, zb = ap->zb, xb = ap->xb;
    const long long T = ap->T;
    const bool hfull = c.g.ntile < hb;                     // this workgroup owns a tile of the h_dim-wide layers
    f32x4 wa[PERH], wb[PERH], w1[1];
    f32x4 gq[GRU_EARLY_BLOCKS];                            // BVC_GRU_FAST: the GRU layer's early-requested weights (rounds 0, 1)
    constexpr bool GFAST = FILL && PERH == 8 && BVC_GRU_FAST && !MULTI;
    constexpr bool GPRE = GFAST;
    if (GFAST && hfull) {                                  // park round 2 (k-blocks 4, 5 x three gates) of this wave's share of W_ih[:, :H]
        const GPtr ux = uniform_ptr(ap->w_ihx, (((size_t)c.g.ntile * 2 * hb + c.g.wave * PERH) * 3 + 12) * c.g.wmul);
#pragma unroll
        for (int i = 0; i < 6; ++i) c.gpark[i * 64 + c.g.lane] = __builtin_bit_cast(u32x4, wload(ux, (unsigned)c.g.lane * 16u, i));
    }
    // BVC_FLOW_PARKV: wide layers whose weights - this wave's eight blocks each - stay in registers for the whole launch.  Layers that
    // carry a filler quantum first (their waves request two weight sets in front of the reduction barrier otherwise): dec.2, dec.4; decode has
    // registers for a third: phi_x.4
    constexpr bool PK = BVC_FLOW_PARKV && FILL && !MULTI;
    constexpr bool PL = BVC_FLOW_PARKL && FILL && !MULTI && PERH == 8;
    constexpr bool P8 = PK && BVC_FLOW_PARK_SET == 1, P9 = PK && (BVC_FLOW_PARK_SET == 1 || (!ENCODE && BVC_FLOW_PARKV >= 2)),
                   P12 = PK && BVC_FLOW_PARK_SET == 0, P13 = PK && (BVC_FLOW_PARK_SET == 0 || (!ENCODE && BVC_FLOW_PARKV >= 2));
    f32x4 k8[P8 ? PERH : 1], k9[P9 ? PERH : 1], k12[P12 ? PERH : 1], k13[P13 ? PERH : 1];
    {
        auto park = [&](auto &k, const FlowLin l) {
            const GPtr u = uniform_ptr(l.w, hfull ? (size_t)c.g.ntile * l.wnb + c.g.wave * PERH : 0);
#pragma unroll
            for (int i = 0; i < PERH; ++i) k[i] = wload(u, (unsigned)c.g.lane * 16u, i);
        };
        if constexpr (P8) park(k8, L(ap->dec1));
        if constexpr (P9) park(k9, L(ap->dec2));
        if constexpr (P12) park(k12, L(ap->px1));
        if constexpr (P13) park(k13, L(ap->px2));
    }
    {
        const FlowLin first = ENCODE ? L(ap->enc0h) : L(ap->dec0h);
        const GPtr ub = uniform_ptr(first.w, hfull ? (size_t)c.g.ntile * first.wnb + c.g.wave * PERH : 0);
#pragma unroll
        for (int u = 0; u < PERH; ++u) wa[u] = wload(ub, (unsigned)c.g.lane * 16u, u);
        if constexpr (PL) {
#pragma unroll
            for (int u = 0; u < PERH; ++u) c.lpark[u * 64 + c.g.lane] = __builtin_bit_cast(u32x4, wa[u]);      // (read back by this very wave only)
        }
    }
    for (long long t = 0; t < T; ++t) {
        if constexpr (PL) {                                // the first layer's weights come out of LDS (requested by nobody)
#pragma unroll
            for (int u = 0; u < PERH; ++u) wa[u] = __builtin_bit_cast(f32x4, c.lpark[u * 64 + c.g.lane]);
        }
        asm volatile("" : "+s"(ap));                       // see FlowArgsC
        c.a = ap;
        const auto &a = *ap;
        c.t = t;
        c.par = (unsigned)(t & 1);
        c.fr = (long long)c.row * T + t;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        if (FILL) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { c.fgh[q] = zero4; c.fgi[q] = zero4; }
            c.fd0 = zero4;
        }
        const unsigned hsrc = (unsigned)(FB_H * 2 + c.par) * c.sb, zsrc = (unsigned)(FB_Q3 * 2 + c.par) * c.sb;
        // FILL: behind a layer, while its successors' inputs are still being produced, a wave multiplies one "quantum" of a product
        // whose input has been complete since the start of the frame (h) or since phi_z (encode): the three gates of W_hh h, the
        // h half of dec.0, the three gates of W_ih[:, H:] phi_z.  The GRU layer then only has phi_x(d_t)'s third left.
        constexpr int G0 = FILL ? 0 : -2, G1 = FILL ? 1 : -2, G2 = FILL ? 2 : -2, GP = FILL ? -1 : -2;
        const FlowFill f_hh = {a.w_hh, hb, hb, hsrc}, f_d0 = {a.dec0h.w, a.dec0h.wnb, hb, hsrc}, f_iz = {a.w_ihz, 2 * hb, hb, zsrc};
        const bool folded = FOLD < 0 ? a.pxc.w != nullptr : FOLD == 1;      // (uniform) dec.6 -> norm -> phi_x.0 folded into one layer (FlowArgs::pxc)
        if constexpr (!MULTI) {
            // (wK: the weights of wide layer K - prefetched into wa / wb by the layer in front of it, or resident in registers: P8 .. P13)
            auto &w8 = pick<P8>(k8, wb);  auto &w9 = pick<P9>(k9, wa);  auto &w12 = pick<P12>(k12, wa);  auto &w13 = pick<P13>(k13, wb);
            if (ENCODE) {
                //         PER   epilogue  two    add    pre_in pre_out       rearm  filler
                flow_layer<PERH, FE_ELU, false, true, true, true, PERH, false, G0, NW>(c, 1, L(a.enc0h), FB_H, L(a.enc0h), 0, hb, hb, FB_E1, wa, L(a.enc1), wb, gq, zero4, f_hh, &c.fgh[0]);
                flow_layer<PERH, FE_ELU, false, false, true, false, PERH, true, G1, NW>(c, 2, L(a.enc1), FB_E1, L(a.enc1), 0, hb, hb, FB_E2, wb, L(a.enc1), wa, gq, zero4, f_hh, &c.fgh[1]);
                flow_layer<PERH, FE_CODE, false, false, false, false, PERH, false, G2, NW>(c, 3, L(a.enc2), FB_E2, L(a.enc2), 0, hb, zb, FB_ZC, wa, L(a.enc2), wb, gq, zero4, f_hh, &c.fgh[2]);
                flow_layer<1, FE_ELU, false, false, false, true, PERH, false, GP, NW>(c, 4, L(a.pz0), FB_ZC, L(a.pz0), 0, zb, hb, FB_Q1, w1, L(a.pz1), wa, gq, zero4, f_d0, &c.fd0);
                flow_layer<PERH, FE_ELU, false, false, true, true, PERH, false, -2, NW>(c, 5, L(a.pz1), FB_Q1, L(a.pz1), 0, hb, hb, FB_Q2, wa, L(a.pz2), wb, gq);
                if (FILL) {
                    FlowLin d0 = L(a.dec0z);                   // dec.0: only the phi_z half is left; the bias travels with dec0h
                    d0.bias = a.dec0h.bias;
                    flow_layer<PERH, FE_ELU, false, false, true, true, PERH, false, -2, NW>(c, 6, L(a.pz2), FB_Q2, L(a.pz2), 0, hb, hb, FB_Q3, wb, d0, wa, gq);
                    flow_layer<PERH, FE_ELU, false, false, true, !P8, PERH, false, G0, NW>(c, 7, d0, FB_Q3, d0, 0, hb, hb, FB_D1, wa, L(a.dec1), wb, gq, c.fd0, f_iz, &c.fgi[0]);
                } else {
                    flow_layer<PERH, FE_ELU, false, false, true, true, PERH, false, -2, NW>(c, 6, L(a.pz2), FB_Q2, L(a.pz2), 0, hb, hb, FB_Q3, wb, L(a.dec0h), wa, gq);
                    flow_layer<PERH, FE_ELU, true, false, true, true, PERH, false, -2, NW>(c, 7, L(a.dec0h), FB_H, L(a.dec0z), FB_Q3, hb, hb, FB_D1, wa, L(a.dec1), wb, gq);
                }
                flow_layer<PERH, FE_ELU, false, false, true, !P9, PERH, false, G1, NW>(c, 8, L(a.dec1), FB_D1, L(a.dec1), 0, hb, hb, FB_D2, w8, L(a.dec2), wa, gq, zero4, f_iz, &c.fgi[1]);
                if (folded) {
                    flow_layer<PERH, FE_ELU_KEEP, false, false, true, true, PERH, false, G2, NW>(c, 9, L(a.dec2), FB_D2, L(a.dec2), 0, hb, hb, FB_D3, w9, L(a.pxc), wb, gq, zero4, f_iz, &c.fgi[2]);
                    flow_layer<PERH, FE_ELU, false, false, true, !P12, PERH, false, -2, NW>(c, 11, L(a.pxc), FB_D3, L(a.pxc), 0, hb, hb, FB_G1, wb, L(a.px1), wa, gq);
                } else {
                    flow_layer<PERH, FE_ELU, false, false, true, false, PERH, false, G2, NW>(c, 9, L(a.dec2), FB_D2, L(a.dec2), 0, hb, hb, FB_D3, w9, L(a.dec2), wb, gq, zero4, f_iz, &c.fgi[2]);
                    flow_layer<PERH, FE_MEL, false, false, false, false, PERH, false, -2, NW>(c, 10, L(a.dec3), FB_D3, L(a.dec3), 0, hb, xb, FB_DN, wa, L(a.dec3), wb, gq);
                }
            } else {
                flow_layer<PERH, FE_ELU, false, true, true, !P8, PERH, false, G0, NW>(c, 7, L(a.dec0h), FB_H, L(a.dec0h), 0, hb, hb, FB_D1, wa, L(a.dec1), wb, gq, zero4, f_hh, &c.fgh[0]);
                flow_layer<PERH, FE_ELU, false, false, true, !P9, PERH, true, G1, NW>(c, 8, L(a.dec1), FB_D1, L(a.dec1), 0, hb, hb, FB_D2, w8, L(a.dec2), wa, gq, zero4, f_hh, &c.fgh[1]);
                if (folded) {                              // dec.4 (its output is kept for all frames) -> phi_x.0 o norm o dec.6 as one wide layer
                    flow_layer<PERH, FE_ELU_KEEP, false, false, true, true, PERH, false, G2, NW>(c, 9, L(a.dec2), FB_D2, L(a.dec2), 0, hb, hb, FB_D3, w9, L(a.pxc), wb, gq, zero4, f_hh, &c.fgh[2]);
                    flow_layer<PERH, FE_ELU, false, false, true, !P12, PERH, false, -2, NW>(c, 11, L(a.pxc), FB_D3, L(a.pxc), 0, hb, hb, FB_G1, wb, L(a.px1), wa, gq);
                } else {
                    flow_layer<PERH, FE_ELU, false, false, true, false, PERH, false, G2, NW>(c, 9, L(a.dec2), FB_D2, L(a.dec2), 0, hb, hb, FB_D3, w9, L(a.dec2), wb, gq, zero4, f_hh, &c.fgh[2]);
                    flow_layer<PERH, FE_MEL, false, false, false, false, PERH, false, -2, NW>(c, 10, L(a.dec3), FB_D3, L(a.dec3), 0, hb, xb, FB_DN, wa, L(a.dec3), wb, gq);
                }
            }
            if (!folded)
                flow_layer<1, FE_ELU, false, false, false, !P12, PERH, false, -2, NW>(c, 11, L(a.px0), FB_DN, L(a.px0), 0, xb, hb, FB_G1, w1, L(a.px1), wa, gq);
            flow_layer<PERH, FE_ELU, false, false, true, !P13, PERH, false, -2, NW>(c, 12, L(a.px1), FB_G1, L(a.px1), 0, hb, hb, FB_G2, w12, L(a.px2), wb, gq);
            flow_layer<PERH, FE_ELU, false, false, true, false, PERH, false, -2, NW, GPRE>(c, 13, L(a.px2), FB_G2, L(a.px2), 0, hb, hb, FB_G3, w13, L(a.px2), wa, gq);
            flow_gru<PERH, ENCODE, PERH, FILL, NW>(c, 14, hb, ENCODE ? L(a.enc0h) : L(a.dec0h), wa, gq);
        } else {
            // the same program on interleaved chains (no filler quanta): wide single-segment layers pipeline their chains
            // (flow_layer_chains), the two narrow-input layers, the two-segment dec.0 of encode and the GRU go chain by chain
            constexpr bool E = ENCODE;
            if (E) {
                flow_layer_chains<PERH, FE_ELU, true, true, true, PERH, false, NW>(c, 1, L(a.enc0h), FB_H, hb, hb, FB_E1, wa, L(a.enc1), wb, mt0, nch, T);
                flow_layer_chains<PERH, FE_ELU, false, true, false, PERH, true, NW>(c, 2, L(a.enc1), FB_E1, hb, hb, FB_E2, wb, L(a.enc1), wa, mt0, nch, T);
                flow_layer_chains<PERH, FE_CODE, false, false, false, PERH, false, NW>(c, 3, L(a.enc2), FB_E2, hb, zb, FB_ZC, wa, L(a.enc2), wb, mt0, nch, T);
                FLOW_EACH_CHAIN(flow_layer<1, FE_ELU, false, false, false, true, PERH, false, -2, NW>(c, 4, L(a.pz0), FB_ZC, L(a.pz0), 0, zb, hb, FB_Q1, w1, L(a.pz1), wa, gq));
                flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, false, NW>(c, 5, L(a.pz1), FB_Q1, hb, hb, FB_Q2, wa, L(a.pz2), wb, mt0, nch, T);
                flow_layer_chains<PERH, FE_ELU, false, true, false, PERH, false, NW>(c, 6, L(a.pz2), FB_Q2, hb, hb, FB_Q3, wb, L(a.pz2), wa, mt0, nch, T);
                // two segments share the weight registers: the first segment's weights are fetched per chain
                {                                          // dec.0 = dec0h . h + dec0z . phi_z: both segments for a group's chains, then one reduction
                    const GPtr ub = uniform_ptr(a.dec0z.w, ((size_t)c.g.ntile * a.dec0z.wnb + c.g.wave * PERH) * c.g.wmul);
#pragma unroll
                    for (int u = 0; u < PERH; ++u) wa[u] = wload(ub, (unsigned)c.g.lane * 16u, u);
                    FlowLin d0 = L(a.dec0z);
                    d0.bias = a.dec0h.bias;
                    flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, false, NW, true>(c, 7, d0, FB_Q3, hb, hb, FB_D1, wa, L(a.dec1), wb, mt0, nch, T, L(a.dec0h), FB_H);
                }
                flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, false, NW>(c, 8, L(a.dec1), FB_D1, hb, hb, FB_D2, wb, L(a.dec2), wa, mt0, nch, T);
                if (folded) {
                    flow_layer_chains<PERH, FE_ELU_KEEP, false, true, true, PERH, false, NW>(c, 9, L(a.dec2), FB_D2, hb, hb, FB_D3, wa, L(a.pxc), wb, mt0, nch, T);
                    flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, false, NW>(c, 11, L(a.pxc), FB_D3, hb, hb, FB_G1, wb, L(a.px1), wa, mt0, nch, T);
                } else {
                    flow_layer_chains<PERH, FE_ELU, false, true, false, PERH, false, NW>(c, 9, L(a.dec2), FB_D2, hb, hb, FB_D3, wa, L(a.dec2), wb, mt0, nch, T);
                }
            } else {
                flow_layer_chains<PERH, FE_ELU, true, true, true, PERH, false, NW>(c, 7, L(a.dec0h), FB_H, hb, hb, FB_D1, wa, L(a.dec1), wb, mt0, nch, T);
                flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, true, NW>(c, 8, L(a.dec1), FB_D1, hb, hb, FB_D2, wb, L(a.dec2), wa, mt0, nch, T);
                if (folded) {
                    flow_layer_chains<PERH, FE_ELU_KEEP, false, true, true, PERH, false, NW>(c, 9, L(a.dec2), FB_D2, hb, hb, FB_D3, wa, L(a.pxc), wb, mt0, nch, T);
                    flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, false, NW>(c, 11, L(a.pxc), FB_D3, hb, hb, FB_G1, wb, L(a.px1), wa, mt0, nch, T);
                } else {
                    flow_layer_chains<PERH, FE_ELU, false, true, false, PERH, false, NW>(c, 9, L(a.dec2), FB_D2, hb, hb, FB_D3, wa, L(a.dec2), wb, mt0, nch, T);
                }
            }
            if (!folded) {
                flow_layer_chains<PERH, FE_MEL, false, false, false, PERH, false, NW>(c, 10, L(a.dec3), FB_D3, hb, xb, FB_DN, wa, L(a.dec3), wb, mt0, nch, T);
                FLOW_EACH_CHAIN(flow_layer<1, FE_ELU, false, false, false, true, PERH, false, -2, NW>(c, 11, L(a.px0), FB_DN, L(a.px0), 0, xb, hb, FB_G1, w1, L(a.px1), wa, gq));
            }
            flow_layer_chains<PERH, FE_ELU, false, true, true, PERH, false, NW>(c, 12, L(a.px1), FB_G1, hb, hb, FB_G2, wa, L(a.px2), wb, mt0, nch, T);
            flow_layer_chains<PERH, FE_ELU, false, true, false, PERH, false, NW>(c, 13, L(a.px2), FB_G2, hb, hb, FB_G3, wb, L(a.px2), wa, mt0, nch, T);
            flow_gru_chains<PERH, ENCODE, PERH, NW>(c, 14, hb, ENCODE ? L(a.enc0h) : L(a.dec0h), wa, mt0, nch, T);
        }
    }
}

// Residency census (bvc_model_create): are `gridDim.x` workgroups with the recurrence kernels' footprint - 512 threads, every
// VGPR of the compute unit, the filler kernels' LDS - resident at once?  Every workgroup adds itself to ctr[0] and waits
// (bounded) until all have; ctr[1] counts those that gave up.  A kernel of its own, so that no profile mixes it with the
// recurrence launches.
__global__ __launch_bounds__(512, 2) void flow_census_kernel(unsigned *ctr, unsigned spin_limit) {
    asm volatile("; the recurrence kernels' register footprint" ::: "v255");
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > spin_limit) {                    // bounded: a workgroup queued behind a resident one never lets the count complete
                __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
}

__global__ void fill_u32_kernel(unsigned *p, unsigned v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// Everything a persistent launch needs in place, as ONE kernel: the flow region filled with the sentinel, its first buffer - h of frame -1:
// n_h0 = mt16 * H floats in operand-fragment order - set to the caller's initial state (natural (B, H) rows; null: zero; padding rows
// zero), and the device copy of the arguments.
__global__ __launch_bounds__(256) void flow_prepare_kernel(FlowArgs *dst, FlowArgs v, unsigned *flow, long long n_flow, long long n_h0,
                                                           const float *__restrict__ h0, int B, int H) {
    if (blockIdx.x == 0) {
        const unsigned *src = reinterpret_cast<const unsigned *>(&v);
        unsigned *d = reinterpret_cast<unsigned *>(dst);
        for (unsigned i = threadIdx.x; i < sizeof(FlowArgs) / 4; i += blockDim.x) d[i] = src[i];
    }
    const int ntile = H >> 4;
    uint4 *flow4 = reinterpret_cast<uint4 *>(flow);
    const long long n4 = n_flow >> 2, h4 = n_h0 >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        uint4 o = {FLOW_POISON, FLOW_POISON, FLOW_POISON, FLOW_POISON};
        if (i < h4) {                                      // 16 bytes of a fragment-packed block: columns n .. n + 3 of row m
            const long long tile = i >> 6;
            const int lane = (int)(i & 63);
            const int m = (int)(tile / ntile) * 16 + (lane & 15), n = (int)(tile % ntile) * 16 + (lane >> 4) * 4;
            float4 hv = {0.f, 0.f, 0.f, 0.f};
            if (h0 && m < B) hv = *reinterpret_cast<const float4 *>(h0 + (long long)m * H + n);
            o = {__float_as_uint(hv.x), __float_as_uint(hv.y), __float_as_uint(hv.z), __float_as_uint(hv.w)};
        }
        flow4[i] = o;
    }
}

int launch_flow_prepare(const FlowArgs &a, FlowArgs *d_args, unsigned *flow, long long n_flow, long long n_h0, const float *d_h0, int B, int H,
                        hipStream_t s) {
    if (n_flow % 4 || n_h0 % 4 || H % 16 || n_h0 > n_flow || (d_h0 && (reinterpret_cast<uintptr_t>(d_h0) & 15))) {
        set_error("flow_prepare: unaligned flow region or initial state");
        return BVC_EINVAL;
    }
    const long long n4 = n_flow / 4;
    const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(flow_prepare_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, d_args, a, flow, n_flow, n_h0, d_h0, B, H);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

int launch_fill_u32(unsigned *p, unsigned v, long long n, hipStream_t s) {
    if (n <= 0) return BVC_OK;
    const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(grid), dim3(256), 0, s, p, v, n);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

constexpr size_t flow_lds(int nw) { return (size_t)(2 * nw * 256 + nw * 6 * 256) * sizeof(float); }      // 64 KiB with 8 waves
constexpr size_t FLOW_LDS = flow_lds(8);
constexpr size_t FLOW_LDS_MULTI = FLOW_LDS + (size_t)2 * 4 * 8 * 256 * sizeof(float);  // + two slots of four chains' partial tiles: 128 KiB
constexpr size_t FLOW_LDS_FILL = (size_t)(2 * 8 * 256 + 8 * 8 * 256 + 8 * 8 * 256) * sizeof(float) + 16;       // partials | stashes (over the GRU partials) | parked weights | flag: 144 KiB

template <int PERH, bool ENC>
static int flow_attr() {
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bvrnn_flow_kernel<PERH, ENC, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS));
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bvrnn_flow_kernel<PERH, ENC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_FILL));
    return BVC_OK;
}

int launch_flow_census(unsigned *ctr, int grid, unsigned spin_limit, hipStream_t s) {
    hipLaunchKernelGGL(flow_census_kernel, dim3(grid), dim3(512), FLOW_LDS_FILL, s, ctr, spin_limit);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

int flow_kernels_init() {
    int rc;
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(flow_census_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_FILL));
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bvrnn_flow_kernel<8, true, false, 8, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_MULTI));
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bvrnn_flow_kernel<8, true, false, 8, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_MULTI));
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bvrnn_flow_kernel<8, false, false, 8, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_MULTI));
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bvrnn_flow_kernel<8, false, false, 8, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_LDS_MULTI));

    if ((rc = flow_attr<1, true>()) || (rc = flow_attr<1, false>()) || (rc = flow_attr<2, true>()) || (rc = flow_attr<2, false>()) ||
        (rc = flow_attr<4, true>()) || (rc = flow_attr<4, false>()) || (rc = flow_attr<8, true>()) || (rc = flow_attr<8, false>())) return rc;
    return BVC_OK;
}

// k-blocks per wave of an h_dim-sized segment, or 0 if this h_dim is not laid out for the persistent kernel
int flow_perh(int h_dim) {
    if (h_dim % 16) return 0;
    if (h_dim == 128) return 1;
    if (h_dim == 256) return 2;
    if (h_dim == 512) return 4;
    if (h_dim == 1024) return 8;
    if (h_dim < 128) return 1;
    return 0;
}

__global__ void flow_set_args_kernel(FlowArgs *dst, FlowArgs v) {
    const unsigned *src = reinterpret_cast<const unsigned *>(&v);
    unsigned *d = reinterpret_cast<unsigned *>(dst);
    for (unsigned i = threadIdx.x; i < sizeof(FlowArgs) / 4; i += blockDim.x) d[i] = src[i];
}

template <int PERH>
static void flow_launch_t(const FlowArgs *d_a, bool encode, bool fill, int grid, hipStream_t s) {
    if (encode && fill)  hipLaunchKernelGGL((bvrnn_flow_kernel<PERH, true, true>), dim3(grid), dim3(512), FLOW_LDS_FILL, s, d_a);
    else if (encode)     hipLaunchKernelGGL((bvrnn_flow_kernel<PERH, true, false>), dim3(grid), dim3(512), FLOW_LDS, s, d_a);
    else if (fill)       hipLaunchKernelGGL((bvrnn_flow_kernel<PERH, false, true>), dim3(grid), dim3(512), FLOW_LDS_FILL, s, d_a);
    else                 hipLaunchKernelGGL((bvrnn_flow_kernel<PERH, false, false>), dim3(grid), dim3(512), FLOW_LDS, s, d_a);
}

// d_args: device memory for a copy of `a` (read by the kernel through the scalar cache); it must stay untouched until the
// launch has finished.
// (A 4-wave form - bvrnn_flow_kernel<16, ., false, 4>: one wave per SIMD, 214 VGPRs, so that the vocoder of the same or of
// another call could share the CUs with the recurrence - was built and measured: correct, but 27 % slower per frame, and with
// four batches in flight 5,120 audio-s/s against 5,700.  NW stays a template parameter; only the 8-wave form is instantiated.)
int launch_flow(const FlowArgs &a, FlowArgs *d_args, int perh, bool encode, bool fill, hipStream_t s, bool args_resident) {
    static_assert(sizeof(FlowArgs) % 4 == 0, "FlowArgs is copied in dwords");
    if (!args_resident) hipLaunchKernelGGL(flow_set_args_kernel, dim3(1), dim3(256), 0, s, d_args, a);
    const FlowArgs *d_a = d_args;
    if (a.MG > 1) {                                        // interleaved chains: more utterance groups than workgroup slots per feature tile
        if (perh != 8) { set_error("launch_flow: interleaved chains are built for h_dim 1024 only"); return BVC_EINVAL; }
        const int grid_m = ((a.NTG + 7) / 8) * 8 * ((a.MT + a.MG - 1) / a.MG);
        if (encode && a.pxc.w) hipLaunchKernelGGL((bvrnn_flow_kernel<8, true, false, 8, true, 1>), dim3(grid_m), dim3(512), FLOW_LDS_MULTI, s, d_a);
        else if (encode) hipLaunchKernelGGL((bvrnn_flow_kernel<8, true, false, 8, true, 0>), dim3(grid_m), dim3(512), FLOW_LDS_MULTI, s, d_a);
        else if (a.pxc.w) hipLaunchKernelGGL((bvrnn_flow_kernel<8, false, false, 8, true, 1>), dim3(grid_m), dim3(512), FLOW_LDS_MULTI, s, d_a);
        else        hipLaunchKernelGGL((bvrnn_flow_kernel<8, false, false, 8, true, 0>), dim3(grid_m), dim3(512), FLOW_LDS_MULTI, s, d_a);
        BVC_HIP_TRY(hipGetLastError());
        return BVC_OK;
    }
    const int grid = ((a.NTG + 7) / 8) * 8 * a.MT;

    switch (perh) {
        case 1: flow_launch_t<1>(d_a, encode, fill, grid, s); break;
        case 2: flow_launch_t<2>(d_a, encode, fill, grid, s); break;
        case 4: flow_launch_t<4>(d_a, encode, fill, grid, s); break;
        case 8: flow_launch_t<8>(d_a, encode, fill, grid, s); break;
        default: set_error("launch_flow: unsupported blocks per wave %d", perh); return BVC_EINVAL;
    }
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
