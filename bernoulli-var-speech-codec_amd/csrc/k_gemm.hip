// BVRNN GEMM kernels for gfx950 (fp32-in / fp32-accumulate MFMA, exact fp32 products).
//
//  gemm_skinny_kernel  - one recurrent-step layer: y[M,N] = epi( [x1|x2|x3] @ W^T + b ), M = batch
//                        (<= a few hundred rows).  One workgroup per 16x16 output tile, the K
//                        dimension is split over the 8 waves of the workgroup (each wave streams
//                        its slice of the activation rows and weight rows straight from L2/MALL
//                        into MFMA operand registers - there is no reuse inside a workgroup, so no
//                        LDS staging), partial tiles are summed in fixed order through LDS, and the
//                        layer's epilogue (bias, ELU, sigmoid/round/bit-mask, mel normalisation,
//                        GRU cell) is applied by the first 256 threads.  Workgroups that share
//                        weight rows (the batch tiles of one feature tile) are placed on the same
//                        XCD so each weight row crosses the fabric once per step.
//  gemm_batched_kernel - phi_x over all frames (bvrnn.py:178): M = B*T rows, 128x128 workgroup
//                        tile, 64x64 per wave (4x4 MFMA tiles), operands loaded as k-contiguous
//                        float4 fragments.
//
// ONE order of summation per output, whichever kernel computes a layer (the persistent recurrence kernel of k_flow.hip, the
// launch-per-layer kernel below, the batched kernels): the K/16 k-blocks of a segment are cut into NCHUNK = 8 chunks - chunk c =
// blocks [nb*c/8, nb*(c+1)/8) -, a chunk is a chain of MFMAs over its blocks in k order starting from zero, the chunks are added
// in ascending order (the first one is copied), then the bias, then an optional pre-computed addend, then the activation.  In the
// recurrent kernels a chunk is a wave's share of the split K; the batched kernels run the chunks one after the other in the same
// accumulators.  A layer over a concatenated input [x1 | x2] cuts EACH segment into its 8 chunks and chains chunk c of x2 behind
// chunk c of x1 in the same accumulator.  So a code bit does not depend on the schedule that computed it.
//
// Reference semantics: nn.Linear / nn.ELU / nn.Sigmoid / torch.round / nn.GRU as used at
// bvrnn.py:44-83,163-229.
#include <cstdlib>

#include <cstdint>
#include <type_traits>

#include "bvc_internal.h"

namespace bvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float elu1(float v) { return v > 0.0f ? v : expf(v) - 1.0f; }
__device__ __forceinline__ float sigmoid1(float v) { return 1.0f / (1.0f + expf(-v)); }

struct Resolved { float *p; long long ld; bool ok; int packed; };

// Effective address of a DynPtr for frame t.  Packed frame tensors hold one fragment-packed
// [mt16][dim] matrix per frame (mt16 = rows rounded up to 16).
// Scalar loads of call-descriptor fields.  hipcc emits VECTOR loads for these (the descriptor is not
// provably invariant), and then waits vmcnt(0) for them - which also waits for every weight load issued
// just before.  s_load + lgkmcnt keeps the descriptor off the vector-memory counter.  (The scalar cache
// is invalidated by the acquire at every kernel dispatch, like for kernel arguments.)
__device__ __forceinline__ int sload_i32(const void *p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ long long sload_i64(const void *p) {
    long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// The frame counter is read from the descriptor only by pointers that need it (kind 1/2): layers whose
// operands are all workspace-static never wait for that line.
// MF: the launch covers several frames (grid.y = frame): static pointers advance by their frame stride (other launches never read blockIdx.y)
template <bool MF = false>
__device__ __forceinline__ Resolved resolve(const DynPtr &d, const CallDesc *c, int mt16, int tstep) {
    const int kind = d.meta & 15, packed = (d.meta >> 4) & 1;
    if (kind == 0) return {MF ? d.base + (long long)blockIdx.y * d.poff : d.base, (long long)d.ld, d.base != nullptr, packed};
    const int toff = ((d.meta >> 16) & 255) - 8;
    const int t = sload_i32(&c->t) + tstep;
    if (kind == 2) return {d.base + (((t + toff) & 1) ? d.poff : 0), (long long)d.ld, true, packed};
    const long long T = sload_i64(&c->T);
    float *b = reinterpret_cast<float *>(sload_i64(&c->p[(d.meta >> 8) & 255]));
    const long long tt = (long long)t + toff;
    const bool ok = (b != nullptr) && tt >= 0 && tt < T;
    if (packed) return {ok ? b + tt * (long long)mt16 * d.dim : nullptr, (long long)d.dim, ok, 1};
    return {ok ? b + tt * d.dim : nullptr, T * d.dim, ok, 0};
}

// offset (in floats) of element (m, n) of a fragment-packed [.][ld] matrix
__device__ __forceinline__ long long packed_off(int m, int n, long long ld) {
    return ((((long long)(m >> 4) * (ld >> 4) + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (m & 15)) << 2) + (n & 3);
}

__device__ __forceinline__ void store_out(const Resolved &y, int m, int n, float v) {
    if (y.packed) y.p[packed_off(m, n, y.ld)] = v;
    else y.p[(long long)m * y.ld + n] = v;
}

// Accumulate k-blocks [lo, hi) of one segment into acc[MTW][NG].  wl: this lane's pointer into the packed
// weights of (n-tile, gate 0, k-block 0).  The weight loads of the first chunk are issued BEFORE the
// activation pointer is resolved: weight addresses come from kernel arguments only, while a frame- or
// parity-indexed activation pointer needs the frame counter from the call descriptor (a dependent
// load of a line another kernel has just written), whose latency is thus hidden behind the weights.
// MTW row tiles share every weight fragment in registers (weights cross the L2->CU path once per MTW*16 rows).
template <int NG, int U, int MTW, bool MF = false>
__device__ __forceinline__ void run_segment(const float *wl, long long gate_stride, long long kbs, const DynPtr &xd,
                                            const CallDesc *dsc, int mt16, int tstep, const int (&mtile)[MTW],
                                            const int (&xrow)[MTW], int lane, int g, int lo, int hi,
                                            f32x4 (&acc)[MTW][NG]) {
    const float *xl[MTW];
    int xstep = 0;
    bool have_x = false;
    auto resolve_x = [&]() {
        const Resolved x = resolve<MF>(xd, dsc, mt16, tstep);
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            if (x.packed) { xl[j] = x.p + (long long)mtile[j] * (x.ld >> 4) * 256 + lane * 4; xstep = 256; }
            else          { xl[j] = x.p + (long long)xrow[j] * x.ld + g * 4;                  xstep = 16; }
        }
        have_x = true;
    };
    int kb = lo;
    for (; kb + U <= hi; kb += U) {
        f32x4 xv[U][MTW];
        f32x4 wv[U][NG];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < NG; ++q)
                wv[u][q] = *reinterpret_cast<const f32x4 *>(wl + (long long)q * gate_stride + (long long)(kb + u) * kbs);
        if (!have_x) resolve_x();
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < MTW; ++j) xv[u][j] = *reinterpret_cast<const f32x4 *>(xl[j] + (long long)(kb + u) * xstep);
        __builtin_amdgcn_sched_barrier(0);      // keep all U blocks' loads in flight ahead of the MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < MTW; ++j)
#pragma unroll
                    for (int q = 0; q < NG; ++q) acc[j][q] = mfma16(xv[u][j][e], wv[u][q][e], acc[j][q]);
        }
    }
    for (; kb < hi; ++kb) {
        f32x4 wv[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q)
            wv[q] = *reinterpret_cast<const f32x4 *>(wl + (long long)q * gate_stride + (long long)kb * kbs);
        if (!have_x) resolve_x();
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(xl[j] + (long long)kb * xstep);
#pragma unroll
            for (int q = 0; q < NG; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][q] = mfma16(xv[e], wv[q][e], acc[j][q]);
        }
    }
}

template <int NG, int NGRP, int NW, int U, int MTW, bool GIL = false, bool MF = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmParams p, int epi) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][NGRP*NG][MTW][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const CallDesc *dsc = p.desc;
    unsigned long long t_start = 0;
    if (p.probe) t_start = wall_clock64();                           // probe builds of the graph only

    const int n_tiles = p.N >> 4;
    const int m_tiles = (p.M + 15) >> 4;
    const int m_groups = (m_tiles + MTW - 1) / MTW;                  // MTW row tiles per workgroup
    const int mt16 = m_tiles << 4;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int ntile = (slot / m_groups) * 8 + xcd;
    const int mgroup = slot % m_groups;
    if (ntile >= n_tiles) return;                                    // uniform per workgroup
    const int n0 = ntile << 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    // epilogue constants are fetched up front so their latency overlaps the operand stream
    constexpr int NACC = NG * NGRP;
    float bias[NACC];
    if (tid < 256) {
        const int n = n0 + (tid & 15);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            bias[q] = p.bias0 ? p.bias0[(long long)q * p.gate_rows + n] : 0.0f;
            if (NGRP > 1) bias[NG + q] = p.bias1[(long long)q * p.gate_rows + n];
        }
    }

    int mtile[MTW], xrow[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int mt = mgroup * MTW + j;
        mtile[j] = mt < m_tiles ? mt : m_tiles - 1;                  // a missing tile re-reads the last one; never stored
        const int row = (mtile[j] << 4) + r;
        xrow[j] = row < p.M ? row : p.M - 1;                         // natural layout: clamp (row never stored)
    }

    f32x4 acc0[MTW][NG], acc1[MTW][NG];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int q = 0; q < NG; ++q) { acc0[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // every segment is cut over the waves on its own (wave w = chunk w of the summation order, see the head of this file): the
    // accumulator of a wave chains its chunk of the first segment, then its chunk of the second, ...
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s > 0 && s >= p.nseg) break;
        const int sb = p.seg[s].K >> 4;
        const int lo = (int)(((long long)sb * wave) / NW);
        const int hi = (int)(((long long)sb * (wave + 1)) / NW);
        if (lo < hi) {
            // floats between k-blocks / between gates: [n/16][k/16][lane][4] per gate, or gate-interleaved [n/16][k/16][gate][lane][4]
            constexpr long long kbs = GIL ? NG * 256 : 256;
            const long long gate_stride = GIL ? 256 : (long long)(p.gate_rows >> 4) * p.seg[s].wnb * 256;
            const float *wl = p.seg[s].w + (long long)ntile * p.seg[s].wnb * kbs + lane * 4;
            if constexpr (NGRP == 1) {
                run_segment<NG, U, MTW, MF>(wl, gate_stride, kbs, p.seg[s].x, dsc, mt16, p.tstep, mtile, xrow, lane, g, lo, hi, acc0);
            } else {
                if (p.seg[s].grp == 0)
                    run_segment<NG, U, MTW, MF>(wl, gate_stride, kbs, p.seg[s].x, dsc, mt16, p.tstep, mtile, xrow, lane, g, lo, hi, acc0);
                else
                    run_segment<NG, U, MTW, MF>(wl, gate_stride, kbs, p.seg[s].x, dsc, mt16, p.tstep, mtile, xrow, lane, g, lo, hi, acc1);
            }
        }
    }

    // ---- cross-wave reduction through LDS, fixed order (deterministic)
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = ((g * 4 + e) << 4) + r;              // D[row=g*4+e][col=r]
                red[((wave * NACC + q) * MTW + j) * 256 + idx] = acc0[j][q][e];
                if (NGRP > 1) red[((wave * NACC + NG + q) * MTW + j) * 256 + idx] = acc1[j][q][e];
            }
    __syncthreads();
    if (tid >= 256) return;
    const int i = tid >> 4, jj = tid & 15;
    const int n = n0 + jj;
#pragma unroll 1
    for (int j = 0; j < MTW; ++j) {
        const int mt = mgroup * MTW + j;
        const int m = (mt << 4) + i;
        if (mt >= m_tiles || m >= p.M) continue;
        float v[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float sum = red[(a * MTW + j) * 256 + tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += red[((w * NACC + a) * MTW + j) * 256 + tid];
            v[a] = sum + bias[a];
        }
        const Resolved y = resolve<MF>(p.y, dsc, mt16, p.tstep);
        if (epi == EPI_LINEAR || epi == EPI_ELU) {
            float o = v[0];
            if (p.aux.base || (p.aux.meta & 15)) {                            // pre-computed half of a split dot product
                const Resolved ad = resolve<MF>(p.aux, dsc, mt16, p.tstep);
                o += ad.p[(long long)m * ad.ld + n];
            }
            if (epi == EPI_ELU) o = elu1(o);
            store_out(y, m, n, o);
            if (p.y2.meta & 15) {                                             // a second copy (the folded hop keeps ELU(dec.4) of every frame)
                const Resolved y2 = resolve<MF>(p.y2, dsc, mt16, p.tstep);
                if (y2.ok) store_out(y2, m, n, o);
            }
        } else if (epi == EPI_SIGMOID) {
            store_out(y, m, n, sigmoid1(v[0]));
        } else if (epi == EPI_CODE) {
            const float pr = sigmoid1(v[0]);
            float z;
            if (p.sample == CS_ENCODE) {
                z = rintf(pr);                                       // round half to even (torch.round)
            } else {                                                 // BVRNN.forward: straight-through forward value
                float arg = pr;
                if (p.sample == CS_SAMPLE) {
                    const Resolved un = resolve<MF>(p.y2, dsc, mt16, p.tstep);
                    arg = __fadd_rn(__fsub_rn(un.p[(long long)m * un.ld + n], 0.5f), pr);     // (u - 0.5) + p
                }
                z = __fadd_rn(__fsub_rn(rintf(arg), pr), pr);        // round(.) - p + p
            }
            if (p.var_bit) {
                const Resolved bt = resolve<MF>(p.aux, dsc, mt16, p.tstep);
                const float bits = bt.p[(long long)m * bt.ld];
                z = (bits > (float)n) ? z : (z != z ? z : 0.5f);     // z*m + 0.5*(1-m): a NaN stays a NaN under the mask too (NaN * 0)
            }
            store_out(y, m, n, z);
            const Resolved y3 = resolve<MF>(p.y3, dsc, mt16, p.tstep);
            if (y3.ok) store_out(y3, m, n, pr);
        } else if (epi == EPI_MEL) {
            const float d = v[0];
            if (y.ok) store_out(y, m, n, d);
            const Resolved y2 = resolve<MF>(p.y2, dsc, mt16, p.tstep);
            store_out(y2, m, n, (d - p.mean[n]) / p.stdv[n]);
        } else if (epi == EPI_GRU_PART) {
          if constexpr (NG == 3 && NGRP == 1) {
            // gi = W_ih [phi_x_gen ; phi_z] + b_ih with the phi_z half (+ b_ih) taken from the side branch,
            // gh = W_hh h + b_hh entirely from the side branch.  NGRP == 1: v[0..2] hold the phi_x_gen half.
            const long long H = p.gate_rows;
            const float *pi = p.part_i + (long long)m * p.ldpart + n;
            const float *ph = p.part_h + (long long)m * p.ldpart + n;
            const float gi_r = v[0] + pi[0], gi_z = v[1] + pi[H], gi_n = v[2] + pi[2 * H];      // bias0 is null here
            const float rg = sigmoid1(ph[0] + gi_r);
            const float zg = sigmoid1(ph[H] + gi_z);
            const float ng = tanhf(gi_n + rg * ph[2 * H]);
            const Resolved hprev = resolve<MF>(p.aux, dsc, mt16, p.tstep);
            const float hp = hprev.packed ? hprev.p[packed_off(m, n, hprev.ld)] : hprev.p[(long long)m * hprev.ld + n];
            const float hn = (hp - ng) * zg + ng;
            store_out(y, m, n, hn);
            const Resolved y2 = resolve<MF>(p.y2, dsc, mt16, p.tstep);
            if (y2.ok) store_out(y2, m, n, hn);
          }
        } else if (epi == EPI_GRU) {
          if constexpr (NG == 3 && NGRP == 2) {
            float gi_r = v[0], gi_z = v[1], gi_n = v[2];
            const float gh_r = v[3], gh_z = v[4], gh_n = v[5];
            if (p.y3.meta & 15) {                                            // W_ih[:, H:] phi_z + b_ih, batched over all frames
                const Resolved pg = resolve<MF>(p.y3, dsc, mt16, p.tstep);
                const float *q = pg.p + (long long)m * pg.ld + n;
                gi_r += q[0]; gi_z += q[p.gate_rows]; gi_n += q[2 * p.gate_rows];
            }
            const float rg = sigmoid1(gh_r + gi_r);
            const float zg = sigmoid1(gh_z + gi_z);
            const float ng = tanhf(gi_n + rg * gh_n);
            const Resolved hprev = resolve<MF>(p.aux, dsc, mt16, p.tstep);
            const float hp = hprev.packed ? hprev.p[packed_off(m, n, hprev.ld)] : hprev.p[(long long)m * hprev.ld + n];
            const float hn = (hp - ng) * zg + ng;
            store_out(y, m, n, hn);
            const Resolved y2 = resolve<MF>(p.y2, dsc, mt16, p.tstep);
            if (y2.ok) store_out(y2, m, n, hn);
          }
        }
    }
    if (p.probe && tid == 0) {   // slots: [0, T*nodes) first-workgroup start, [T*nodes, 2*T*nodes) last end
        const int nps = sload_i32(&dsc->nodes_per_step);
        const long long slot_i = (long long)(sload_i32(&dsc->t) + p.tstep) * nps + p.node;
        atomicMin(&p.probe[slot_i], t_start);
        atomicMax(&p.probe[sload_i64(&dsc->T) * nps + slot_i], (unsigned long long)wall_clock64());
    }
}

template <int NG, int NGRP, int NW, int U, int MTW, bool GIL = false>
static void launch_skinny_t(const GemmParams &p, int epi, hipStream_t s) {
    const int n_tiles = p.N / 16, m_tiles = (p.M + 15) / 16, m_groups = (m_tiles + MTW - 1) / MTW;
    const int grid = 8 * ((n_tiles + 7) / 8) * m_groups;
    const size_t lds = (size_t)NW * NG * NGRP * MTW * 256 * sizeof(float);
    if (p.frames > 1) hipLaunchKernelGGL((gemm_skinny_kernel<NG, NGRP, NW, U, MTW, GIL, true>), dim3(grid, p.frames), dim3(NW * 64), lds, s, p, epi);
    else              hipLaunchKernelGGL((gemm_skinny_kernel<NG, NGRP, NW, U, MTW, GIL>), dim3(grid), dim3(NW * 64), lds, s, p, epi);
}

template <typename K>
static int allow_lds(K kern, int bytes) {
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return BVC_OK;
}

int skinny_kernels_init() {
    int rc;
    if ((rc = allow_lds(gemm_skinny_kernel<3, 2, 8, 1, 1, true>, 8 * 6 * 1024))) return rc;
    if ((rc = allow_lds(gemm_skinny_kernel<3, 2, 8, 1, 1, false>, 8 * 6 * 1024))) return rc;
    if ((rc = allow_lds(gemm_skinny_kernel<3, 2, 8, 3, 2, true>, 8 * 6 * 2 * 1024))) return rc;
    return BVC_OK;
}

// mtw = row tiles (of 16) per workgroup: 1 maximises the number of workgroups (lowest latency of a single
// dependent chain), 2 / 4 share every weight fragment among more rows (less L2->CU traffic: higher
// throughput when several independent chains run concurrently).
int launch_gemm_skinny(const GemmParams &p, int epi, hipStream_t s, int mtw) {
    if (p.M <= 0) return BVC_OK;
    if (p.N % 16) { set_error("gemm_skinny: N=%d not a multiple of 16", p.N); return BVC_EINVAL; }
    int nb = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const DynPtr &x = p.seg[i].x;
        const long long xl = dp_kind(x) == 1 ? x.dim : x.ld;
        if (p.seg[i].K % 16 || xl % (dp_packed(x) ? 16 : 4) || p.seg[i].wnb <= 0) {
            set_error("gemm_skinny: segment %d K=%d row length %lld: K must be a multiple of 16, rows of 4 (16 packed)",
                      i, p.seg[i].K, xl);
            return BVC_EINVAL;
        }
        nb += p.seg[i].K / 16;
    }
    if (p.frames > 1) {                                      // frames in grid.y: static pointers only (the frame counter of a call descriptor is not advanced per block)
        bool ok = dp_kind(p.y) == 0 && (dp_kind(p.aux) == 0) && (p.y2.meta & 15) == 0 && (p.y3.meta & 15) == 0 && !p.probe;
        for (int i = 0; i < p.nseg; ++i) ok = ok && dp_kind(p.seg[i].x) == 0;
        if (!ok) { set_error("gemm_skinny: a multi-frame launch takes static pointers only"); return BVC_EINVAL; }
    }
    if (nb != p.nb_total) { set_error("gemm_skinny: nb_total %d does not match the segments (%d)", p.nb_total, nb); return BVC_EINVAL; }
    const int m_tiles = (p.M + 15) / 16;
    if (mtw > m_tiles) mtw = m_tiles >= 4 ? 4 : (m_tiles >= 2 ? 2 : 1);
    if (mtw != 2 && mtw != 4) mtw = 1;
    // How many k-blocks a wave keeps in flight is a trade between one chain and several: chunks of 4 give the shortest
    // single launch (4.0 us, 4,180 audio-s/s on one stream) but their load bursts crowd out the other streams' kernels.
    // With FOUR chains in flight: chunks of 4 -> 6,750 audio-s/s, of 2 -> 6,830 (one stream 4,080), of 1 -> 6,850 (3,850);
    // with three chains the spread was 6,100 / 6,100 / 6,270.  Default: chunks of 2 (1 at K = 2048, 12 waves);
    // BVC_LATENCY=1 selects the single-chain optimum.
    static const bool latency_mode = getenv("BVC_LATENCY") != nullptr && getenv("BVC_LATENCY")[0] == '1';
    ProbeScope probe((epi == EPI_GRU || epi == EPI_GRU_PART) ? PK_GRU : PK_LINEAR, s);
    if (p.gate_il && epi != EPI_GRU) { set_error("gemm_skinny: gate-interleaved weights are for the GRU launch only"); return BVC_EINVAL; }
    // EIGHT waves everywhere: a wave is one chunk of the summation order (head of this file), so the wave count is part of the result.
    // (Round 1 ran the GRU launch and the K = 2048 layers on 12 / 16 waves for 2 % more throughput with three chains in flight.)
    if (epi == EPI_GRU) {
        // chunks of ONE k-block in flight: a workgroup that bursts 8 KiB of loads per wave only queues them (tools/gru_splitk_bench.hip)
        if (!p.gate_il)    launch_skinny_t<3, 2, 8, 1, 1, false>(p, epi, s);
        else if (mtw >= 2) launch_skinny_t<3, 2, 8, 3, 2, true>(p, epi, s);       // (mtw 4: two row tiles per workgroup here - four sets of partial tiles exceed the LDS)
        else               launch_skinny_t<3, 2, 8, 1, 1, true>(p, epi, s);
    } else if (epi == EPI_GRU_PART) {
        launch_skinny_t<3, 1, 8, 4, 1>(p, epi, s);
    } else {
        // mtw 1: chunks of 4 k-blocks (52 VGPRs) measured 2 % faster than chunks of 8 (88 VGPRs), alone and
        // with three chains in flight
        if (mtw == 4)      launch_skinny_t<1, 1, 8, 4, 4>(p, epi, s);
        else if (mtw == 2) launch_skinny_t<1, 1, 8, 8, 2>(p, epi, s);
        else if (latency_mode) launch_skinny_t<1, 1, 8, 4, 1>(p, epi, s);
        else                   launch_skinny_t<1, 1, 8, 2, 1>(p, epi, s);
    }
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

__global__ void step_advance_kernel(CallDesc *d, int by) {
    if (threadIdx.x == 0) d->t = d->t + by;
}

int launch_step_advance(CallDesc *d, int by, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, d, by);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

__global__ void set_desc_kernel(CallDesc *d, CallDesc v) {
    if (threadIdx.x == 0) *d = v;
}

int launch_set_desc(CallDesc *d, const CallDesc &v, hipStream_t s) {
    hipLaunchKernelGGL(set_desc_kernel, dim3(1), dim3(64), 0, s, d, v);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// Where element (row, col) of a batched GEMM goes (GemmOut):
//   GO_NATURAL           y[row][col]
//   GO_FRAME_MAJOR_ROWS  rows come utterance-major (row = b*T + t) and leave frame-major (row' = t*mt16 + b):
//                        the following layers then see whole 16-utterance groups of ONE frame per row tile
//   GO_PACKED_FRAMES     rows are frame-major (row = t*mt16 + b); frame t is written as the fragment-packed
//                        [mt16][N] matrix the recurrent kernels read: a 16x16 MFMA tile is one 1 KiB block
//   GO_PACKED_FROM_UTT   utterance-major rows straight to packed frames (16-byte granules; small jobs only)
__device__ __forceinline__ long long out_index(int row, int col, int N, long long ldy, long long T, int mt16, int mode) {
    if (mode == GO_NATURAL) return (long long)row * ldy + col;
    if (mode == GO_FRAME_MAJOR_ROWS) {
        const long long bb = row / T, tt = row - bb * T;
        return (tt * mt16 + bb) * ldy + col;
    }
    if (mode == GO_PACKED_FRAMES) {
        const int tt = row / mt16, bb = row - tt * mt16;
        return (long long)tt * mt16 * N + packed_off(bb, col, N);
    }
    const long long bb = row / T, tt = row - bb * T;
    return tt * (long long)mt16 * N + packed_off((int)bb, col, N);
}

// ------------------------------------------------------------------------------------------------
// Batched GEMM: y[M,N] = act(x[M,K] @ w[N,K]^T + bias).  128x128 per workgroup, 64x64 per wave.
// ROWVEC (GO_NATURAL output, N and ldy multiples of 4): the MFMA operands are swapped (tile of y^T = w x^T), so that a lane's four
// results are four CONSECUTIVE COLUMNS of one output row and leave as one 16-byte store (the K = 64 / 80 first layers are bound by
// their 113 MB of output: 64 four-byte stores per lane otherwise).  Same products, same order of summation per output.
template <int ACT, bool ROWVEC = false>
__global__ __launch_bounds__(256, 2) void gemm_batched_kernel(const float *__restrict__ x, long long ldx,
                                                           const float *__restrict__ w, long long ldw,
                                                           const float *__restrict__ bias, int M, int N,
                                                           int K, float *__restrict__ y, long long ldy,
                                                           long long frames_T, int mt16, int out_mode) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (m0 >= M || n0 >= N) return;                                  // wave-uniform

    f32x4 acc[4][4], tot[4][4];                          // the running chunk | the sum of the finished chunks (head of this file)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const float *xp[4];
    const float *wp[4];
    bool xok[4], wok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + i * 16 + r;
        xok[i] = row < M;
        xp[i] = x + (long long)(xok[i] ? row : (M - 1)) * ldx + g * 4;
        const int col = n0 + i * 16 + r;
        wok[i] = col < N;
        wp[i] = w + (long long)(wok[i] ? col : (N - 1)) * ldw + g * 4;
    }
    // The fragments of k-block kb+1 are requested before block kb is multiplied (unconditionally: past the end the last block
    // is read again), so that only the first request's latency is exposed - these are the K = 64 / 80 first layers, five blocks.
    const int nblk = K >> 4;
    int chunk = 0, cend = nblk >> 3;                       // chunk c = blocks [nblk*c/8, nblk*(c+1)/8)
    while (cend == 0) { ++chunk; cend = (nblk * (chunk + 1)) >> 3; }
    bool first_chunk = true;
    f32x4 xv[4], wv[4], xn[4], wn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xv[i] = *reinterpret_cast<const f32x4 *>(xp[i]);
        wv[i] = *reinterpret_cast<const f32x4 *>(wp[i]);
    }
    // one k-block; START: the block opens a chunk - its first MFMA per tile starts from zero (no accumulator to clear)
    auto block = [&](auto start_c) {
        constexpr bool START = decltype(start_c)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 c = (START && e == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j];
                    acc[i][j] = ROWVEC ? mfma16(wv[j][e], xv[i][e], c) : mfma16(xv[i][e], wv[j][e], c);
                }
    };
    bool cstart = true;
#pragma unroll 1
    for (int kb = 0; kb < nblk; ++kb) {
        const long long nxt = (long long)(kb + 1 < nblk ? kb + 1 : nblk - 1) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xn[i] = *reinterpret_cast<const f32x4 *>(xp[i] + nxt);
            wn[i] = *reinterpret_cast<const f32x4 *>(wp[i] + nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cstart) block(std::true_type());
        else        block(std::false_type());
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xv[i] = xn[i]; wv[i] = wn[i]; }
        cstart = kb + 1 == cend;
        if (cstart) {                                      // (uniform) block kb ends chunk `chunk`: add it to the sum
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) tot[i][j] = first_chunk ? acc[i][j] : tot[i][j] + acc[i][j];
            first_chunk = false;
            do { ++chunk; cend = (nblk * (chunk + 1)) >> 3; } while (chunk < 7 && cend == kb + 1);     // (empty chunks: K < 128)
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = tot[i][j];
    if (ROWVEC) {                                          // acc[i][j][e] = y[m0 + 16 i + r][n0 + 16 j + 4 g + e]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + i * 16 + r;
            if (row >= M) continue;
            float *yr = y + (long long)row * ldy + n0 + g * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + j * 16 + g * 4;
                if (col >= N) continue;
                const f32x4 b4 = bias ? *reinterpret_cast<const f32x4 *>(bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 v = acc[i][j] + b4;
                if (ACT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = elu1(v[e]);
                }
                *reinterpret_cast<f32x4 *>(yr + j * 16) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + j * 16 + r;
        if (col >= N) continue;
        const float b = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + i * 16 + g * 4 + e;
                if (row < M) {
                    float v = acc[i][j][e] + b;
                    if (ACT == 1) v = elu1(v);
                    y[out_index(row, col, N, ldy, frames_T, mt16, out_mode)] = v;
                }
            }
    }
}

// LDS-tiled variant for the large layers (K a multiple of 32, N a multiple of 128): 128x128 tile, 32-deep
// stages, two LDS buffers.  The next stage travels global -> registers while the current one feeds the
// MFMAs, and is parked in the other LDS buffer afterwards (one barrier per stage).  Rows are padded to
// 36 floats so that the 16-byte fragment reads of 8 consecutive lanes cover all banks.  The k order seen
// by each accumulator is the one of gemm_batched_kernel (16-blocks in order, k = 4*g + e inside), so both
// produce the same bits: eight chunks of K/8, each a chain from zero, added in order (head of this file).
// BM: rows per workgroup tile, 128 or 64 (the half-height form finishes the last, partly filled round of a launch: see
// launch_gemm_batched); m_off: first row of this launch's tiles.
template <int ACT, int BM, bool ROWVEC = false>
__global__ __launch_bounds__(256, 2) void gemm_batched_lds_kernel(const float *__restrict__ x, long long ldx,
                                                                  const float *__restrict__ w, long long ldw,
                                                                  const float *__restrict__ bias, int M, int N,
                                                                  int K, float *__restrict__ y, long long ldy,
                                                                  long long frames_T, int mt16, int out_mode, int m_off) {
    constexpr int BK = 32, LDT = BK + 4;                 // floats per LDS row
    static_assert(BM == 128 || BM == 64 || BM == 32, "tile heights");
    constexpr int MI = BM / 32;                          // 16-row MFMA tiles per wave (waves: 2 along M x 2 along N)
    constexpr int PA = BM / 32;                          // staging passes of 32 rows for the A operand
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][A BMxLDT | B 128xLDT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int mblk = m_off + blockIdx.y * BM, nblk = blockIdx.x * 128;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 64;

    // global staging: thread -> (row = tid/8 + 32*p, 16-byte piece tid%8) of the BM x 32 / 128 x 32 stage of the operands
    const int srow = tid >> 3, spc = (tid & 7) * 4;
    const float *xg[PA], *wg[4];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = mblk + srow + 32 * p;
        xg[p] = x + (long long)(row < M ? row : M - 1) * ldx + spc;      // rows past M: re-read the last one, never stored
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) wg[p] = w + (long long)(nblk + srow + 32 * p) * ldw + spc;
    f32x4 acc[MI][4], tot[MI][4];                        // the running chunk | the sum of the finished chunks (head of this file)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    constexpr int STAGE = (BM + 128) * LDT;              // floats per LDS buffer
    f32x4 ga[PA], gb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < PA; ++p) ga[p] = *reinterpret_cast<const f32x4 *>(xg[p] + (long long)kt * BK);
#pragma unroll
        for (int p = 0; p < 4; ++p) gb[p] = *reinterpret_cast<const f32x4 *>(wg[p] + (long long)kt * BK);
    };
    auto park = [&](int buf) {
        float *A = smem + buf * STAGE, *B = A + BM * LDT;
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<f32x4 *>(A + (srow + 32 * p) * LDT + spc) = ga[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(B + (srow + 32 * p) * LDT + spc) = gb[p];
    };
    const int nk = K / BK;
    const int cst = nk >> 3;                             // stages per chunk (K is a multiple of 256: launch_gemm_batched)
    gload(0);
    park(0);
    __syncthreads();
    // one 32-deep stage; FIRST: the stage opens a chunk - its first MFMA per tile starts from zero (no accumulator to clear)
    auto stage = [&](int kt, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const float *A = smem + (kt & 1) * STAGE + (wm + r) * LDT + g * 4;
        const float *B = smem + (kt & 1) * STAGE + BM * LDT + (wn + r) * LDT + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 av[MI], bv[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = *reinterpret_cast<const f32x4 *>(A + i * 16 * LDT + h * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(B + i * 16 * LDT + h * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 c = (FIRST && h == 0 && e == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = ROWVEC ? mfma16(bv[j][e], av[i][e], c) : mfma16(av[i][e], bv[j][e], c);
                    }
        }
    };
    int in_chunk = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        if (in_chunk == 0) stage(kt, std::true_type());
        else               stage(kt, std::false_type());
        if (++in_chunk == cst) {                         // (uniform) the chunk is complete: add it to the sum
            in_chunk = 0;
            if (kt < cst) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) tot[i][j] = acc[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) tot[i][j] += acc[i][j];
            }
        }
        if (more) park((kt + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = tot[i][j];

    const int m0 = mblk + wm, n0 = nblk + wn;
    if (ROWVEC) {           // operands swapped (tile of y^T): acc[i][j][e] = y[m0 + 16 i + r][n0 + 16 j + 4 g + e], one 16-byte granule in every
#pragma unroll              // output layout (see gemm_batched_kernel)
        for (int i = 0; i < MI; ++i) {
            const int row = m0 + i * 16 + r;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + j * 16 + g * 4;
                const f32x4 b4 = bias ? *reinterpret_cast<const f32x4 *>(bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 v = acc[i][j] + b4;
                if (ACT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = elu1(v[e]);
                }
                *reinterpret_cast<f32x4 *>(y + out_index(row, col, N, ldy, frames_T, mt16, out_mode)) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + j * 16 + r;
        const float b = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + i * 16 + g * 4 + e;
                if (row < M) {
                    float v = acc[i][j][e] + b;
                    if (ACT == 1) v = elu1(v);
                    y[out_index(row, col, N, ldy, frames_T, mt16, out_mode)] = v;
                }
            }
    }
}

int launch_gemm_batched(const float *x, long long ldx, const float *w, long long ldw, const float *bias,
                        int M, int N, int K, int act, float *y, long long ldy, hipStream_t s, int out_mode,
                        long long frames_T, int mt16) {
    if (M <= 0) return BVC_OK;
    if (K % 16 || ldx % 4 || ldw % 4) {
        set_error("gemm_batched: K=%d ldx=%lld ldw=%lld must be multiples of 16/4/4", K, ldx, ldw);
        return BVC_EINVAL;
    }
    if (out_mode != GO_NATURAL) {
        const bool utt_rows = out_mode == GO_FRAME_MAJOR_ROWS || out_mode == GO_PACKED_FROM_UTT;
        if (mt16 <= 0 || mt16 % 16 || N % 16 || (utt_rows && (frames_T <= 0 || M % frames_T || M / frames_T > mt16)) ||
            (out_mode == GO_PACKED_FRAMES && M % mt16)) {
            set_error("gemm_batched: inconsistent frame layout (M=%d T=%lld mt16=%d N=%d mode=%d)", M, frames_T, mt16, N, out_mode);
            return BVC_EINVAL;
        }
    }
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    ProbeScope probe(PK_BATCHED, s);
    static const bool no_lds = getenv("BVC_NO_LDS_GEMM") != nullptr;      // A/B switch for the profiles
    if (!no_lds && K % 256 == 0 && N % 128 == 0) {          // (eight chunks of whole 32-deep stages)
        const size_t lds = (size_t)2 * 2 * 128 * 36 * sizeof(float);
        static bool attr = false;
        if (!attr) {
            const void *ks[] = {(const void *)gemm_batched_lds_kernel<0, 128, false>, (const void *)gemm_batched_lds_kernel<1, 128, false>,
                                (const void *)gemm_batched_lds_kernel<0, 64, false>,  (const void *)gemm_batched_lds_kernel<1, 64, false>,
                                (const void *)gemm_batched_lds_kernel<0, 32, false>,  (const void *)gemm_batched_lds_kernel<1, 32, false>,
                                (const void *)gemm_batched_lds_kernel<0, 128, true>,  (const void *)gemm_batched_lds_kernel<1, 128, true>,
                                (const void *)gemm_batched_lds_kernel<0, 64, true>,   (const void *)gemm_batched_lds_kernel<1, 64, true>,
                                (const void *)gemm_batched_lds_kernel<0, 32, true>,   (const void *)gemm_batched_lds_kernel<1, 32, true>};
            for (const void *k : ks) BVC_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        // 16-byte epilogue stores (operands swapped, see gemm_batched_kernel) whenever the output granules are aligned
        const bool rv = N % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
        auto launch = [&](auto act_c, auto bm_c, dim3 grid_, int m_off_) {
            constexpr int A_ = decltype(act_c)::value, BM_ = decltype(bm_c)::value;
            const size_t lds_ = (size_t)2 * (BM_ + 128) * 36 * sizeof(float);
            if (rv) hipLaunchKernelGGL((gemm_batched_lds_kernel<A_, BM_, true>), grid_, dim3(256), lds_, s, x, ldx, w, ldw, bias, M, N, K, y, ldy, frames_T, mt16, out_mode, m_off_);
            else    hipLaunchKernelGGL((gemm_batched_lds_kernel<A_, BM_, false>), grid_, dim3(256), lds_, s, x, ldx, w, ldw, bias, M, N, K, y, ldy, frames_T, mt16, out_mode, m_off_);
        };
        // Tail: 2 workgroups fit a CU, so the chip takes 512 tiles per round; a grid that ends with a partly filled round
        // (configs[1]: 215 x 8 = 1,720 tiles = 3.36 rounds) pays a whole round for it.  The rows of that last round are
        // computed with half-height tiles instead (a second launch; the same k order per accumulator, so the same bits) - or with
        // quarter-height tiles (three workgroups of 45 KiB LDS per CU: 768 slots) when those still fit one round: a quarter tile
        // has a quarter of the products but the same 128-column weight stage, so it is about 0.4 of a full tile's time against the
        // half tile's 0.55 (the balance a stream-K split would buy, without a fix-up pass and with the summation order untouched).
        static const bool no_tail = getenv("BVC_NO_GEMM_TAIL") != nullptr;
        static const bool no_q = getenv("BVC_NO_GEMM_QUARTER") != nullptr;
        const int ncol = N / 128, rows_blk = (M + 127) / 128;
        const int slots = 512, per_round = slots / ncol > 0 ? slots / ncol : 1;     // row blocks per full round
        int full_blk = rows_blk;
        if (!no_tail && rows_blk > per_round && rows_blk % per_round != 0 && (rows_blk % per_round) * 2 <= per_round)   // the half tiles fit ONE round
            full_blk = (rows_blk / per_round) * per_round;
        if (full_blk > 0) {
            dim3 g1(ncol, full_blk);
            if (act == 1) launch(std::integral_constant<int, 1>(), std::integral_constant<int, 128>(), g1, 0);
            else          launch(std::integral_constant<int, 0>(), std::integral_constant<int, 128>(), g1, 0);
        }
        if (full_blk < rows_blk) {
            const int m_off = full_blk * 128;
            const int q_tiles = ncol * ((M - m_off + 31) / 32);
            if (!no_q && q_tiles <= 768) {
                dim3 g2(ncol, (M - m_off + 31) / 32);
                if (act == 1) launch(std::integral_constant<int, 1>(), std::integral_constant<int, 32>(), g2, m_off);
                else          launch(std::integral_constant<int, 0>(), std::integral_constant<int, 32>(), g2, m_off);
            } else {
                dim3 g2(ncol, (M - m_off + 63) / 64);
                if (act == 1) launch(std::integral_constant<int, 1>(), std::integral_constant<int, 64>(), g2, m_off);
                else          launch(std::integral_constant<int, 0>(), std::integral_constant<int, 64>(), g2, m_off);
            }
        }
        BVC_HIP_TRY(hipGetLastError());
        return BVC_OK;
    }
    const bool rowvec = out_mode == GO_NATURAL && N % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                        (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
    if (rowvec) {
        if (act == 1) hipLaunchKernelGGL((gemm_batched_kernel<1, true>), grid, dim3(256), 0, s, x, ldx, w, ldw, bias, M, N, K, y, ldy, frames_T, mt16, out_mode);
        else          hipLaunchKernelGGL((gemm_batched_kernel<0, true>), grid, dim3(256), 0, s, x, ldx, w, ldw, bias, M, N, K, y, ldy, frames_T, mt16, out_mode);
    } else if (act == 1)
        hipLaunchKernelGGL(gemm_batched_kernel<1>, grid, dim3(256), 0, s, x, ldx, w, ldw, bias, M, N, K, y, ldy, frames_T, mt16, out_mode);
    else
        hipLaunchKernelGGL(gemm_batched_kernel<0>, grid, dim3(256), 0, s, x, ldx, w, ldw, bias, M, N, K, y, ldy, frames_T, mt16, out_mode);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void normalize_rows_kernel(const float *__restrict__ y, const float *__restrict__ mean,
                                      const float *__restrict__ stdv, long long total, int n,
                                      float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % n);
        out[i] = (y[i] - mean[c]) / stdv[c];
    }
}

int launch_normalize_rows(const float *y, const float *mean, const float *stdv, long long rows, int n,
                          float *out, hipStream_t s) {
    const long long total = rows * n;
    if (total <= 0) return BVC_OK;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(256), 0, s, y, mean, stdv, total, n, out);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

__global__ void repack_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, long long ld, int rows,
                                   int n, int dir) {
    const long long total = (long long)rows * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n), c = (int)(i % n);
        if (dir == 0) dst[packed_off(m, c, n)] = src[(long long)m * ld + c];
        else          dst[(long long)m * ld + c] = src[packed_off(m, c, n)];
    }
}

int launch_repack_rows(const float *src, float *dst, long long ld_natural, int rows, int n, int dir, hipStream_t s) {
    const long long total = (long long)rows * n;
    if (total <= 0) return BVC_OK;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(repack_rows_kernel, dim3(grid), dim3(256), 0, s, src, dst, ld_natural, rows, n, dir);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

__global__ void fill_kernel(float *p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}

int launch_fill(float *p, float v, long long n, hipStream_t s) {
    if (n <= 0) return BVC_OK;
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, s, p, v, n);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

__global__ void copy_rows_kernel(const float *__restrict__ src, long long lds_, float *__restrict__ dst,
                                 long long ldd, int rows, int n) {
    const long long total = (long long)rows * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long rr = i / n;
        const int c = (int)(i % n);
        dst[rr * ldd + c] = src[rr * lds_ + c];
    }
}

int launch_copy_rows(const float *src, long long lds_, float *dst, long long ldd, int rows, int n,
                     hipStream_t s) {
    const long long total = (long long)rows * n;
    if (total <= 0) return BVC_OK;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid), dim3(256), 0, s, src, lds_, dst, ldd, rows, n);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// ---- KL term of BVRNN.forward ---------------------------------------------------------------------
// One workgroup per frame; thread (b, j) pairs strided, fixed-order tree reduction (deterministic).
__global__ __launch_bounds__(256) void kld_frames_kernel(const float *__restrict__ prob, const float *__restrict__ prior,
                                                         const float *__restrict__ bits, int B, long long T, int Z,
                                                         float *__restrict__ kld) {
    __shared__ float red[256];
    const long long t = blockIdx.x;
    float sum = 0.0f;
    for (int i = threadIdx.x; i < B * Z; i += 256) {
        const int b = i / Z, j = i - b * Z;
        const long long o = ((long long)b * T + t) * Z + j;
        if (bits && !(bits[(long long)b * T + t] > (float)j)) continue;        // kld_elem * bit_mask
        const float e = prob[o], q = prior[o];
        const float a = e * (logf(fmaxf(e, 1e-3f)) - logf(fmaxf(q, 1e-3f)));
        const float c = (1.0f - e) * (logf(fmaxf(1.0f - e, 1e-3f)) - logf(fmaxf(1.0f - q, 1e-3f)));
        sum += a + c;
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) kld[t] = red[0] / (float)B;
}

int launch_kld_frames(const float *prob, const float *prior, const float *bits, int B, long long T, int Z, float *kld,
                      hipStream_t s) {
    if (T <= 0) return BVC_OK;
    hipLaunchKernelGGL(kld_frames_kernel, dim3((unsigned)T), dim3(256), 0, s, prob, prior, bits, B, T, Z, kld);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
