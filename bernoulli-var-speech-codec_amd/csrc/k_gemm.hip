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
// Reference semantics: nn.Linear / nn.ELU / nn.Sigmoid / torch.round / nn.GRU as used at
// bvrnn.py:44-83,163-229.
#include "bvc_internal.h"

namespace bvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float elu1(float v) { return v > 0.0f ? v : expf(v) - 1.0f; }
__device__ __forceinline__ float sigmoid1(float v) { return 1.0f / (1.0f + expf(-v)); }

struct Resolved { float *p; long long ld; bool ok; };

__device__ __forceinline__ Resolved resolve(const DynPtr &d, const CallDesc *c, int t, long long T) {
    if (d.kind == 0) return {d.base, d.ld, d.base != nullptr};
    if (d.kind == 2) return {d.base + (((t + d.toff) & 1) ? d.poff : 0), d.ld, true};
    float *b = c->p[d.sel];
    const long long tt = (long long)t + d.toff;
    const bool ok = (b != nullptr) && tt >= 0 && tt < T;
    return {ok ? b + tt * d.dim : nullptr, T * d.dim, ok};
}

// Accumulate blocks [lo, hi) (16 k each) of one segment into acc[NG].  Rows beyond M read a clamped
// (valid) row: output row i of an MFMA tile depends only on operand row i, and those rows are never stored.
template <int NG, int U>
__device__ __forceinline__ void run_segment(const float *w, long long ldw, int lo, int hi, const float *xrow_base,
                                            int wrow0, long long gate_rows, int g, f32x4 (&acc)[NG]) {
    const float *xb = xrow_base + (long long)g * 4;
    const float *wb = w + (long long)wrow0 * ldw + (long long)g * 4;
    int kb = lo;
    for (; kb + U <= hi; kb += U) {
        f32x4 xv[U];
        f32x4 wv[U][NG];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = *reinterpret_cast<const f32x4 *>(xb + (long long)(kb + u) * 16);
#pragma unroll
            for (int q = 0; q < NG; ++q)
                wv[u][q] = *reinterpret_cast<const f32x4 *>(wb + (long long)q * gate_rows * ldw +
                                                            (long long)(kb + u) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep all U blocks' loads in flight ahead of the MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < NG; ++q) acc[q] = mfma16(xv[u][e], wv[u][q][e], acc[q]);
        }
    }
    for (; kb < hi; ++kb) {
        f32x4 xv = *reinterpret_cast<const f32x4 *>(xb + (long long)kb * 16);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            f32x4 wv = *reinterpret_cast<const f32x4 *>(wb + (long long)q * gate_rows * ldw + (long long)kb * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q] = mfma16(xv[e], wv[e], acc[q]);
        }
    }
}

template <int NG, int NGRP, int NW, int U>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmParams p, int epi) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][NGRP*NG][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const CallDesc *dsc = p.desc;
    const int t = dsc ? dsc->t : 0;
    const long long T = dsc ? dsc->T : 1;
    unsigned long long *probe = dsc ? dsc->probe : nullptr;
    if (probe && tid == 0)       // slots: [0, T*nodes) first-wave start times, [T*nodes, 2*T*nodes) last end times
        atomicMin(&probe[(long long)t * dsc->nodes_per_step + p.node], (unsigned long long)wall_clock64());

    const int n_tiles = p.N >> 4;
    const int m_tiles = (p.M + 15) >> 4;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int ntile = (slot / m_tiles) * 8 + xcd;
    const int mtile = slot % m_tiles;
    if (ntile >= n_tiles) return;                                    // uniform per workgroup
    const int m0 = mtile << 4, n0 = ntile << 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    f32x4 acc0[NG], acc1[NGRP > 1 ? NG : 1];
#pragma unroll
    for (int q = 0; q < NG; ++q) acc0[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < (NGRP > 1 ? NG : 1); ++q) acc1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int nb = 0;
    for (int s = 0; s < p.nseg; ++s) nb += p.seg[s].K >> 4;
    const int my_lo = (int)(((long long)nb * wave) / NW);
    const int my_hi = (int)(((long long)nb * (wave + 1)) / NW);
    const int xrow = (m0 + r) < p.M ? (m0 + r) : (p.M - 1);

    int base = 0;
    for (int s = 0; s < p.nseg; ++s) {
        const int sb = p.seg[s].K >> 4;
        int lo = my_lo - base, hi = my_hi - base;
        lo = lo < 0 ? 0 : lo;
        hi = hi > sb ? sb : hi;
        if (lo < hi) {
            const Resolved x = resolve(p.seg[s].x, dsc, t, T);
            const float *xrow_base = x.p + (long long)xrow * x.ld;
            if (NGRP == 1 || p.seg[s].grp == 0)
                run_segment<NG, U>(p.seg[s].w, p.seg[s].ldw, lo, hi, xrow_base, n0 + r, p.gate_rows, g, acc0);
            else
                run_segment<NG, U>(p.seg[s].w, p.seg[s].ldw, lo, hi, xrow_base, n0 + r, p.gate_rows, g, acc1);
        }
        base += sb;
    }

    // ---- cross-wave reduction through LDS, fixed order (deterministic)
    constexpr int NACC = NG * NGRP;
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = ((g * 4 + e) << 4) + r;                  // D[row=g*4+e][col=r]
            red[(wave * NACC + q) * 256 + idx] = acc0[q][e];
            if (NGRP > 1) red[(wave * NACC + NG + q) * 256 + idx] = acc1[q][e];
        }
    __syncthreads();
    if (tid >= 256) return;
    float v[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        float sum = red[a * 256 + tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += red[(w * NACC + a) * 256 + tid];
        v[a] = sum;
    }
    const int i = tid >> 4, j = tid & 15;
    const int m = m0 + i, n = n0 + j;
    if (m < p.M) {
        const Resolved y = resolve(p.y, dsc, t, T);
        if (epi == EPI_LINEAR || epi == EPI_ELU) {
            float o = v[0] + p.bias0[n];
            if (epi == EPI_ELU) o = elu1(o);
            y.p[(long long)m * y.ld + n] = o;
        } else if (epi == EPI_CODE) {
            const float logit = v[0] + p.bias0[n];
            const float pr = sigmoid1(logit);
            float z = rintf(pr);                                     // round half to even (torch.round)
            if (p.var_bit) {
                const Resolved bt = resolve(p.aux, dsc, t, T);
                const float bits = bt.p[(long long)m * bt.ld];
                z = (bits > (float)n) ? z : 0.5f;                    // z*m + 0.5*(1-m)
            }
            y.p[(long long)m * y.ld + n] = z;
            const Resolved y3 = resolve(p.y3, dsc, t, T);
            if (y3.ok) y3.p[(long long)m * y3.ld + n] = pr;
        } else if (epi == EPI_MEL) {
            const float d = v[0] + p.bias0[n];
            if (y.ok) y.p[(long long)m * y.ld + n] = d;
            const Resolved y2 = resolve(p.y2, dsc, t, T);
            y2.p[(long long)m * y2.ld + n] = (d - p.mean[n]) / p.stdv[n];
        } else if (NGRP > 1 && NG == 3) {                            // EPI_GRU
            const long long H = p.gate_rows;
            const float gi_r = v[0] + p.bias0[n], gi_z = v[1] + p.bias0[H + n], gi_n = v[2] + p.bias0[2 * H + n];
            const float gh_r = v[NACC > 3 ? 3 : 0] + p.bias1[n];
            const float gh_z = v[NACC > 4 ? 4 : 0] + p.bias1[H + n];
            const float gh_n = v[NACC > 5 ? 5 : 0] + p.bias1[2 * H + n];
            const float rg = sigmoid1(gh_r + gi_r);
            const float zg = sigmoid1(gh_z + gi_z);
            const float ng = tanhf(gi_n + rg * gh_n);
            const Resolved hprev = resolve(p.aux, dsc, t, T);
            const float hp = hprev.p[(long long)m * hprev.ld + n];
            const float hn = (hp - ng) * zg + ng;
            y.p[(long long)m * y.ld + n] = hn;
            const Resolved y2 = resolve(p.y2, dsc, t, T);
            if (y2.ok) y2.p[(long long)m * y2.ld + n] = hn;
        }
    }
    if (probe && tid == 0)
        atomicMax(&probe[(T + t) * dsc->nodes_per_step + p.node], (unsigned long long)wall_clock64());
}

template <int NG, int NGRP, int NW, int U>
static void launch_skinny_t(const GemmParams &p, int epi, int grid, hipStream_t s) {
    const size_t lds = (size_t)NW * NG * NGRP * 256 * sizeof(float);
    hipLaunchKernelGGL((gemm_skinny_kernel<NG, NGRP, NW, U>), dim3(grid), dim3(NW * 64), lds, s, p, epi);
}

int skinny_kernels_init() {
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_kernel<3, 2, 16, 3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 6 * 1024));
    return BVC_OK;
}

int launch_gemm_skinny(const GemmParams &p, int epi, hipStream_t s) {
    if (p.M <= 0) return BVC_OK;
    if (p.N % 16) { set_error("gemm_skinny: N=%d not a multiple of 16", p.N); return BVC_EINVAL; }
    int nb = 0;
    for (int i = 0; i < p.nseg; ++i) {
        if (p.seg[i].K % 16 || p.seg[i].ldw % 4 || (p.seg[i].x.kind != 1 && p.seg[i].x.ld % 4) ||
            (p.seg[i].x.kind == 1 && p.seg[i].x.dim % 4)) {
            set_error("gemm_skinny: segment %d K=%d ldw=%lld: K must be a multiple of 16, strides of 4", i,
                      p.seg[i].K, p.seg[i].ldw);
            return BVC_EINVAL;
        }
        nb += p.seg[i].K / 16;
    }
    const int n_tiles = p.N / 16, m_tiles = (p.M + 15) / 16;
    const int grid = 8 * ((n_tiles + 7) / 8) * m_tiles;
    ProbeScope probe(epi == EPI_GRU ? PK_GRU : PK_LINEAR, s);
    if (epi == EPI_GRU)      launch_skinny_t<3, 2, 16, 3>(p, epi, grid, s);
    else if (nb >= 128)      launch_skinny_t<1, 1, 16, 8>(p, epi, grid, s);
    else                     launch_skinny_t<1, 1, 8, 8>(p, epi, grid, s);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

__global__ void step_advance_kernel(CallDesc *d) {
    if (threadIdx.x == 0) d->t = d->t + 1;
}

int launch_step_advance(CallDesc *d, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, d);
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

// ------------------------------------------------------------------------------------------------
// Batched GEMM: y[M,N] = act(x[M,K] @ w[N,K]^T + bias).  128x128 per workgroup, 64x64 per wave.
template <int ACT>
__global__ __launch_bounds__(256) void gemm_batched_kernel(const float *__restrict__ x, long long ldx,
                                                           const float *__restrict__ w, long long ldw,
                                                           const float *__restrict__ bias, int M, int N,
                                                           int K, float *__restrict__ y, long long ldy) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (m0 >= M || n0 >= N) return;                                  // wave-uniform

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
    const int nblk = K >> 4;
    for (int kb = 0; kb < nblk; ++kb) {
        f32x4 xv[4], wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xv[i] = *reinterpret_cast<const f32x4 *>(xp[i] + (long long)kb * 16);
            wv[i] = *reinterpret_cast<const f32x4 *>(wp[i] + (long long)kb * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(xv[i][e], wv[j][e], acc[i][j]);
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
                    y[(long long)row * ldy + col] = v;
                }
            }
    }
}

int launch_gemm_batched(const float *x, long long ldx, const float *w, long long ldw, const float *bias,
                        int M, int N, int K, int act, float *y, long long ldy, hipStream_t s) {
    if (M <= 0) return BVC_OK;
    if (K % 16 || ldx % 4 || ldw % 4) {
        set_error("gemm_batched: K=%d ldx=%lld ldw=%lld must be multiples of 16/4/4", K, ldx, ldw);
        return BVC_EINVAL;
    }
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    ProbeScope probe(PK_BATCHED, s);
    if (act == 1)
        hipLaunchKernelGGL(gemm_batched_kernel<1>, grid, dim3(256), 0, s, x, ldx, w, ldw, bias, M, N, K, y, ldy);
    else
        hipLaunchKernelGGL(gemm_batched_kernel<0>, grid, dim3(256), 0, s, x, ldx, w, ldw, bias, M, N, K, y, ldy);
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

}  // namespace bvc
