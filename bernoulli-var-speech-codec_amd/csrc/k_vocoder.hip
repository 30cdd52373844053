// Causal-tiny BigVGAN kernels for gfx950.
//
// Every convolution of the generator (conv_pre models.py:212-213, the four ConvTranspose1d
// upsamplers :216-217, the 72 AMPBlock1 convolutions :103-121) runs through ONE kernel template:
// a causal 1-D convolution written as an implicit GEMM on the fp32-input MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation):
//
//      out[t, co] = bias[co] + sum_{j<ks} sum_{ci} W[co, ci, j] * act(in[t - (ks-1-j)*dil, ci])
//      M = 16 time steps, N = 16 output channels, K = 4 input channels of one tap.
//
// Activations live in HBM channels-last (B, L, C), so a time tile plus its causal halo is ONE
// contiguous span: it is loaded with coalesced float4 reads, SnakeBeta (activations.py:107-120) is
// applied on the way in, and the tile is parked in LDS with a row stride of C+2 floats, which
// makes the MFMA A-operand reads (16 rows x 4 channels per instruction) bank-conflict free.
// B operands (weights) are pre-packed on the host in MFMA fragment order, so every weight fetch is
// one coalesced 256-B read that stays L1/L2 resident.  The epilogue fuses bias, the AMP residual
// add, the running sum over the three parallel resblocks and the final /3 (models.py:219-225).
//
// A ConvTranspose1d with kernel 2*stride is the same kernel: its polyphase form
//      out[q*u + p, co] = b[co] + sum_ci in[q, ci] W[ci, co, p] + in[q-1, ci] W[ci, co, p+u]
// is a 2-tap causal convolution with u*Cout output columns whose (B, Lin+1, u*Cout) result IS the
// (B, (Lin+1)*u, Cout) channels-last signal.
#include <cstdlib>
#include <type_traits>

#include "bvc_internal.h"

namespace bvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// xs / num_kernels (models.py:225), a true IEEE division like the reference's - in ONE of a stage's nine launches.  The test is uniform,
// but hipcc turns `if (epi == DIV) o = o / d` into the division (a dozen vector instructions per element) on EVERY launch plus a select;
// the empty asm statement cannot be speculated, so the division stays behind a scalar branch (a fifth of these kernels' vector
// instructions were this).
__device__ __forceinline__ void divide_if(bool div, f32x4 &v, float d) {
    if (div) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] / d;
    }
}

// tile index -> (batch item, tile of the item): the quotient by a run-time divisor through a host-made reciprocal (one scalar multiply-high)
// instead of hipcc's float-reciprocal emulation of the 32-bit division - some forty vector instructions, per tile and wave in the persistent
// kernels.  Exact while bid * tiles_per_batch < 2^32 (launch_* check it).
__device__ __forceinline__ unsigned div_tpb(unsigned bid, unsigned tpb_magic) { return __umulhi(bid, tpb_magic); }
static inline unsigned tpb_magic_of(unsigned d) { return d <= 1u ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / d) + 1u; }      // (d == 1: q = bid handled by the callers)

// Rows of a channels-last (L, C) signal through a BUFFER descriptor of exactly L * C floats: a row before the start or behind the end of the
// signal is out of the descriptor's range and reads as zeros by itself (also a negative row: its byte offset wraps to a huge unsigned one) -
// no clamping, no compare, no select per item (a third of the vector instructions the tile loads of these kernels issued beside SnakeBeta).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float *base, long long L, int C) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(L * C * 4), 0x00020000);
}
__device__ __forceinline__ f32x4 rows_load4(__amdgpu_buffer_rsrc_t rs, int row, int C, int c0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (row * C + c0) * 4, 0, 0));
}

struct ConvArgs {
    const float *in;  long long Lin;
    float *out;       long long Lout;
    long long in_bs, out_bs;      // floats between consecutive batch items of in / out (res, acc like out)
    long long row_begin;          // first output row to compute (rows before it are history: streaming)
    const float *res; const float *acc;
    const float *wp;  const float *bias;
    const float *act_a; const float *act_ib;
    float divisor;
    int epi, ks, dil, cout, ntiles, tiles_per_batch;
};

// sin(x)^2 with |error| < 2.5e-7 (checked against float64 up to |x| = 8060: tests/test_gpu_numerics.py; the reduction
// constants keep their accuracy while k = x*2/pi stays below ~2^17): three-constant Cody-Waite reduction by pi/2 with fma to
// r in [-pi/4, pi/4], then ONE even minimax polynomial sin(r)^2 = u*P(u), u = r^2 (|P error| < 5e-10).  The square removes
// the quadrant sign: sin(x)^2 = sin(r)^2 in even quadrants and 1 - sin(r)^2 in odd ones, i.e. 1/2 -+ (1/2 - sin(r)^2) - so
// h = u*P(u) - 1/2 is computed by the last fma and its sign is flipped for odd quadrants by a multiply with +-1 whose sign bit is
// the quadrant's parity.  k comes from the round-to-nearest of adding 1.5 * 2^23 (no rint, no conversion: the parity is the sum's
// lowest mantissa bit).  14 VALU operations per element, 12 of them packable two elements at a time (round 2: 16 + two
// conversions, two masks and two selects per pair); max |error| 9.7e-8 against 1.1e-7 before (numpy emulation over +-8060).
// ocml's sinf is equally accurate but carries a Payne-Hanek path and costs ~4x the instructions, and the generator evaluates
// 476 of these per output sample on SIMDs whose issue slots it shares with the MFMAs.
__device__ __forceinline__ float sin_squared(float x) {
    const float t = fmaf(x, 0.636619772367581343f, 12582912.0f);
    const float k = t - 12582912.0f;
    float r = fmaf(-k, 1.57079625129699707031e+00f, x);
    r = fmaf(-k, 7.54978941586159635335e-08f, r);
    r = fmaf(-k, 5.39030252995776476554e-15f, r);
    const float u = r * r;
    const float p = fmaf(fmaf(fmaf(fmaf(1.345194032182917e-4f, u, -3.1710113398730755e-3f), u, 4.444364085793495e-2f), u,
                              -3.33333283662796e-1f), u, 1.0f);
    const float h = fmaf(p, u, -0.5f);
    const float sg = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, t) << 31) | 0x3F800000u);      // -1 in odd quadrants
    return fmaf(h, sg, 0.5f);
}

// SnakeBeta (activations.py:107-120): x + 1/(exp(beta)+1e-9) * sin(x*exp(alpha))^2
__device__ __forceinline__ float snakebeta(float x, float a, float ib) {
    return __fadd_rn(x, __fmul_rn(ib, sin_squared(__fmul_rn(x, a))));
}

// Two elements per lane: the same operations as sin_squared / snakebeta on both halves (bit-identical results),
// written on 2-vectors so that the multiplies and fused multiply-adds become packed-fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32: two fp32 lanes per instruction at full rate) - SnakeBeta is ~40 % of the
// generator's vector instructions and shares the SIMD's issue slots with the MFMAs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 sin_squared2(f32x2 x) {
    const f32x2 t = __builtin_elementwise_fma(x, splat2(0.636619772367581343f), splat2(12582912.0f));
    const f32x2 nk = splat2(12582912.0f) - t;             // -k
    f32x2 r = __builtin_elementwise_fma(nk, splat2(1.57079625129699707031e+00f), x);
    r = __builtin_elementwise_fma(nk, splat2(7.54978941586159635335e-08f), r);
    r = __builtin_elementwise_fma(nk, splat2(5.39030252995776476554e-15f), r);
    const f32x2 u = r * r;
    f32x2 p = __builtin_elementwise_fma(splat2(1.345194032182917e-4f), u, splat2(-3.1710113398730755e-3f));
    p = __builtin_elementwise_fma(p, u, splat2(4.444364085793495e-2f));
    p = __builtin_elementwise_fma(p, u, splat2(-3.33333283662796e-1f));
    p = __builtin_elementwise_fma(p, u, splat2(1.0f));
    const f32x2 h = __builtin_elementwise_fma(p, u, splat2(-0.5f));
    // +-1 with the quadrant's parity (the sum's lowest mantissa bit) as sign: one v_lshl_or_b32 per element.  Written as asm: from the
    // C expression (bits(t[i]) << 31) | 0x3F800000 on the two elements hipcc 7.2 built ONE such instruction, on element 0, and fed
    // its result to both halves of the packed fma below (op_sel_hi:[1,0,0]) - tests/test_gpu_numerics.py caught it on pairs that
    // straddle a quadrant; the 2-vector integer form is right but takes two instructions per element.
    // (the s_nop: hipcc pads a packed-fp32 result by one state before its next reader and knows nothing about the asm's reads)
    float s0, s1;
    asm("s_nop 0\n\tv_lshl_or_b32 %0, %2, 31, 1.0\n\tv_lshl_or_b32 %1, %3, 31, 1.0" : "=&v"(s0), "=v"(s1) : "v"(t[0]), "v"(t[1]));
    return __builtin_elementwise_fma(h, (f32x2){s0, s1}, splat2(0.5f));
}
__device__ __forceinline__ f32x2 snakebeta2(f32x2 x, f32x2 a, f32x2 ib) {
#pragma clang fp contract(off)
    const f32x2 q = ib * sin_squared2(x * a);
    return x + q;
}

// CIN: input channels; NTW: 16-column tiles per wave; MT: 16-row tiles per wave.  The four waves of a workgroup split the
// rows (each wave MT row tiles x the same NTW column tiles) or, NSPLIT, the columns (all waves the same MT row tiles, each its
// own NTW column tiles): the second form is for streaming hops, whose one or two new frames are a handful of rows.
template <int CIN, int NTW, int MT, bool NSPLIT = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int S = CIN + 2;                 // LDS row stride (floats): (S/2) odd -> conflict-free
    constexpr int TT = (NSPLIT ? 1 : 4) * MT * 16;   // output rows per workgroup
    constexpr int C4 = CIN / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_batch;
    const long long t0 = a.row_begin + (long long)(blockIdx.x % a.tiles_per_batch) * TT;
    const int ntile0 = NSPLIT ? (blockIdx.y * 4 + wave) * NTW : blockIdx.y * NTW;
    const int halo = (a.ks - 1) * a.dil;
    const int rows = TT + halo;

    // ---- stage the activated input span [t0-halo, t0+TT) in LDS
    const float *inb = a.in + (long long)b * a.in_bs;
    for (int idx = tid; idx < rows * C4; idx += 256) {
        const int row = idx / C4, c4 = idx - row * C4;
        const long long tg = t0 - halo + row;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tg >= 0 && tg < a.Lin) {
            v = *reinterpret_cast<const f32x4 *>(inb + tg * CIN + c4 * 4);
            if (a.act_a) {
                const f32x4 aa = *reinterpret_cast<const f32x4 *>(a.act_a + c4 * 4);
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(a.act_ib + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 o2 = snakebeta2((f32x2){v[e], v[e + 1]}, (f32x2){aa[e], aa[e + 1]}, (f32x2){bb[e], bb[e + 1]});
                    v[e] = o2[0]; v[e + 1] = o2[1];
                }
            }
        }
        float2 *dst = reinterpret_cast<float2 *>(tile + row * S + c4 * 4);
        dst[0] = make_float2(v[0], v[1]);
        dst[1] = make_float2(v[2], v[3]);
    }
    __syncthreads();

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int mbase = NSPLIT ? 0 : wave * MT * 16;
    const float *wl = a.wp + (long long)ntile0 * 64 + lane;
    const long long kstride = (long long)a.ntiles * 64;          // floats per k-step in the packed weights
    bool nok[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) nok[n] = (ntile0 + n) < a.ntiles;

    for (int j = 0; j < a.ks; ++j) {
        const float *arow = tile + (mbase + r + j * a.dil) * S + g;
        const float *wj = wl + (long long)j * C4 * kstride;
        constexpr int CGU = (C4 % 8 == 0) ? 8 : (C4 % 5 == 0) ? 5 : (C4 % 4 == 0) ? 4 : (C4 % 2 == 0) ? 2 : 1;
#pragma unroll 1
        for (int cg0 = 0; cg0 < C4; cg0 += CGU) {
            float bw[CGU][NTW];
#pragma unroll
            for (int u = 0; u < CGU; ++u)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    bw[u][n] = nok[n] ? wj[(long long)(cg0 + u) * kstride + n * 64] : 0.0f;
#pragma unroll
            for (int u = 0; u < CGU; ++u) {
                float av[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = arow[i * 16 * S + (cg0 + u) * 4];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int n = 0; n < NTW; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][n], av[i], acc[i][n], 0, 0, 0);      // tile of out^T (below)
            }
        }
    }

    // ---- epilogue.  The operands are swapped (weights as A, activations as B), so acc[i][n][e] = out[row mbase+16i+r][col 16(ntile0+n)+4g+e]:
    // a lane's four results are four consecutive channels of one output row - one 16-byte store (cout is a multiple of 4).
    const long long ob = (long long)b * a.out_bs;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long long t = t0 + mbase + i * 16 + r;
        if (t >= a.Lout) continue;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int col = (ntile0 + n) * 16 + g * 4;
            if (col >= a.cout) continue;
            const long long o = ob + t * a.cout + col;
            f32x4 v = acc[i][n] + *reinterpret_cast<const f32x4 *>(a.bias + col);
            if (a.epi >= CE_RES) v = v + *reinterpret_cast<const f32x4 *>(a.res + o);        // x = xt + x      (models.py:119)
            if (a.epi >= CE_RES_ACC) v = *reinterpret_cast<const f32x4 *>(a.acc + o) + v;    // xs += resblock  (models.py:224)
            divide_if(a.epi == CE_RES_ACC_DIV, v, a.divisor);                                // xs / num_kernels (models.py:225)
            *reinterpret_cast<f32x4 *>(a.out + o) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One AMPBlock1 iteration in one kernel (models.py:106-119):
//     x' = x + conv2( S2( conv1_dil( S1(x) ) ) )            S = SnakeBeta, both convs causal
// Phase 1 parks S1(x) for the output rows plus both halos in LDS; phase 2 runs conv1 on the MFMA for
// TR rows starting (ks-1) rows before the tile, applies bias + S2 and parks the result in a second LDS
// tile (rows before t=0 are zero: the reference pads AFTER the activation); phase 3 runs conv2 on that
// tile and fuses bias, residual, the sum over the three parallel AMP blocks and the final /3.
// The intermediate never touches HBM: 2 tensor passes per iteration instead of 5.
struct AmpArgs {
    const float *x; float *out; const float *acc;
    long long L;
    const float *w1, *b1, *a1, *ib1;
    const float *w2, *b2, *a2, *ib2;
    float divisor;
    int epi, ks, dil, tiles_per_batch;
    unsigned tpb_magic;           // tpb_magic_of(tiles_per_batch)
    unsigned ntile;               // workgroups that have a tile (grid is padded to a multiple of 8)
    long long bs;                 // floats between batch items of x / out / acc
    long long row_begin;          // first output row (streaming: rows before it are history)
    long long t_origin;           // global time of buffer row 0 (streaming); 0 offline
};

#ifndef BVC_AMP_PINGPONG
#define BVC_AMP_PINGPONG 1
#endif
#ifdef BVC_PHASE_PROBE
__device__ unsigned long long g_phase[16];
#define PHASE(i) do { if (threadIdx.x == 0) { unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&g_phase[i], now_ - last_); last_ = now_; } } while (0)
#else
#define PHASE(i)
#endif
// CS: how many of the four waves lie along the COLUMN tiles (1, 2 or 4); the other 4 / CS lie along the rows.  A wave computes MT row
// tiles x NT / CS column tiles, a workgroup (4 / CS) * MT * 16 rows.  CS = 1 re-reads every weight fragment in all four waves
// (from L2: the weight set of a conv does not fit L1) and feeds MT MFMAs with it; with the waves along the columns a fragment is
// read once per workgroup and feeds CS * MT MFMAs at the same rows per workgroup - the C = 64 stage (180 KB of weights per conv
// at ks = 11) 2.63 -> 2.36 ms per step with CS = 4, MT = 8.  Streaming hops (a hop's one or two new frames are a handful of
// rows: row-split tiles would mostly compute rows nobody reads) use CS = 4 with MT = 2.
template <int C, int MT, int OCC, bool ALIAS, int CS = 1>
__global__ __launch_bounds__(256, OCC) void amp_pair_kernel(AmpArgs a) {
#ifdef BVC_PHASE_PROBE
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = C + 2;
    constexpr int NT = (C + 15) / 16;
    constexpr int C4 = C / 4;
    constexpr int TR = (4 / CS) * MT * 16;                // rows computed by each conv phase
    constexpr int NTL = NT / CS;                          // column tiles of one wave
    static_assert((CS == 1 || CS == 2 || CS == 4) && NT % CS == 0, "the waves along the columns must divide the column tiles");
    constexpr int CGU = C4 < 16 / NTL ? C4 : 16 / NTL;    // k-steps per weight chunk: 16 fragments per lane and register set (offline C = 64,
                                                          // CS = 1: 4 / 8 / 16 k-steps measured, 2.63 / 2.60 / 2.65 ms for the stage)
    constexpr bool W4 = C >= 32 && CGU % 4 == 0;          // streamed weights in 16-byte granules (a.w1 / a.w2 = ConvLayer::wp4)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ks = a.ks, dil = a.dil;
    const int TT = TR - (ks - 1);                          // valid output rows of this workgroup
    // Workgroups are dealt round-robin to the 8 XCDs; neighbouring tiles share their halo rows, so each
    // XCD takes a contiguous run of tiles (the halo then hits in that XCD's L2).
    const unsigned nwg = gridDim.x, per = (nwg + 7u) >> 3;
    unsigned bid = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (bid >= a.ntile) return;                           // grid is padded to a multiple of 8
    const int b = a.tiles_per_batch == 1 ? (int)bid : (int)div_tpb(bid, a.tpb_magic);
    const long long t0 = a.row_begin + (long long)(bid - (unsigned)b * (unsigned)a.tiles_per_batch) * TT;
    const int halo1 = (ks - 1) * dil;
    const int rows1 = TR + halo1;                          // S1(x) rows [t0-(ks-1)-halo1, t0-(ks-1)+TR)
    float *t1 = lds;
    float *t2 = ALIAS ? lds : lds + rows1 * S;             // S2(u) rows [t0-(ks-1), t0-(ks-1)+TR) (+ ks-1 spare): takes over
                                                           // the S1(x) tile once conv1 has consumed it (halves the LDS)
    const float *xb = a.x + (long long)b * a.bs;
    const long long tbase = t0 - (ks - 1);                 // global row of local row 0 of phase 2 / t2

    // ---- phase 1: activated input span.  All global loads of the span are issued before the first
    // SnakeBeta is evaluated (one exposed memory round trip per workgroup instead of one per row group).
    {
        constexpr int NLD = ((TR + 10 * 5) * C4 + 255) / 256;       // ks <= 11, dil <= 5
        // The loads are UNCONDITIONAL (rows outside the signal read a clamped row and are zeroed afterwards): a load under a branch
        // made hipcc wait `vmcnt(0)` behind every other one - five exposed round trips per tile at C = 32 instead of one.
        f32x4 v[NLD];
        const int total = rows1 * C4;
        const __amdgpu_buffer_rsrc_t rs = rows_rsrc(xb, a.L, C);       // rows outside the signal read as zeros (rows_load4)
        const int tfirst = (int)(tbase - halo1);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C4, c4 = idx - row * C4;
            v[i] = rows_load4(rs, tfirst + row, C, c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            if (idx < total) {
                const int row = idx / C4, c4 = idx - row * C4;
                const f32x4 aa = *reinterpret_cast<const f32x4 *>(a.a1 + c4 * 4);
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(a.ib1 + c4 * 4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {                                        // S(0) = 0 keeps the zero padding
                    const f32x2 o2 = snakebeta2((f32x2){v[i][e], v[i][e + 1]}, (f32x2){aa[e], aa[e + 1]}, (f32x2){bb[e], bb[e + 1]});
                    o[e] = o2[0]; o[e + 1] = o2[1];
                }
                float2 *dst = reinterpret_cast<float2 *>(t1 + row * S + c4 * 4);
                dst[0] = make_float2(o[0], o[1]);
                dst[1] = make_float2(o[2], o[3]);
            }
        }
    }
    __syncthreads();
    PHASE(0);

    const int mbase = (wave / CS) * MT * 16;
    const int nt0 = (wave % CS) * NTL;                    // first column tile of this wave
    f32x4 acc[MT][NTL];
    auto mma = [&](const float *tile, int d, const float *wp) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NTL; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // weight fragments of chunk q+1 are fetched while chunk q feeds the MFMAs
        const float *wl = wp + lane;
        constexpr int CPT = C4 / CGU;                       // chunks per tap
        const int nch = ks * CPT;
        // W4 (C >= 32): the weights come packed [tap][cin/16][ntile][64][4] (ConvLayer::wp4): four k-steps' fragments per 16-byte load
        constexpr int G4 = W4 ? CGU / 4 : CGU;                 // load granules per column tile and chunk
        typedef typename std::conditional<W4, f32x4, float>::type BW;
        BW bcur[G4][NTL], bnxt[G4][NTL];
        auto loadb = [&](BW (&dstb)[G4][NTL], int q) {
            if constexpr (W4) {
                const f32x4 *wq = reinterpret_cast<const f32x4 *>(wp) + (long long)q * G4 * NT * 64 + lane;
#pragma unroll
                for (int u = 0; u < G4; ++u)
#pragma unroll
                    for (int n = 0; n < NTL; ++n) dstb[u][n] = wq[(u * NT + nt0 + n) * 64];
            } else {
                const float *wq = wl + (long long)q * CGU * NT * 64;      // packed [tap][cin/4][ntile][64] is chunk-linear
#pragma unroll
                for (int u = 0; u < CGU; ++u)
#pragma unroll
                    for (int n = 0; n < NTL; ++n) dstb[u][n] = wq[(u * NT + nt0 + n) * 64];
            }
        };
        auto bfrag = [&](const BW (&bw)[G4][NTL], int u, int n) -> float {
            if constexpr (W4) return bw[u >> 2][n][u & 3];
            else return bw[u][n];
        };
        // two register sets take turns (no copies): chunk q+1 is in flight while chunk q feeds the MFMAs
        auto compute = [&](const BW (&bw)[G4][NTL], int q) {
            const int j = q / CPT, cg0 = (q - j * CPT) * CGU;
            const float *arow = tile + (mbase + r + j * d) * S + g;
#pragma unroll
            for (int u = 0; u < CGU; ++u) {
                float av[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = arow[i * 16 * S + (cg0 + u) * 4];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int n = 0; n < NTL; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bfrag(bw, u, n), acc[i][n], 0, 0, 0);
            }
        };
        // The prefetch is UNCONDITIONAL (past the end it re-reads the last chunk): a load under a branch makes
        // the compiler wait with vmcnt(0) before the next MFMA, i.e. for the prefetch it has just issued, or
        // sink the load next to its use.  The register copy at the end of a chunk is where the wait belongs.
        loadb(bcur, 0);
        if constexpr (BVC_AMP_PINGPONG && C == 32) {
            // the two register sets take turns (no copies: 16 v_mov per tap otherwise); an odd last chunk is multiplied behind the loop.
            // (C = 64 with the waves along the columns spills in this form.)
            int q = 0;
#pragma unroll 1
            for (; q + 1 < nch; q += 2) {
                loadb(bnxt, q + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(bcur, q);
                __builtin_amdgcn_sched_barrier(0);
                loadb(bcur, q + 2 < nch ? q + 2 : nch - 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(bnxt, q + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q < nch) compute(bcur, q);
        } else {
#pragma unroll 1
            for (int q = 0; q < nch; ++q) {
                loadb(bnxt, q + 1 < nch ? q + 1 : nch - 1);
                __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ahead of this chunk's MFMAs
                compute(bcur, q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < G4; ++u)
#pragma unroll
                    for (int n = 0; n < NTL; ++n) bcur[u][n] = bnxt[u][n];
            }
        }
    };

    // Narrow stages (C <= 16): a conv's whole weight set is <= 44 fragments per lane, so it is fetched once
    // into registers and the tap loop is fully unrolled (no per-chunk wait on a weight load; the LDS reads
    // of later taps are scheduled under the MFMAs of earlier ones).  Same accumulation order as mma().
    auto mma_small = [&](const float *tile, int d, const float *wp, auto ks_c) {
        constexpr int KS = decltype(ks_c)::value;
        static_assert(NT == 1 || KS == 0, "mma_small is for one 16-column tile");
        float wreg[KS][C4];
        const float *wl = wp + lane;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int c = 0; c < C4; ++c) wreg[j][c] = wl[(j * C4 + c) * NT * 64];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const float *arow = tile + (mbase + r + j * d) * S + g;
#pragma unroll
            for (int c = 0; c < C4; ++c) {
                float av[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = arow[i * 16 * S + c * 4];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wreg[j][c], acc[i][0], 0, 0, 0);
            }
        }
    };
    auto conv = [&](const float *tile, int d, const float *wp) {
        if constexpr (C <= 16) {
            if (ks == 11) { mma_small(tile, d, wp, std::integral_constant<int, 11>()); return; }
            if (ks == 7) { mma_small(tile, d, wp, std::integral_constant<int, 7>()); return; }
            if (ks == 3) { mma_small(tile, d, wp, std::integral_constant<int, 3>()); return; }
        }
        mma(tile, d, wp);
    };

    // ---- phase 2: u = conv1(S1(x)) ; t2 = S2(u + b1), zero before the start of the signal
    conv(t1, dil, a.w1);
    PHASE(1);
    if (ALIAS) __syncthreads();                            // every wave is done with S1(x): its LDS becomes t2
    for (int idx = tid; idx < (ks - 1) * S; idx += 256) t2[TR * S + idx] = 0.0f;    // spare rows read by discarded outputs
    // local rows before `zrow` lie before the start of the signal: zero there (the reference pads AFTER the activation)
    const long long zr64 = -(tbase + a.t_origin);
    const int zrow = zr64 <= 0 ? 0 : (zr64 > TR ? TR : (int)zr64);
    if constexpr (C == 8) {
        // Only 8 of the tile's 16 columns exist: lanes r >= 8 hold padding.  They take over rows g*4+2, g*4+3 of
        // column r-8 from their neighbour 8 lanes down (DPP row_shr:8), so that every lane evaluates ONE SnakeBeta
        // pair per row tile instead of half the lanes evaluating two.
        const int col = r & 7, e0 = (r >> 3) * 2;
        const float bias = a.b1[col], aa = a.a2[col], bb = a.ib2[col];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float a0 = acc[i][0][0], a1 = acc[i][0][1], a2 = acc[i][0][2], a3 = acc[i][0][3];
            const float hi0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), 0x118, 0xf, 0xf, false));   // row_shr:8
            const float hi1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a3), 0x118, 0xf, 0xf, false));
            const float v0 = r < 8 ? a0 : hi0, v1 = r < 8 ? a1 : hi1;
            const int row = mbase + i * 16 + g * 4 + e0;
            const f32x2 s2 = snakebeta2((f32x2){v0 + bias, v1 + bias}, splat2(aa), splat2(bb));
            t2[row * S + col] = row >= zrow ? s2[0] : 0.0f;
            t2[(row + 1) * S + col] = row + 1 >= zrow ? s2[1] : 0.0f;
        }
    } else {
        // (all tiles but the first of a signal lie wholly inside it: no row to zero - two selects per pair less; the test is uniform)
        auto s2_tile = [&](auto edge_c) {
            constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
                const int col = (nt0 + n) * 16 + r;
                if (col < C) {
                    const float bias = a.b1[col], aa = a.a2[col], bb = a.ib2[col];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const int row = mbase + i * 16 + g * 4 + e;
                            const f32x2 u2 = (f32x2){acc[i][n][e] + bias, acc[i][n][e + 1] + bias};
                            const f32x2 s2 = snakebeta2(u2, splat2(aa), splat2(bb));
                            t2[row * S + col] = (!EDGE || row >= zrow) ? s2[0] : 0.0f;
                            t2[(row + 1) * S + col] = (!EDGE || row + 1 >= zrow) ? s2[1] : 0.0f;
                        }
                }
            }
        };
        if (zrow > 0) s2_tile(std::true_type());
        else          s2_tile(std::false_type());
    }
    __syncthreads();
    PHASE(2);

    // ---- phase 3: x' = conv2(t2) + b2 + x   (+ running sum over the AMP blocks, / num_kernels)
    // The residual (and running-sum) operands are fetched BEFORE the MFMA loop so that their latency is
    // covered by it instead of being exposed in the epilogue.
    // The rows of a workgroup are consecutive and C is the whole row, so its output (and the residual /
    // running-sum operands) is ONE contiguous span of global memory: it is moved as float4 per lane, with
    // the conv2 result transposed from the MFMA layout through LDS (the S1(x) tile is dead by now).
    const long long ob = (long long)b * a.bs + t0 * C;
    constexpr int NLD3 = (TR * C4 + 255) / 256;
    const long long rows_left = a.L - t0;
    const int nvalid4 = (int)(rows_left < TT ? rows_left : TT) * C4;
    f32x4 resq[NLD3], accq[NLD3];
    const bool with_acc = a.epi >= CE_RES_ACC;             // (uniform)
#pragma unroll
    for (int i = 0; i < NLD3; ++i) {                       // unconditional, clamped: items past the tile's valid rows are never stored
        const int idx = tid + i * 256;
        const int idc = idx < nvalid4 ? idx : 0;
        resq[i] = reinterpret_cast<const f32x4 *>(a.x + ob)[idc];
        accq[i] = with_acc ? reinterpret_cast<const f32x4 *>(a.acc + ob)[idc] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    PHASE(3);
    conv(t2, 1, a.w2);
    PHASE(4);
    if (ALIAS) __syncthreads();                            // t2 is dead: the same LDS now stages the output tile
#pragma unroll
    for (int n = 0; n < NTL; ++n) {
        const int col = (nt0 + n) * 16 + r;
        if (col >= C) continue;
        const float bias = a.b2[col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) t1[(mbase + i * 16 + g * 4 + e) * S + col] = acc[i][n][e] + bias;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD3; ++i) {
        const int idx = tid + i * 256;
        if (idx >= nvalid4) continue;
        const int row = idx / C4, c4 = idx - row * C4;
        const float2 lo = *reinterpret_cast<const float2 *>(t1 + row * S + c4 * 4);
        const float2 hi = *reinterpret_cast<const float2 *>(t1 + row * S + c4 * 4 + 2);
        f32x4 v = (f32x4){lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float o = v[e] + resq[i][e];                             // x = xt + x      (models.py:119)
            if (a.epi >= CE_RES_ACC) o = accq[i][e] + o;             // xs += resblock  (models.py:224)
            v[e] = o;
        }
        divide_if(a.epi == CE_RES_ACC_DIV, v, a.divisor);            // xs / num_kernels (models.py:225)
        reinterpret_cast<f32x4 *>(a.out + ob)[idx] = v;
    }
    PHASE(5);
}

// ------------------------------------------------------------------------------------------------
// The C = 8 stage (the last and longest signal: 256 rows per frame) on FULL MFMA tiles.  With eight channels a 16-column
// tile of the kernel above is half padding.  Here the 16 columns are (p, co): output rows t and t + d (d = the conv's
// dilation) side by side, p = 0 / 1.  Row t + p*d reads x[t + p*d - m*d] = x[t - (m - p)*d], so both rows read the same
// ks + 1 input rows t - l*d, l = -1 .. ks-1, and the B matrix holds W_j in the p = 0 columns and W_(j-1) in the p = 1
// columns of k-step j (zero where that runs off the kernel): (ks + 1) / (2 ks) of the MFMAs of the padded form, and every
// lane of the SnakeBeta epilogue has work.  A column still accumulates its taps in the order j = 0 .. ks-1, input
// channels ascending (the added zero products come first or last), so results equal the generic kernel's bit for bit.
// Rows are paired inside blocks of 2d rows: pair m <-> rows R(m) + p*d, R(m) = (m / d) * 2d + m % d.  The LDS tiles are
// stored de-interleaved to match: row rho = 2d*blk + half*d + i sits at position half * H + blk*d + i, which puts the A
// operand of pair m, k-step k' = 2a + b at position m + a*d + b*H: lane stride = one row (S = 10 floats: conflict-free
// ds_read_b32), one compile-time offset per k-step.
// which of the stage-specific kernels a model starts with (bvc_model_set_option "vocoder_full_tiles" / "vocoder_c16_kernel": validation switches)
unsigned amp_kernels_default() {
    return (getenv("BVC_NO_AMP8") == nullptr ? AMPK_C8 : 0u) | (getenv("BVC_NO_AMP16") == nullptr ? AMPK_C16 : 0u);
}

template <int D>
__device__ __forceinline__ int pair_row(int m) { return (m / D) * (2 * D) + m % D; }
template <int D>
__device__ __forceinline__ int row_pos(int rho, int H) {
    const int blk = rho / (2 * D), w = rho - blk * (2 * D);
    return w >= D ? H + blk * D + (w - D) : blk * D + w;
}
template <int KS, int D, int MT2>
struct Amp8Geom {
    static constexpr int NP = 4 * MT2 * 16;                     // row pairs per conv phase and workgroup
    static constexpr int NPE = (NP / D) * D;                    // pairs in whole blocks (conv1)
    static constexpr int TR1 = 2 * NPE;                         // rows conv1 produces: [tbase, tbase + TR1)
    static constexpr int TR = 2 * NP;                           // rows conv2 sweeps
    static constexpr int TT = TR1 - (KS - 1);                   // valid output rows per workgroup
    static constexpr int HALO1 = (KS - 1) * D;
    static constexpr int ROWS1 = TR1 + HALO1;                   // S1(x) rows [tbase - HALO1, tbase + TR1)
    static constexpr int H1 = NP + ((KS + 1) / 2) * D;          // half size of the S1(x) tile (positions)
    static constexpr int H2 = NP + (KS + 1) / 2;                // half size of the S2(u) tile
    static constexpr int S = 10;
    static constexpr int LDS_ROWS = 2 * H1 > TR ? 2 * H1 : TR;  // (2 * H2 <= 2 * H1)
    static constexpr size_t LDS_BYTES = (size_t)LDS_ROWS * S * sizeof(float);
};

// The kernel is persistent: a workgroup keeps both convs' weights, biases and SnakeBeta parameters in registers and walks
// over tiles; the input rows of its NEXT tile are requested as soon as the registers that held the current ones are free
// (after S1), so that their latency lies under the current tile's two convs instead of in front of every tile.
template <int KS, int D, int MT2, int OCC>
__global__ __launch_bounds__(256, OCC) void amp_pair8_kernel(AmpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = Amp8Geom<KS, D, MT2>;
    constexpr int C = 8, S = G::S, NP = G::NP, NPE = G::NPE, TR = G::TR, TT = G::TT, H1 = G::H1, H2 = G::H2;
    constexpr int NLD = (G::ROWS1 * 2 + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    // Workgroups are dealt round-robin to the 8 XCDs; neighbouring tiles share their halo rows, so each XCD takes a
    // contiguous run of `per` tiles (the halo then hits in that XCD's L2) and its workgroups stride through the run.
    const unsigned per = (a.ntile + 7u) >> 3, nli = gridDim.x >> 3;
    const unsigned xcd = blockIdx.x & 7u;
    unsigned li = blockIdx.x >> 3;
    const unsigned run_end = (xcd + 1u) * per < a.ntile ? (xcd + 1u) * per : a.ntile;
    if (xcd * per + li >= run_end) return;

    float w1reg[KS + 1][2], w2reg[KS + 1][2];
#pragma unroll
    for (int k = 0; k <= KS; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w1reg[k][q] = a.w1[(k * 2 + q) * 64 + lane];
            w2reg[k][q] = a.w2[(k * 2 + q) * 64 + lane];
        }
    const f32x4 aa1 = *reinterpret_cast<const f32x4 *>(a.a1 + (tid & 1) * 4);      // phase 1: item idx has channels (idx & 1) * 4 ..; idx & 1 == tid & 1
    const f32x4 bb1 = *reinterpret_cast<const f32x4 *>(a.ib1 + (tid & 1) * 4);
    // The MFMA operands are swapped (weights as A, activations as B): acc[i][e] = out[pair mbase + 16 i + r][column 4 g + e], column =
    // p * 8 + co - a lane's four results are four consecutive channels of ONE row, so the epilogues work on 16-byte granules
    const int p = g >> 1, co0 = (g & 1) * 4;
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(a.b1 + co0), aa2 = *reinterpret_cast<const f32x4 *>(a.a2 + co0);
    const f32x4 bb2 = *reinterpret_cast<const f32x4 *>(a.ib2 + co0), bias2 = *reinterpret_cast<const f32x4 *>(a.b2 + co0);

    auto tile_origin = [&](unsigned bid, int &b, long long &t0) {
        b = a.tiles_per_batch == 1 ? (int)bid : (int)div_tpb(bid, a.tpb_magic);
        t0 = a.row_begin + (long long)(bid - (unsigned)b * (unsigned)a.tiles_per_batch) * TT;
    };
    auto load_rows = [&](unsigned bid, f32x4 (&v)[NLD]) {       // x rows [t0 - (KS-1) - HALO1, .. + ROWS1) of tile bid
        int b; long long t0;
        tile_origin(bid, b, t0);
        const __amdgpu_buffer_rsrc_t rs = rows_rsrc(a.x + (long long)b * a.bs, a.L, C);
        const int tfirst = (int)(t0 - (KS - 1) - G::HALO1);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {                        // (items past the tile's rows: loaded like the others, never parked)
            const int idx = tid + i * 256;
            v[i] = rows_load4(rs, tfirst + (idx >> 1), C, (idx & 1) * 4);
        }
    };

    const int mbase = wave * MT2 * 16;
    f32x4 acc[MT2];
    // one conv on row pairs: KS + 1 k-steps of two MFMAs (input channels 0-3, 4-7); all offsets are compile-time
    auto conv = [&](auto dd, int H, const float (&wreg)[KS + 1][2]) {
        constexpr int DD = decltype(dd)::value;
        const float *arow = lds + (mbase + r) * S + g;
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k <= KS; ++k) {
            const int pos = (k >> 1) * DD + (k & 1) * H;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float av[MT2];
#pragma unroll
                for (int i = 0; i < MT2; ++i) av[i] = arow[(pos + i * 16) * S + q * 4];
#pragma unroll
                for (int i = 0; i < MT2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[k][q], av[i], acc[i], 0, 0, 0);      // tile of out^T: see the epilogues
            }
        }
    };

    f32x4 v[NLD];
    load_rows(xcd * per + li, v);
    for (;;) {
        const unsigned bid = xcd * per + li;
        int b; long long t0;
        tile_origin(bid, b, t0);
        const long long tbase = t0 - (KS - 1);             // global row of conv1's local output row 0

        // ---- phase 1: S1(x) rows [tbase - HALO1, tbase + TR1) into LDS, de-interleaved by D
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            if (idx < G::ROWS1 * 2) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {                                        // S(0) = 0 keeps the zero padding
                    const f32x2 o2 = snakebeta2((f32x2){v[i][e], v[i][e + 1]}, (f32x2){aa1[e], aa1[e + 1]}, (f32x2){bb1[e], bb1[e + 1]});
                    o[e] = o2[0]; o[e + 1] = o2[1];
                }
                float2 *dst = reinterpret_cast<float2 *>(lds + row_pos<D>(idx >> 1, H1) * S + (idx & 1) * 4);
                dst[0] = make_float2(o[0], o[1]);
                dst[1] = make_float2(o[2], o[3]);
            }
        }
        li += nli;
        const bool more = xcd * per + li < run_end;        // (uniform)
        if (more) load_rows(xcd * per + li, v);             // the next tile's rows travel under this tile's convs
        __syncthreads();

        // ---- phase 2: u = conv1(S1(x)); the tile of S2(u + b1) (de-interleaved by 1) takes over the LDS
        conv(std::integral_constant<int, D>(), H1, w1reg);
        __syncthreads();                                   // every wave is done with S1(x)
        {
            const long long zr64 = -(tbase + a.t_origin);  // local rows before zrow lie before the start of the signal: zero
            const int zrow = zr64 <= 0 ? 0 : (zr64 > TR ? TR : (int)zr64);      // (the reference pads AFTER the activation)
            auto s2_tile = [&](auto edge_c) {                  // (only a signal's first tile has rows to zero: the test is uniform)
                constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
                for (int i = 0; i < MT2; ++i) {
                    const int m = mbase + i * 16 + r;      // this lane's pair; its row of column block p
                    const int row = pair_row<D>(m) + p * D;
                    const f32x4 u4 = acc[i] + bias1;
                    const f32x2 s01 = snakebeta2((f32x2){u4[0], u4[1]}, (f32x2){aa2[0], aa2[1]}, (f32x2){bb2[0], bb2[1]});
                    const f32x2 s23 = snakebeta2((f32x2){u4[2], u4[3]}, (f32x2){aa2[2], aa2[3]}, (f32x2){bb2[2], bb2[3]});
                    const bool keep = !EDGE || row >= zrow;
                    if (NPE == NP || m < NPE) {
                        float2 *dst = reinterpret_cast<float2 *>(lds + row_pos<1>(row, H2) * S + co0);
                        dst[0] = keep ? make_float2(s01[0], s01[1]) : make_float2(0.f, 0.f);
                        dst[1] = keep ? make_float2(s23[0], s23[1]) : make_float2(0.f, 0.f);
                    }
                }
            };
            if (zrow > 0) s2_tile(std::true_type());
            else          s2_tile(std::false_type());
            // rows conv1 did not produce ([TR1, TR + KS]): read only by discarded outputs, but keep them defined
            for (int idx = tid; idx < (TR + KS + 1 - G::TR1) * C; idx += 256) {
                const int row = G::TR1 + idx / C;
                if (row_pos<1>(row, H2) < 2 * H2) lds[row_pos<1>(row, H2) * S + (idx % C)] = 0.0f;
            }
        }
        __syncthreads();

        // ---- phase 3: x' = conv2(t2) + b2 + x (+ running sum, / num_kernels); operands requested before the MFMAs
        const long long ob = (long long)b * a.bs + t0 * C;
        constexpr int NLD3 = (TT * 2 + 255) / 256;
        const long long rows_left = a.L - t0;
        const int nvalid4 = (int)(rows_left < TT ? rows_left : TT) * 2;
        f32x4 resq[NLD3], accq[NLD3];
#pragma unroll
        for (int i = 0; i < NLD3; ++i) {
            const int idx = tid + i * 256;
            const bool ok = idx < nvalid4;
            resq[i] = ok ? reinterpret_cast<const f32x4 *>(a.x + ob)[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
            accq[i] = (ok && a.epi >= CE_RES_ACC) ? reinterpret_cast<const f32x4 *>(a.acc + ob)[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        conv(std::integral_constant<int, 1>(), H2, w2reg);
        __syncthreads();                                   // t2 is dead: the LDS now stages the output tile, row-major
#pragma unroll
        for (int i = 0; i < MT2; ++i) {
            const f32x4 o4 = acc[i] + bias2;
            float2 *dst = reinterpret_cast<float2 *>(lds + (2 * (mbase + i * 16 + r) + p) * S + co0);
            dst[0] = make_float2(o4[0], o4[1]);
            dst[1] = make_float2(o4[2], o4[3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NLD3; ++i) {
            const int idx = tid + i * 256;
            if (idx >= nvalid4) continue;
            const float2 lo = *reinterpret_cast<const float2 *>(lds + (idx >> 1) * S + (idx & 1) * 4);
            const float2 hi = *reinterpret_cast<const float2 *>(lds + (idx >> 1) * S + (idx & 1) * 4 + 2);
            f32x4 o4 = (f32x4){lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float o = o4[e] + resq[i][e];                            // x = xt + x      (models.py:119)
                if (a.epi >= CE_RES_ACC) o = accq[i][e] + o;             // xs += resblock  (models.py:224)
                o4[e] = o;
            }
            divide_if(a.epi == CE_RES_ACC_DIV, o4, a.divisor);           // xs / num_kernels (models.py:225)
            reinterpret_cast<f32x4 *>(a.out + ob)[idx] = o4;
        }
        if (!more) break;
        __syncthreads();                                   // the staging rows are read: the next tile's S1(x) may overwrite them
    }
}

// workgroups of one instance the device holds at once (first call: conv_kernels_init, outside any stream capture)
template <int KS, int D, int MT2, int OCC>
static int amp8_slots() {
    static int slots = 0;
    if (!slots) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(amp_pair8_kernel<KS, D, MT2, OCC>), 256,
                                                         Amp8Geom<KS, D, MT2>::LDS_BYTES) != hipSuccess) {
            set_error("amp_pair8: occupancy query failed");
            return -1;
        }
        slots = (per_cu > 0 ? per_cu : 1) * cus;
    }
    return slots;
}
template <int MT2, int OCC>
static bool amp8_slots_all() {
    return amp8_slots<3, 1, MT2, OCC>() > 0 && amp8_slots<3, 3, MT2, OCC>() > 0 && amp8_slots<3, 5, MT2, OCC>() > 0 &&
           amp8_slots<7, 1, MT2, OCC>() > 0 && amp8_slots<7, 3, MT2, OCC>() > 0 && amp8_slots<7, 5, MT2, OCC>() > 0 &&
           amp8_slots<11, 1, MT2, OCC>() > 0 && amp8_slots<11, 3, MT2, OCC>() > 0 && amp8_slots<11, 5, MT2, OCC>() > 0;
}

template <int KS, int D, int MT2, int OCC>
static int launch_amp8_t(AmpArgs a, int B, hipStream_t s) {
    using G = Amp8Geom<KS, D, MT2>;
    a.tiles_per_batch = (int)((a.L - a.row_begin + G::TT - 1) / G::TT);
    if (a.tiles_per_batch <= 0) return BVC_OK;
    a.tpb_magic = tpb_magic_of((unsigned)a.tiles_per_batch);
    if ((unsigned long long)a.tiles_per_batch * a.tiles_per_batch * (unsigned long long)B >= 0x100000000ull) { set_error("vocoder: tile count beyond the reciprocal's range"); return BVC_EINVAL; }
    const int slots = amp8_slots<KS, D, MT2, OCC>();
    if (slots <= 0) return BVC_EHIP;
    ProbeScope probe(PK_CONV, s);
    const unsigned ntile = (unsigned)(a.tiles_per_batch * (long long)B);
    a.ntile = ntile;
    const unsigned grid = ntile < (unsigned)slots ? ((ntile + 7u) & ~7u) : ((unsigned)slots & ~7u);
    hipLaunchKernelGGL((amp_pair8_kernel<KS, D, MT2, OCC>), dim3(grid), dim3(256), G::LDS_BYTES, s, a);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}
template <int MT2, int OCC>
static int launch_amp8(AmpArgs a, int B, hipStream_t s) {
    switch (a.ks * 8 + a.dil) {
        case 3 * 8 + 1:  return launch_amp8_t<3, 1, MT2, OCC>(a, B, s);
        case 3 * 8 + 3:  return launch_amp8_t<3, 3, MT2, OCC>(a, B, s);
        case 3 * 8 + 5:  return launch_amp8_t<3, 5, MT2, OCC>(a, B, s);
        case 7 * 8 + 1:  return launch_amp8_t<7, 1, MT2, OCC>(a, B, s);
        case 7 * 8 + 3:  return launch_amp8_t<7, 3, MT2, OCC>(a, B, s);
        case 7 * 8 + 5:  return launch_amp8_t<7, 5, MT2, OCC>(a, B, s);
        case 11 * 8 + 1: return launch_amp8_t<11, 1, MT2, OCC>(a, B, s);
        case 11 * 8 + 3: return launch_amp8_t<11, 3, MT2, OCC>(a, B, s);
        case 11 * 8 + 5: return launch_amp8_t<11, 5, MT2, OCC>(a, B, s);
        default: return -1;                                // not one of the generator's shapes: the generic kernel takes it
    }
}

// ------------------------------------------------------------------------------------------------
// The C = 16 stage on a kernel of its own, in the manner of amp_pair8_kernel: persistent (a workgroup keeps both convs' weights,
// the biases and the SnakeBeta parameters in registers and walks over tiles; the input rows of its NEXT tile travel under the
// current tile's convs), taps and dilation at compile time, and the MFMA operands swapped (weights as A, activations as B), so
// that a lane's four results are four consecutive channels of ONE row: the S2 tile is written as one 16-byte LDS store per row
// tile and the output (with its residual / running-sum operands) moves as 16 bytes per lane straight from the accumulators -
// a wave's 16 rows x 64 bytes are one contiguous KiB - with no transposition through LDS.  Row stride 20 floats: the B-operand
// reads of a wave (16 rows x 4 channels) fall into 64 different banks, and rows stay 16-byte aligned.  Two barriers per tile (the
// S1 and S2 tiles do not share LDS).  Same taps, same k order per output as amp_pair_kernel<16, ...>: the same bits
// (tests/test_gpu_parity.py::test_vocoder_c16_kernel_equals_generic).
template <int KS, int D, int MT>
struct Amp16Geom {
    static constexpr int S = 20;
    static constexpr int TR = 4 * MT * 16;                      // rows per conv phase and workgroup
    static constexpr int TT = TR - (KS - 1);                    // valid output rows
    static constexpr int HALO1 = (KS - 1) * D;
    static constexpr int ROWS1 = TR + HALO1;                    // S1(x) rows [tbase - HALO1, tbase + TR)
    static constexpr int ROWS2 = TR + KS - 1;                   // S2(u) rows [tbase, tbase + TR) + spare rows read by discarded outputs
    static constexpr size_t LDS_BYTES = (size_t)(ROWS1 + ROWS2) * S * sizeof(float);
};

template <int KS, int D, int MT, int OCC>
__global__ __launch_bounds__(256, OCC) void amp_pair16_kernel(AmpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = Amp16Geom<KS, D, MT>;
    constexpr int C = 16, C4 = 4, S = G::S, TR = G::TR, TT = G::TT;
    constexpr int NLD = (G::ROWS1 * C4 + 255) / 256;
    float *t1 = lds, *t2 = lds + G::ROWS1 * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    // tiles: as in amp_pair8_kernel (each XCD a contiguous run, its workgroups stride through it)
    const unsigned per = (a.ntile + 7u) >> 3, nli = gridDim.x >> 3;
    const unsigned xcd = blockIdx.x & 7u;
    unsigned li = blockIdx.x >> 3;
    const unsigned run_end = (xcd + 1u) * per < a.ntile ? (xcd + 1u) * per : a.ntile;
    if (xcd * per + li >= run_end) return;

    float w1reg[KS][C4], w2reg[KS][C4];
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            w1reg[j][c] = a.w1[(j * C4 + c) * 64 + lane];
            w2reg[j][c] = a.w2[(j * C4 + c) * 64 + lane];
        }
    const f32x4 aa1 = *reinterpret_cast<const f32x4 *>(a.a1 + (tid & 3) * 4);       // phase 1: item idx has channels (idx & 3) * 4 ..; idx & 3 == tid & 3
    const f32x4 bb1 = *reinterpret_cast<const f32x4 *>(a.ib1 + (tid & 3) * 4);
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(a.b1 + g * 4), aa2 = *reinterpret_cast<const f32x4 *>(a.a2 + g * 4);
    const f32x4 bb2 = *reinterpret_cast<const f32x4 *>(a.ib2 + g * 4), bias2 = *reinterpret_cast<const f32x4 *>(a.b2 + g * 4);
    for (int idx = tid; idx < (KS - 1) * S; idx += 256) t2[TR * S + idx] = 0.0f;     // spare rows: never written again

    auto tile_origin = [&](unsigned bid, int &b, long long &t0) {
        b = a.tiles_per_batch == 1 ? (int)bid : (int)div_tpb(bid, a.tpb_magic);
        t0 = a.row_begin + (long long)(bid - (unsigned)b * (unsigned)a.tiles_per_batch) * TT;
    };
    auto load_rows = [&](unsigned bid, f32x4 (&v)[NLD]) {       // x rows [t0 - (KS-1) - HALO1, .. + ROWS1) of tile bid
        int b; long long t0;
        tile_origin(bid, b, t0);
        const __amdgpu_buffer_rsrc_t rs = rows_rsrc(a.x + (long long)b * a.bs, a.L, C);
        const int tfirst = (int)(t0 - (KS - 1) - G::HALO1);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {                        // (items past the tile's rows: loaded like the others, never parked)
            const int idx = tid + i * 256;
            v[i] = rows_load4(rs, tfirst + (idx >> 2), C, (idx & 3) * 4);
        }
    };

    const int mbase = wave * MT * 16;
    f32x4 acc[MT];
    // acc[i][e] = out[row mbase + 16 i + r][channel 4 g + e]; tap j of output row m reads tile row m + j * DD
    auto conv = [&](auto dd, const float *tile, const float (&wreg)[KS][C4]) {
        constexpr int DD = decltype(dd)::value;
        const float *brow = tile + (mbase + r) * S + g;
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int c = 0; c < C4; ++c) {
                float bv[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) bv[i] = brow[(j * DD + i * 16) * S + c * 4];
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][c], bv[i], acc[i], 0, 0, 0);
            }
    };

    f32x4 v[NLD];
    load_rows(xcd * per + li, v);
    for (;;) {
        const unsigned bid = xcd * per + li;
        int b; long long t0;
        tile_origin(bid, b, t0);
        const long long tbase = t0 - (KS - 1);             // global row of conv1's local output row 0

        // ---- phase 1: S1(x) rows [tbase - HALO1, tbase + TR) into LDS
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            if (idx < G::ROWS1 * C4) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {                                        // S(0) = 0 keeps the zero padding
                    const f32x2 o2 = snakebeta2((f32x2){v[i][e], v[i][e + 1]}, (f32x2){aa1[e], aa1[e + 1]}, (f32x2){bb1[e], bb1[e + 1]});
                    o[e] = o2[0]; o[e + 1] = o2[1];
                }
                *reinterpret_cast<f32x4 *>(t1 + (idx >> 2) * S + (idx & 3) * 4) = o;
            }
        }
        li += nli;
        const bool more = xcd * per + li < run_end;        // (uniform)
        if (more) load_rows(xcd * per + li, v);             // the next tile's rows travel under this tile's convs
        __syncthreads();

        // ---- phase 2: u = conv1(S1(x)); S2(u + b1) into its own tile, zero before the start of the signal
        conv(std::integral_constant<int, D>(), t1, w1reg);
        {
            const long long zr64 = -(tbase + a.t_origin);  // local rows before zrow lie before the start of the signal
            const int zrow = zr64 <= 0 ? 0 : (zr64 > TR ? TR : (int)zr64);      // (the reference pads AFTER the activation)
            auto s2_tile = [&](auto edge_c) {                  // (only a signal's first tile has rows to zero: the test is uniform)
                constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = mbase + i * 16 + r;
                    const f32x4 u4 = acc[i] + bias1;
                    const f32x2 s01 = snakebeta2((f32x2){u4[0], u4[1]}, (f32x2){aa2[0], aa2[1]}, (f32x2){bb2[0], bb2[1]});
                    const f32x2 s23 = snakebeta2((f32x2){u4[2], u4[3]}, (f32x2){aa2[2], aa2[3]}, (f32x2){bb2[2], bb2[3]});
                    const bool keep = !EDGE || row >= zrow;
                    *reinterpret_cast<f32x4 *>(t2 + row * S + g * 4) = keep ? (f32x4){s01[0], s01[1], s23[0], s23[1]} : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            };
            if (zrow > 0) s2_tile(std::true_type());
            else          s2_tile(std::false_type());
        }
        __syncthreads();

        // ---- phase 3: x' = conv2(t2) + b2 + x (+ running sum, / num_kernels): operands requested before the MFMAs, 16 bytes per
        // lane, straight from / to the accumulator layout (output row m of this lane: global row t0 + m)
        const long long ob = (long long)b * a.bs + t0 * C;
        const long long rows_left = a.L - t0;
        const int nvalid = (int)(rows_left < TT ? rows_left : TT);
        f32x4 resq[MT], accq[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = mbase + i * 16 + r;
            const bool ok = row < nvalid;
            resq[i] = ok ? *reinterpret_cast<const f32x4 *>(a.x + ob + row * C + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            accq[i] = (ok && a.epi >= CE_RES_ACC) ? *reinterpret_cast<const f32x4 *>(a.acc + ob + row * C + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        conv(std::integral_constant<int, 1>(), t2, w2reg);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = mbase + i * 16 + r;
            if (row >= nvalid) continue;
            f32x4 o4 = acc[i] + bias2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float o = o4[e] + resq[i][e];                            // x = xt + x      (models.py:119)
                if (a.epi >= CE_RES_ACC) o = accq[i][e] + o;             // xs += resblock  (models.py:224)
                o4[e] = o;
            }
            divide_if(a.epi == CE_RES_ACC_DIV, o4, a.divisor);           // xs / num_kernels (models.py:225)
            *reinterpret_cast<f32x4 *>(a.out + ob + row * C + g * 4) = o4;
        }
        if (!more) break;
    }
}

template <int KS, int D, int MT, int OCC>
static int amp16_slots() {
    static int slots = 0;
    if (!slots) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(amp_pair16_kernel<KS, D, MT, OCC>), 256,
                                                         Amp16Geom<KS, D, MT>::LDS_BYTES) != hipSuccess) {
            set_error("amp_pair16: occupancy query failed");
            return -1;
        }
        slots = (per_cu > 0 ? per_cu : 1) * cus;
    }
    return slots;
}
template <int KS, int D, int MT, int OCC>
static int launch_amp16_t(AmpArgs a, int B, hipStream_t s) {
    using G = Amp16Geom<KS, D, MT>;
    a.tiles_per_batch = (int)((a.L - a.row_begin + G::TT - 1) / G::TT);
    if (a.tiles_per_batch <= 0) return BVC_OK;
    a.tpb_magic = tpb_magic_of((unsigned)a.tiles_per_batch);
    if ((unsigned long long)a.tiles_per_batch * a.tiles_per_batch * (unsigned long long)B >= 0x100000000ull) { set_error("vocoder: tile count beyond the reciprocal's range"); return BVC_EINVAL; }
    const int slots = amp16_slots<KS, D, MT, OCC>();
    if (slots <= 0) return BVC_EHIP;
    ProbeScope probe(PK_CONV, s);
    const unsigned ntile = (unsigned)(a.tiles_per_batch * (long long)B);
    a.ntile = ntile;
    const unsigned grid = ntile < (unsigned)slots ? ((ntile + 7u) & ~7u) : ((unsigned)slots & ~7u);
    hipLaunchKernelGGL((amp_pair16_kernel<KS, D, MT, OCC>), dim3(grid), dim3(256), G::LDS_BYTES, s, a);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}
// OCC: the register budget the taps leave (both convs' weights live in registers: 8 KS floats per lane)
template <int KS, int MT> struct Amp16Occ { static constexpr int V = MT >= 4 ? 2 : (KS == 3 ? 4 : KS == 7 ? 3 : 2); };
template <int MT>
static int launch_amp16(AmpArgs a, int B, hipStream_t s) {
    switch (a.ks * 8 + a.dil) {
        case 3 * 8 + 1:  return launch_amp16_t<3, 1, MT, Amp16Occ<3, MT>::V>(a, B, s);
        case 3 * 8 + 3:  return launch_amp16_t<3, 3, MT, Amp16Occ<3, MT>::V>(a, B, s);
        case 3 * 8 + 5:  return launch_amp16_t<3, 5, MT, Amp16Occ<3, MT>::V>(a, B, s);
        case 7 * 8 + 1:  return launch_amp16_t<7, 1, MT, Amp16Occ<7, MT>::V>(a, B, s);
        case 7 * 8 + 3:  return launch_amp16_t<7, 3, MT, Amp16Occ<7, MT>::V>(a, B, s);
        case 7 * 8 + 5:  return launch_amp16_t<7, 5, MT, Amp16Occ<7, MT>::V>(a, B, s);
        case 11 * 8 + 1: return launch_amp16_t<11, 1, (MT > 4 ? 4 : MT), Amp16Occ<11, MT>::V>(a, B, s);      // (88 weight registers: four row tiles at most)
        case 11 * 8 + 3: return launch_amp16_t<11, 3, (MT > 4 ? 4 : MT), Amp16Occ<11, MT>::V>(a, B, s);      // (88 weight registers: four row tiles at most)
        case 11 * 8 + 5: return launch_amp16_t<11, 5, (MT > 4 ? 4 : MT), Amp16Occ<11, MT>::V>(a, B, s);      // (88 weight registers: four row tiles at most)
        default: return -1;                                // not one of the generator's shapes: the generic kernel takes it
    }
}
template <int MT>
static bool amp16_slots_all() {
    return amp16_slots<3, 1, MT, Amp16Occ<3, MT>::V>() > 0 && amp16_slots<3, 3, MT, Amp16Occ<3, MT>::V>() > 0 && amp16_slots<3, 5, MT, Amp16Occ<3, MT>::V>() > 0 &&
           amp16_slots<7, 1, MT, Amp16Occ<7, MT>::V>() > 0 && amp16_slots<7, 3, MT, Amp16Occ<7, MT>::V>() > 0 && amp16_slots<7, 5, MT, Amp16Occ<7, MT>::V>() > 0 &&
           amp16_slots<11, 1, (MT > 4 ? 4 : MT), Amp16Occ<11, MT>::V>() > 0 && amp16_slots<11, 3, (MT > 4 ? 4 : MT), Amp16Occ<11, MT>::V>() > 0 &&
           amp16_slots<11, 5, (MT > 4 ? 4 : MT), Amp16Occ<11, MT>::V>() > 0;
}

template <int C, int MT, int OCC, bool ALIAS, int CS = 1>
static int launch_amp_t(AmpArgs a, int B, hipStream_t s) {
    constexpr int TR = (4 / CS) * MT * 16;
    const int TT = TR - (a.ks - 1);
    a.tiles_per_batch = (int)((a.L - a.row_begin + TT - 1) / TT);
    if (a.tiles_per_batch <= 0) return BVC_OK;
    a.tpb_magic = tpb_magic_of((unsigned)a.tiles_per_batch);
    if ((unsigned long long)a.tiles_per_batch * a.tiles_per_batch * (unsigned long long)B >= 0x100000000ull) { set_error("vocoder: tile count beyond the reciprocal's range"); return BVC_EINVAL; }
    const size_t lds = (size_t)((TR + (a.ks - 1) * a.dil) + (ALIAS ? 0 : TR + (a.ks - 1))) * (C + 2) * sizeof(float);
    if (lds > 160 * 1024 || TT <= 0) { set_error("amp_pair tile needs %zu B of LDS", lds); return BVC_EINVAL; }
    ProbeScope probe(PK_CONV, s);
    // grid rounded up to a multiple of 8 so that the XCD-contiguous renumbering covers every tile exactly once
    const unsigned ntile = (unsigned)(a.tiles_per_batch * (long long)B);
    a.ntile = ntile;
    {
        static bool attr = false;
        if (!attr) { BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(amp_pair_kernel<C, MT, OCC, ALIAS, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    }
    hipLaunchKernelGGL((amp_pair_kernel<C, MT, OCC, ALIAS, CS>), dim3((ntile + 7u) & ~7u), dim3(256), lds, s, a);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

int launch_amp_pair(const ConvLayer &c1, const ConvLayer &c2, const float *x, long long L, float *out, int B, int epi,
                    const float *acc, float divisor, hipStream_t s, const ConvWindow *win, unsigned kernels) {
    if (B <= 0 || L <= 0) return BVC_OK;
    if (c1.cin != c1.cout || c2.cin != c1.cin || c2.ks != c1.ks || c2.dil != 1 || !c1.act_a || !c2.act_a) {
        set_error("amp_pair: unsupported layer pair");
        return BVC_EINVAL;
    }
    AmpArgs a;
    a.x = x; a.out = out; a.acc = acc; a.L = L;
    a.w1 = c1.wp; a.b1 = c1.bias; a.a1 = c1.act_a; a.ib1 = c1.act_ib;
    a.w2 = c2.wp; a.b2 = c2.bias; a.a2 = c2.act_a; a.ib2 = c2.act_ib;
    if (c1.cin >= 32) {                                    // amp_pair_kernel<32 / 64, ...> streams its weights in 16-byte granules
        if (!c1.wp4 || !c2.wp4) { set_error("amp_pair: layer pair without the 16-byte weight packing"); return BVC_EINVAL; }
        a.w1 = c1.wp4; a.w2 = c2.wp4;
    }
    a.divisor = divisor; a.epi = epi; a.ks = c1.ks; a.dil = c1.dil; a.tiles_per_batch = 0;
    a.bs = win ? win->in_bs : L * c1.cin;
    a.row_begin = win ? win->row_begin : 0;
    a.t_origin = win ? win->t_origin : 0;
    // streaming hops compute a few new rows behind a 64-row history: the 128 / 256-row tiles of the offline sweep would spend
    // most of their MFMAs on rows nobody reads, so short windows take the smallest tile (4 waves x 16 rows)
    const long long new_rows = L - a.row_begin;
    if (win && c1.cin == 64 && new_rows <= 32 - (c1.ks - 1)) return launch_amp_t<64, 2, 2, true, 4>(a, B, s);      // 32 rows, waves split the columns
    if (win && c1.cin == 64 && new_rows <= 2 * (64 - (c1.ks - 1))) return launch_amp_t<64, 1, 2, true>(a, B, s);
    static const int s32 = getenv("BVC_AMP32S") ? atoi(getenv("BVC_AMP32S")) : 3;       // (3 tiles of 64 rows against one of 256 for a two-frame hop: 1.53 -> 1.47 ms per tick at 256 streams)
    if (win && c1.cin == 32 && new_rows <= s32 * (64 - (c1.ks - 1))) return launch_amp_t<32, 1, 3, true>(a, B, s);
    if (c1.cin == 8 && c1.wp2 && c2.wp2 && (kernels & AMPK_C8)) {           // full-tile form of the C = 8 stage
        AmpArgs a8 = a;
        a8.w1 = c1.wp2; a8.w2 = c2.wp2;
        // tile shapes from a measured sweep (16-pair tiles per wave x register budget): <2, 2> 2.07 ms per step for the stage,
        // <2, 4> 2.17 (spills at ks = 11), <4, 4> 2.19, <1, 4> 2.4; the generic padded kernel 2.68.  Short streaming windows
        // take the smallest tile (4 waves x 16 pairs = 128 rows).
        const int rc8 = (win && new_rows <= 128) ? launch_amp8<1, 4>(a8, B, s) : launch_amp8<2, 2>(a8, B, s);
        if (rc8 != -1) return rc8;
    }
    switch (c1.cin) {
        // tile shapes from a measured sweep (tools/voc_stage_times.py): MT = 16-row tiles per wave, OCC = workgroups per CU the
        // register budget is set for, ALIAS = the S2 tile re-uses the LDS of the S1 tile (one more barrier, half the LDS)
        case 64: {
            static const bool rows = getenv("BVC_AMP64") && atoi(getenv("BVC_AMP64")) == 0;       // A/B: the waves along the rows (2.63 ms per step for the stage)
            if (rows) return launch_amp_t<64, 2, 2, true>(a, B, s);
            return launch_amp_t<64, 8, 2, true, 4>(a, B, s);       // waves along the columns: 2.33 (two column groups x two row groups: 2.60)
        }
        case 32: return launch_amp_t<32, 4, 3, true>(a, B, s);     // (waves along the columns, CS = 2 with 4 or 8 row tiles: 4.64 / 4.41 against 4.48)
        case 16: {
            if ((kernels & AMPK_C16) && !win) {                  // offline sweep: the persistent C = 16 kernel
                // four row tiles per wave (256 rows per workgroup): 2.82 (generic kernel) -> 2.62 ms per step for the stage; two tiles 2.82,
                // six (KS <= 7) 2.60
                const int rc16 = launch_amp16<4>(a, B, s);
                if (rc16 != -1) return rc16;
            }
            return launch_amp_t<16, 2, 4, false>(a, B, s);     // (MT 4 / ALIAS: no gain)
        }
        case 8:  return launch_amp_t<8, 4, 4, true>(a, B, s);       // 3.14 -> 2.74
        default: set_error("amp_pair: unsupported channel count %d", c1.cin); return BVC_EINVAL;
    }
}

template <int CIN, int NTW, int MT, bool NSPLIT = false>
static int launch_one(const ConvArgs &a, int B, hipStream_t s) {
    constexpr int TT = (NSPLIT ? 1 : 4) * MT * 16;
    ConvArgs k = a;
    k.tiles_per_batch = (int)((a.Lout - a.row_begin + TT - 1) / TT);
    if (k.tiles_per_batch <= 0) return BVC_OK;
    const size_t lds = (size_t)(TT + (a.ks - 1) * a.dil) * (CIN + 2) * sizeof(float);
    if (lds > 160 * 1024) { set_error("conv tile needs %zu B of LDS", lds); return BVC_EINVAL; }
    constexpr int NPW = NTW * (NSPLIT ? 4 : 1);            // column tiles per workgroup
    dim3 grid((unsigned)(k.tiles_per_batch * (long long)B), (unsigned)((a.ntiles + NPW - 1) / NPW));
    auto kern = conv_mfma_kernel<CIN, NTW, MT, NSPLIT>;
    ProbeScope probe(PK_CONV, s);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, k);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// Allow > 64 KiB of dynamic LDS for every instantiation (called once from bvc_model_create, so the
// compute entry points stay free of non-stream API calls).
template <int CIN, int NTW, int MT, bool NSPLIT = false>
static int allow_big_lds() {
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mfma_kernel<CIN, NTW, MT, NSPLIT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return BVC_OK;
}

// test helper: y[i] = SnakeBeta(x[i]) with one (exp(alpha), 1/(exp(beta)+1e-9)) pair, through the same device functions the
// generator uses: the packed form on elements 4k, 4k+1, the scalar form on 4k+2, 4k+3
__global__ void snakebeta_test_kernel(const float *__restrict__ x, long long n, float a, float ib, float *__restrict__ y) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (long long)gridDim.x * blockDim.x * 2) {
        if (i + 1 < n && !(i & 2)) {
            const f32x2 v = snakebeta2((f32x2){x[i], x[i + 1]}, splat2(a), splat2(ib));
            y[i] = v[0];
            y[i + 1] = v[1];
        } else {
            y[i] = snakebeta(x[i], a, ib);
            if (i + 1 < n) y[i + 1] = snakebeta(x[i + 1], a, ib);
        }
    }
}

int launch_snakebeta_test(const float *x, long long n, float a, float ib, float *y, hipStream_t s) {
    if (n <= 0) return BVC_OK;
    hipLaunchKernelGGL(snakebeta_test_kernel, dim3(256), dim3(256), 0, s, x, n, a, ib, y);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

int conv_kernels_init() {
    int rc;
    if (!amp8_slots_all<1, 4>() || !amp8_slots_all<2, 2>()) return BVC_EHIP;
    if (!amp16_slots_all<4>()) return BVC_EHIP;
    if ((rc = allow_big_lds<128, 4, 2>())) return rc;
    if ((rc = allow_big_lds<80, 4, 2>())) return rc;
    if ((rc = allow_big_lds<64, 4, 2>())) return rc;
    if ((rc = allow_big_lds<32, 2, 4>())) return rc;
    if ((rc = allow_big_lds<16, 1, 4>())) return rc;
    if ((rc = allow_big_lds<8, 1, 4>())) return rc;
    return BVC_OK;
}

int launch_conv_mfma(const ConvLayer &c, const float *in, long long Lin, float *out, long long Lout, int B,
                     int epi, const float *res, const float *acc, float divisor, hipStream_t s, const ConvWindow *win) {
    if (B <= 0 || Lout <= 0) return BVC_OK;
    if (c.cout % 4) { set_error("conv_mfma: %d output columns (the epilogue stores 16-byte granules)", c.cout); return BVC_EINVAL; }
    ConvArgs a;
    a.in = in; a.Lin = Lin; a.out = out; a.Lout = Lout; a.res = res; a.acc = acc;
    a.in_bs = win ? win->in_bs : Lin * c.cin;
    a.out_bs = win ? win->out_bs : Lout * c.cout;
    a.row_begin = win ? win->row_begin : 0;
    a.wp = c.wp; a.bias = c.bias; a.act_a = c.act_a; a.act_ib = c.act_ib;
    a.divisor = divisor; a.epi = epi; a.ks = c.ks; a.dil = c.dil; a.cout = c.cout; a.ntiles = c.ntiles;
    a.tiles_per_batch = 0;
    // streaming hops: one or two new frames = at most 16 rows in front of the first two upsamplers; the row-split tiles (64 rows
    // and more per workgroup) would compute mostly rows nobody reads, so the waves split the columns instead
    if (win && Lout - a.row_begin <= 16) {
        switch (c.cin) {
            case 128: return launch_one<128, 4, 1, true>(a, B, s);
            case 80:  return launch_one<80, 2, 1, true>(a, B, s);
            case 64:  return launch_one<64, 4, 1, true>(a, B, s);
            default: break;
        }
    }
    switch (c.cin) {
        case 128: return launch_one<128, 4, 2>(a, B, s);     // ConvT 128->8x64
        case 80:  return launch_one<80, 4, 2>(a, B, s);      // conv_pre 80->128
        case 64:  return launch_one<64, 4, 2>(a, B, s);      // AMP C=64, ConvT 64->8x32
        case 32:  return launch_one<32, 2, 4>(a, B, s);      // AMP C=32, ConvT 32->2x16
        case 16:  return launch_one<16, 1, 4>(a, B, s);      // AMP C=16, ConvT 16->2x8
        case 8:   return launch_one<8, 1, 4>(a, B, s);       // AMP C=8
        default:
            set_error("conv_mfma: unsupported input channel count %d", c.cin);
            return BVC_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------
// activation_post -> pad[6,0] -> conv_post (C -> 1) -> tanh -> [:length] -> / SCALING
// (models.py:228-238, bvrnn_codec_model.py:71).  VALU kernel: C*ks = 56 MACs per sample.
template <int C>
__global__ __launch_bounds__(256) void conv_post_kernel(const float *__restrict__ in, long long Lin, int ks,
                                                        const float *__restrict__ w, const float *__restrict__ bias,
                                                        const float *__restrict__ act_a,
                                                        const float *__restrict__ act_ib, float div,
                                                        float *__restrict__ wav, long long n_out,
                                                        int tiles_per_batch, long long in_bs, long long row_begin) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [(256 + ks-1)][C]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles_per_batch;
    const long long t0 = row_begin + (long long)(blockIdx.x % tiles_per_batch) * 256;     // input row of output 0 of the tile
    const int halo = ks - 1;
    const float *inb = in + (long long)b * in_bs;
    for (int idx = tid; idx < (256 + halo) * C; idx += 256) {
        const int row = idx / C, c = idx - row * C;
        const long long tg = t0 - halo + row;
        float v = 0.0f;
        if (tg >= 0 && tg < Lin) v = snakebeta(inb[tg * C + c], act_a[c], act_ib[c]);
        tile[idx] = v;
    }
    __syncthreads();
    const long long t = t0 + tid - row_begin;                        // output sample index
    if (t >= n_out) return;
    float acc = 0.0f;
    for (int j = 0; j < ks; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) acc = fmaf(w[c * ks + j], tile[(tid + j) * C + c], acc);
    wav[(long long)b * n_out + t] = tanhf(acc + bias[0]) / div;
}

int launch_conv_post(const float *in, long long Lin, int C, int ks, const float *w, const float *bias,
                     const float *act_a, const float *act_ib, float div, float *wav, long long n_out, int B,
                     hipStream_t s, const ConvWindow *win) {
    if (B <= 0 || n_out <= 0) return BVC_OK;
    if (C != 8) { set_error("conv_post: unsupported channel count %d", C); return BVC_EINVAL; }
    const int tiles = (int)((n_out + 255) / 256);
    const size_t lds = (size_t)(256 + ks - 1) * C * sizeof(float);
    ProbeScope probe(PK_POST, s);
    hipLaunchKernelGGL(conv_post_kernel<8>, dim3((unsigned)(tiles * (long long)B)), dim3(256), lds, s, in, Lin, ks,
                       w, bias, act_a, act_ib, div, wav, n_out, tiles, win ? win->in_bs : Lin * C, win ? win->row_begin : 0);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

#ifdef BVC_PHASE_PROBE
int phase_probe_read(unsigned long long *out, int reset) {
    BVC_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; BVC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z))); }
    return BVC_OK;
}
#endif
}  // namespace bvc
