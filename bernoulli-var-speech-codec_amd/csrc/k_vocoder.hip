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
#include "bvc_internal.h"

namespace bvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *in;  long long Lin;
    float *out;       long long Lout;
    const float *res; const float *acc;
    const float *wp;  const float *bias;
    const float *act_a; const float *act_ib;
    float divisor;
    int epi, ks, dil, cout, ntiles, tiles_per_batch;
};

__device__ __forceinline__ float snakebeta(float x, float a, float ib) {
    const float s = sinf(__fmul_rn(x, a));
    return __fadd_rn(x, __fmul_rn(ib, __fmul_rn(s, s)));
}

// CIN: input channels; NTW: 16-column tiles per workgroup; MT: 16-row tiles per wave.
template <int CIN, int NTW, int MT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int S = CIN + 2;                 // LDS row stride (floats): (S/2) odd -> conflict-free
    constexpr int TT = 4 * MT * 16;            // output rows per workgroup
    constexpr int C4 = CIN / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_batch;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_batch) * TT;
    const int ntile0 = blockIdx.y * NTW;
    const int halo = (a.ks - 1) * a.dil;
    const int rows = TT + halo;

    // ---- stage the activated input span [t0-halo, t0+TT) in LDS
    const float *inb = a.in + (long long)b * a.Lin * CIN;
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
                for (int e = 0; e < 4; ++e) v[e] = snakebeta(v[e], aa[e], bb[e]);
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

    const int mbase = wave * MT * 16;
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
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bw[u][n], acc[i][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D[row = g*4+e][col = r]
    const long long ob = (long long)b * a.Lout;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int col = (ntile0 + n) * 16 + r;
        if (col >= a.cout) continue;
        const float bias = a.bias[col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long t = t0 + mbase + i * 16 + g * 4 + e;
                if (t >= a.Lout) continue;
                const long long o = (ob + t) * a.cout + col;
                float v = acc[i][n][e] + bias;
                if (a.epi >= CE_RES) v = v + a.res[o];               // x = xt + x      (models.py:119)
                if (a.epi >= CE_RES_ACC) v = a.acc[o] + v;           // xs += resblock  (models.py:224)
                if (a.epi == CE_RES_ACC_DIV) v = v / a.divisor;      // xs / num_kernels (models.py:225)
                a.out[o] = v;
            }
    }
}

template <int CIN, int NTW, int MT>
static int launch_one(const ConvArgs &a, int B, hipStream_t s) {
    constexpr int TT = 4 * MT * 16;
    ConvArgs k = a;
    k.tiles_per_batch = (int)((a.Lout + TT - 1) / TT);
    const size_t lds = (size_t)(TT + (a.ks - 1) * a.dil) * (CIN + 2) * sizeof(float);
    if (lds > 160 * 1024) { set_error("conv tile needs %zu B of LDS", lds); return BVC_EINVAL; }
    dim3 grid((unsigned)(k.tiles_per_batch * (long long)B), (unsigned)((a.ntiles + NTW - 1) / NTW));
    auto kern = conv_mfma_kernel<CIN, NTW, MT>;
    ProbeScope probe(PK_CONV, s);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, k);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// Allow > 64 KiB of dynamic LDS for every instantiation (called once from bvc_model_create, so the
// compute entry points stay free of non-stream API calls).
template <int CIN, int NTW, int MT>
static int allow_big_lds() {
    BVC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mfma_kernel<CIN, NTW, MT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return BVC_OK;
}

int conv_kernels_init() {
    int rc;
    if ((rc = allow_big_lds<128, 4, 2>())) return rc;
    if ((rc = allow_big_lds<80, 4, 2>())) return rc;
    if ((rc = allow_big_lds<64, 4, 2>())) return rc;
    if ((rc = allow_big_lds<32, 2, 4>())) return rc;
    if ((rc = allow_big_lds<16, 1, 4>())) return rc;
    if ((rc = allow_big_lds<8, 1, 4>())) return rc;
    return BVC_OK;
}

int launch_conv_mfma(const ConvLayer &c, const float *in, long long Lin, float *out, long long Lout, int B,
                     int epi, const float *res, const float *acc, float divisor, hipStream_t s) {
    if (B <= 0 || Lout <= 0) return BVC_OK;
    ConvArgs a;
    a.in = in; a.Lin = Lin; a.out = out; a.Lout = Lout; a.res = res; a.acc = acc;
    a.wp = c.wp; a.bias = c.bias; a.act_a = c.act_a; a.act_ib = c.act_ib;
    a.divisor = divisor; a.epi = epi; a.ks = c.ks; a.dil = c.dil; a.cout = c.cout; a.ntiles = c.ntiles;
    a.tiles_per_batch = 0;
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
                                                        int tiles_per_batch) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [(256 + ks-1)][C]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles_per_batch;
    const long long t0 = (long long)(blockIdx.x % tiles_per_batch) * 256;
    const int halo = ks - 1;
    const float *inb = in + (long long)b * Lin * C;
    for (int idx = tid; idx < (256 + halo) * C; idx += 256) {
        const int row = idx / C, c = idx - row * C;
        const long long tg = t0 - halo + row;
        float v = 0.0f;
        if (tg >= 0 && tg < Lin) v = snakebeta(inb[tg * C + c], act_a[c], act_ib[c]);
        tile[idx] = v;
    }
    __syncthreads();
    const long long t = t0 + tid;
    if (t >= n_out) return;
    float acc = 0.0f;
    for (int j = 0; j < ks; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) acc = fmaf(w[c * ks + j], tile[(tid + j) * C + c], acc);
    wav[(long long)b * n_out + t] = tanhf(acc + bias[0]) / div;
}

int launch_conv_post(const float *in, long long Lin, int C, int ks, const float *w, const float *bias,
                     const float *act_a, const float *act_ib, float div, float *wav, long long n_out, int B,
                     hipStream_t s) {
    if (B <= 0 || n_out <= 0) return BVC_OK;
    if (C != 8) { set_error("conv_post: unsupported channel count %d", C); return BVC_EINVAL; }
    const int tiles = (int)((n_out + 255) / 256);
    const size_t lds = (size_t)(256 + ks - 1) * C * sizeof(float);
    ProbeScope probe(PK_POST, s);
    hipLaunchKernelGGL(conv_post_kernel<8>, dim3((unsigned)(tiles * (long long)B)), dim3(256), lds, s, in, Lin, ks,
                       w, bias, act_a, act_ib, div, wav, n_out, tiles);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
