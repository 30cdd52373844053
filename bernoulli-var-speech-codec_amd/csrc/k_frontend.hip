// STFT + log-mel front-end for gfx950: one 64-lane wavefront per 1024-sample frame.
//
// Replaces mel_spectrogram() third_party/BigVGAN/meldataset.py:60-95 (reflect pad :72-81,
// torch.stft :84-85, magnitude :86-87, mel matmul :89, log-clamp :38-39,:90) and the
// `x * SCALING` / `.permute(0,2,1)` around it (bvrnn_codec_model.py:49,56).
//
// A frame is windowed while it is loaded (coalesced 512-B rows, reflect indexing and the -10 dB
// scale folded into the load), packed as 512 complex points z[m] = x[2m] + i x[2m+1], and
// transformed by a radix-8 x 8 x 8 Stockham FFT: every lane owns 8 points, the two exchanges go
// through padded (bank-conflict-free) LDS tiles, twiddles come from tables built in double
// precision on the host.  The real-input split, sqrt(re^2+im^2+1e-9), the 80 triangular mel
// filters (stored sparse: 727 non-zero weights) and log(max(.,1e-5)) finish the frame in the same
// kernel; the result is written time-major (B,T,80) - the layout the BVRNN kernels consume.
// HBM traffic is the compulsory 1 KiB in (served 4x from L2 because frames overlap) + 320 B out.
#include "bvc_internal.h"

namespace bvc {

struct cplx { float re, im; };

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cplx mul_negi(cplx a) { return {a.im, -a.re}; }   // -i * a
__device__ __forceinline__ cplx mul_posi(cplx a) { return {-a.im, a.re}; }   // +i * a

// forward 8-point DFT, in place (decimation in time)
__device__ __forceinline__ void dft8(cplx (&x)[8]) {
    const float h = 0.70710678118654752440f;
    const cplx a0 = cadd(x[0], x[4]), a1 = csub(x[0], x[4]);
    const cplx a2 = cadd(x[2], x[6]), a3 = csub(x[2], x[6]);
    const cplx a4 = cadd(x[1], x[5]), a5 = csub(x[1], x[5]);
    const cplx a6 = cadd(x[3], x[7]), a7 = csub(x[3], x[7]);
    const cplx b0 = cadd(a0, a2), b2 = csub(a0, a2);
    const cplx b1 = cadd(a1, mul_negi(a3)), b3 = cadd(a1, mul_posi(a3));
    const cplx b4 = cadd(a4, a6), b6 = csub(a4, a6);
    const cplx b5 = cadd(a5, mul_negi(a7)), b7 = cadd(a5, mul_posi(a7));
    const cplx w1b5 = {h * (b5.re + b5.im), h * (b5.im - b5.re)};      // (h,-h)*b5
    const cplx w3b7 = {h * (b7.im - b7.re), -h * (b7.re + b7.im)};     // (-h,-h)*b7
    const cplx nib6 = mul_negi(b6);
    x[0] = cadd(b0, b4);   x[4] = csub(b0, b4);
    x[1] = cadd(b1, w1b5); x[5] = csub(b1, w1b5);
    x[2] = cadd(b2, nib6); x[6] = csub(b2, nib6);
    x[3] = cadd(b3, w3b7); x[7] = csub(b3, w3b7);
}

constexpr int SA = 72;                 // k0 stride of exchange buffer A (complex elements)
constexpr int SB = 65;                 // n0 stride of exchange buffer B
constexpr int WAVE_LDS = 8 * SA + 8 * SB + 512;      // complex elements per wave
constexpr int MAG_LDS = 520;                         // floats per wave

__global__ __launch_bounds__(256) void stft_logmel_kernel(FrontendTables t, const float *__restrict__ wav,
                                                          long long L, long long T, long long nframes,
                                                          int pad_left, float scale, float *__restrict__ mel) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cplx *bufA = reinterpret_cast<cplx *>(smem) + (long long)wave * WAVE_LDS;
    cplx *bufB = bufA + 8 * SA;
    cplx *zbuf = bufB + 8 * SB;
    float *mag = smem + 4 * WAVE_LDS * 2 + wave * MAG_LDS;

    // per-lane twiddles, loaded once
    cplx tw1[8], tw2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 a = t.tw1[k * 64 + lane];
        tw1[k] = {a.x, a.y};
        const float2 b = t.tw2[k * 8 + (lane & 7)];
        tw2[k] = {b.x, b.y};
    }
    float win0[8], win1[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) {
        win0[n2] = t.window[2 * (64 * n2 + lane)];
        win1[n2] = t.window[2 * (64 * n2 + lane) + 1];
    }

    const long long nblocks = (nframes + 3) / 4;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {     // uniform trip count
        long long f = blk * 4 + wave;
        const bool live = f < nframes;
        if (!live) f = nframes - 1;
        const long long b = f / T, tt = f - b * T;
        const float *src = wav + b * L;

        // ---- load + window: lane holds z[64*n2 + lane], n2 = 0..7
        cplx v[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const int m = 64 * n2 + lane;
            long long i0 = tt * 256 + 2 * m - pad_left, i1 = i0 + 1;
            i0 = i0 < 0 ? -i0 : (i0 >= L ? 2 * (L - 1) - i0 : i0);          // reflect, no edge repeat
            i1 = i1 < 0 ? -i1 : (i1 >= L ? 2 * (L - 1) - i1 : i1);
            v[n2].re = __fmul_rn(__fmul_rn(src[i0], scale), win0[n2]);
            v[n2].im = __fmul_rn(__fmul_rn(src[i1], scale), win1[n2]);
        }
        // ---- pass 1: DFT over n2, twiddle W_512^(lane*k0), scatter to [k0][lane]
        dft8(v);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) bufA[k0 * SA + lane] = (k0 == 0) ? v[0] : cmul(v[k0], tw1[k0]);
        __syncthreads();
        // ---- pass 2: lane = (k0, n0); DFT over n1, twiddle W_64^(n0*k1), scatter to [n0][k0][k1]
        {
            const int k0 = lane >> 3, n0 = lane & 7;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[n1] = bufA[k0 * SA + n1 * 8 + n0];
            dft8(v);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) bufB[n0 * SB + k0 * 8 + k1] = (k1 == 0) ? v[0] : cmul(v[k1], tw2[k1]);
        }
        __syncthreads();
        // ---- pass 3: lane = (k0, k1); DFT over n0 -> Z[k0 + 8 k1 + 64 k2]
        {
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) v[n0] = bufB[n0 * SB + lane];
            dft8(v);
            const int kbase = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) zbuf[kbase + 64 * k2] = v[k2];
        }
        __syncthreads();
        // ---- real-input split + magnitude for bins 0..kmax-1 (bins above carry zero mel weight)
        for (int k = lane; k < t.kmax; k += 64) {
            const cplx zk = zbuf[k & 511];
            const cplx zn = zbuf[(512 - k) & 511];
            const cplx xe = {0.5f * (zk.re + zn.re), 0.5f * (zk.im - zn.im)};
            const cplx xo = {0.5f * (zk.im + zn.im), -0.5f * (zk.re - zn.re)};
            const float2 w = t.tws[k];
            const float re = xe.re + (w.x * xo.re - w.y * xo.im);
            const float im = xe.im + (w.x * xo.im + w.y * xo.re);
            mag[k] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)), 1e-9f));
        }
        __syncthreads();
        // ---- triangular mel filters (sparse) + log compression
        for (int j = lane; j < t.num_mels; j += 64) {
            const int st = t.mel_start[j], ln = t.mel_len[j];
            const float *w = t.mel_w + t.mel_off[j];
            float acc = 0.0f;
            for (int i = 0; i < ln; ++i) acc = fmaf(w[i], mag[st + i], acc);
            // torch.clamp(min=1e-5) keeps a NaN (fmaxf would not).  The logarithm goes through double precision and is rounded once: ocml's
            // logf is good to about two units in the last place - it gave -11.512927 for the floor itself, where the reference's
            // torch.log gives the correctly rounded -11.512925 (meldataset.py:38-39) - and 80 logarithms per frame cost nothing
            if (live) mel[f * t.num_mels + j] = (float)log((double)(acc < 1e-5f ? 1e-5f : acc));
        }
        // next iteration's first LDS write (bufA) is ordered after this iteration's last read of
        // bufA by the two barriers above; mag/zbuf are rewritten only after further barriers.
    }
}

// ---- wire format of the codes (the reference keeps codes as float32 {0,1,0.5})
// frame = ceil(nbits/8) bytes, bit i of the frame at byte i/8, position i%8 (LSB first); only the
// nbits active bits of a frame are transmitted, the masked ones (0.5) are re-created on unpack.
__global__ void pack_codes_kernel(const float *__restrict__ codes, long long frames, int z, int nbits, int nbytes,
                                  unsigned char *__restrict__ out) {
    const long long total = frames * nbytes;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / nbytes;
        const int j = (int)(i - f * nbytes);
        unsigned v = 0;
        for (int b = 0; b < 8; ++b) {
            const int bit = j * 8 + b;
            if (bit < nbits && codes[f * z + bit] > 0.75f) v |= 1u << b;
        }
        out[i] = (unsigned char)v;
    }
}

__global__ void unpack_codes_kernel(const unsigned char *__restrict__ in, long long frames, int z, int nbits, int nbytes,
                                    float *__restrict__ codes) {
    const long long total = frames * z;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / z;
        const int bit = (int)(i - f * z);
        codes[i] = bit < nbits ? (float)((in[f * nbytes + (bit >> 3)] >> (bit & 7)) & 1u) : 0.5f;
    }
}

int launch_pack_codes(const float *codes, long long frames, int z, int nbits, unsigned char *out, hipStream_t s) {
    const int nbytes = (nbits + 7) / 8;
    const long long total = frames * nbytes;
    if (total <= 0) return BVC_OK;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_codes_kernel, dim3(grid), dim3(256), 0, s, codes, frames, z, nbits, nbytes, out);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

int launch_unpack_codes(const unsigned char *in, long long frames, int z, int nbits, float *codes, hipStream_t s) {
    const int nbytes = (nbits + 7) / 8;
    const long long total = frames * z;
    if (total <= 0) return BVC_OK;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(unpack_codes_kernel, dim3(grid), dim3(256), 0, s, in, frames, z, nbits, nbytes, codes);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// ---- pre-processing of example.py:15-17: rational-rate polyphase resampling
// (scipy.signal.resample_poly = upfirdn with a Kaiser-windowed sinc, zero padding at the edges) and
// per-utterance peak normalisation.  One thread per output sample walks the ~len(h)/up taps of its
// phase; accumulation in double, like the float64 reference call.
__global__ void resample_poly_kernel(const float *__restrict__ x, long long Lin, const double *__restrict__ h, int ntaps,
                                     int up, int down, long long n_pre_remove, float *__restrict__ y, long long n_out) {
    const long long b = blockIdx.y;
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < n_out; m += (long long)gridDim.x * blockDim.x) {
        const long long pos = (m + n_pre_remove) * down;          // index into the zero-stuffed, up-sampled signal
        long long i = pos / up;                                   // newest input sample under the filter
        int k = (int)(pos - i * up);                              // its tap (polyphase component)
        if (i >= Lin) { const long long skip = i - (Lin - 1); i -= skip; k += (int)(skip * up); }
        double acc = 0.0;
        for (; k < ntaps && i >= 0; k += up, --i) acc += h[k] * (double)x[b * Lin + i];
        y[b * n_out + m] = (float)acc;
    }
}

int launch_resample_poly(const float *x, int B, long long Lin, const double *h, int ntaps, int up, int down,
                         long long n_pre_remove, float *y, long long n_out, hipStream_t s) {
    if (B <= 0 || n_out <= 0) return BVC_OK;
    const int gx = (int)((n_out + 255) / 256 > 2048 ? 2048 : (n_out + 255) / 256);
    hipLaunchKernelGGL(resample_poly_kernel, dim3(gx, B), dim3(256), 0, s, x, Lin, h, ntaps, up, down, n_pre_remove, y, n_out);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

// x[b, :] /= max |x[b, :]|   (speech / np.max(np.abs(speech)), example.py:17); one workgroup per utterance
__global__ __launch_bounds__(1024) void peak_normalize_kernel(float *__restrict__ x, long long L) {
    __shared__ float red[1024];
    float *row = x + (long long)blockIdx.x * L;
    float mx = 0.0f;
    for (long long i = threadIdx.x; i < L; i += blockDim.x) mx = fmaxf(mx, fabsf(row[i]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    const float peak = red[0];
    if (peak > 0.0f)
        for (long long i = threadIdx.x; i < L; i += blockDim.x) row[i] = row[i] / peak;
}

int launch_peak_normalize(float *x, int B, long long L, hipStream_t s) {
    if (B <= 0 || L <= 0) return BVC_OK;
    hipLaunchKernelGGL(peak_normalize_kernel, dim3(B), dim3(1024), 0, s, x, L);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

int launch_stft_logmel(const FrontendTables &t, const float *wav, int B, long long L, long long T,
                       int pad_left, float scale, float *mel, hipStream_t s) {
    const long long nframes = (long long)B * T;
    if (nframes <= 0) return BVC_OK;
    const long long nblocks = (nframes + 3) / 4;
    const int grid = (int)(nblocks > 2048 ? 2048 : nblocks);
    const size_t lds = (size_t)(4 * WAVE_LDS * 2 + 4 * MAG_LDS) * sizeof(float);
    ProbeScope probe(PK_STFT, s);
    hipLaunchKernelGGL(stft_logmel_kernel, dim3(grid), dim3(256), lds, s, t, wav, L, T, nframes, pad_left,
                       scale, mel);
    BVC_HIP_TRY(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
