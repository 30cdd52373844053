// Internal declarations shared by the gfx950 kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/bvcodec.h"

namespace bvc {

void set_error(const char *fmt, ...);

#define BVC_HIP_TRY(expr)                                                              \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            bvc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BVC_EHIP;                                                           \
        }                                                                              \
    } while (0)

// ------------------------------------------------------------------ in-situ kernel timing (bench only)
// bvc_probe_begin(kind, every, max) makes every `every`-th launch of kernel family `kind` be
// bracketed by a hipEvent pair on its own stream; bvc_probe_end() reports the mean elapsed time.
enum ProbeKind { PK_NONE = 0, PK_LINEAR = 1, PK_GRU = 2, PK_CONV = 3, PK_BATCHED = 4, PK_STFT = 5, PK_POST = 6 };
struct ProbeScope {
    hipStream_t s; int slot;
    ProbeScope(int kind, hipStream_t stream);
    ~ProbeScope();
};

// ------------------------------------------------------------------ skinny / recurrent GEMM (k_gemm.hip)
// Per-call dynamic state, resident in device memory.  The recurrent-step kernels are captured ONCE
// into a hipGraph and replayed for every frame, so nothing that changes per call or per frame may be a
// kernel argument: caller tensors are reached through this descriptor and the frame index `t` is a
// device-side counter advanced by step_advance_kernel at the end of every step.
enum DescSlot { DS_PX = 0, DS_CODES = 1, DS_BITS = 2, DS_PROB = 3, DS_ALLH = 4, DS_MEL = 5, DS_PZ = 6, DS_NOISE = 7, DS_NSLOT = 8,
                DS_PRIOR = DS_ALLH /* BVRNN.forward has no all_h output: the slot carries the prior probabilities */,
                DS_PARTD = DS_PX, DS_PARTG = DS_CODES /* decode has neither: pre-computed phi_z halves of dec.0 / of the GRU input */,
                DS_KEEP = DS_PROB /* decode with the folded hop: ELU(dec.4) of all frames, (B,T,H) */,
                DS_KEEP_ENC = DS_PZ /* ... and encode's, when the fused forward wants the decoder's output (encode has no phi_z tensor) */ };
struct CallDesc {
    float *p[DS_NSLOT];              // base pointers of the (B, T, dim) tensors of this call (may be null)
    long long T;                     // frames per utterance
    int t;                           // current frame
    int nodes_per_step;              // kernels per step (probe slot = t * nodes_per_step + node)
};

// A pointer whose value depends on the frame counter.
//   kind 0 (static) : base, row stride ld
//   kind 1 (frame)  : desc->p[sel] + (t + toff) * dim, row stride T * dim; invalid outside [0, T) or if null
//   kind 2 (parity) : base + (((t + toff) & 1) ? poff : 0), row stride ld      (GRU state ping-pong)
//   packed != 0: the [M][ld] matrix is stored in MFMA A-operand fragment order
//       [m/16][k/16][lane = ((k%16)/4)*16 + m%16][k%4]   (one 16x16 block = 1 KiB contiguous),
//       so a wave's operand load is one fully coalesced 1 KiB read; frame tensors then hold one such
//       packed [M][dim] matrix per frame.
struct DynPtr {
    float *base;
    int ld;                       // row stride in floats (static / parity kinds)
    int poff;                     // parity kind: offset added when ((t + toff) & 1); static kind: floats between the frames of a multi-frame launch (GemmParams::frames)
    int meta;                     // kind | packed << 4 | sel << 8 | (toff + 8) << 16
    int dim;                      // frame kind: floats per frame row
};
inline int dp_meta(int kind, int packed, int sel, int toff) { return kind | (packed << 4) | (sel << 8) | ((toff + 8) << 16); }
inline DynPtr dp_static(const float *p, long long ld, int packed = 0) { return DynPtr{const_cast<float *>(p), (int)ld, 0, dp_meta(0, packed, 0, 0), 0}; }
// static pointer of a launch that covers several frames at once (grid.y = frame): frame f reads / writes base + f * fstride
inline DynPtr dp_static_frames(const float *p, long long ld, long long fstride, int packed = 0) { return DynPtr{const_cast<float *>(p), (int)ld, (int)fstride, dp_meta(0, packed, 0, 0), 0}; }
inline DynPtr dp_frame(int sel, int dim, int toff = 0, int packed = 0) { return DynPtr{nullptr, 0, 0, dp_meta(1, packed, sel, toff), dim}; }
inline DynPtr dp_parity(float *p, long long ld, long long poff, int flip, int packed = 0) { return DynPtr{p, (int)ld, (int)poff, dp_meta(2, packed, 0, flip), 0}; }
inline DynPtr dp_null() { return DynPtr{nullptr, 0, 0, 0, 0}; }
inline int dp_kind(const DynPtr &d) { return d.meta & 15; }
inline int dp_packed(const DynPtr &d) { return (d.meta >> 4) & 1; }

// One K-segment of a (possibly concatenated) input:  acc[grp] += x[M,K] @ w[rows,K]^T
struct GemmSeg {
    const float *w;      // weights in MFMA B-operand fragment order [n/16][k/16][lane][4], already offset
                         // to this segment's first k-block
    int          wnb;    // k-blocks (of 16) per weight row, i.e. floats between n-tiles / 256
    int          K;      // multiple of 16
    DynPtr       x;      // [M][ld]
    int          grp;    // accumulator group (0: input part, 1: hidden part of the GRU)
    int          pad_;
};

enum GemmEpi {
    EPI_LINEAR = 0,      // y = acc + bias
    EPI_ELU = 1,         // y = elu(acc + bias)
    EPI_CODE = 2,        // p = sigmoid(acc+bias); z = rint(p); masked by bits  (bvrnn.py:189-194)
    EPI_MEL = 3,         // y = acc + bias (mel frame) ; y2 = (y - mean) / std  (bvrnn.py:202-204)
    EPI_GRU = 4,         // PyTorch GRU cell, 3 gates x 2 groups (whole cell in one launch)
    EPI_GRU_PART = 5,    // GRU cell whose hidden part and phi_z part were pre-computed on the side branch:
                         //   gi = acc + y2part (W_ih[:, H:] phi_z + b_ih), gh = y3part (W_hh h + b_hh)
    EPI_SIGMOID = 6      // y = sigmoid(acc + bias)   (prior net, bvrnn.py:68-73)
};
// EPI_CODE variants (GemmParams::sample): what is rounded and which value z takes
enum CodeSample {
    CS_ENCODE = 0,       // z = round(p)                              BVRNN.encode (bvrnn.py:191)
    CS_GREEDY = 1,       // z = (round(p) - p) + p                    BVRNN.forward greedy (bvrnn.py:124), straight-through value
    CS_SAMPLE = 2        // z = (round((u - 0.5) + p) - p) + p        Bernoulli sampler (bvrnn.py:126), u from y2
};

// The first 16 dwords (M .. seg[0].x) hold everything a layer needs to map its tile and issue the operand
// loads of its first segment, so they can be pre-loaded into SGPRs at dispatch (build with
// BVC_KERNARG_PRELOAD=1 -> -mllvm -amdgpu-kernarg-preload-count=16; worth 0.3 us per layer in
// tools/skinny_bench, neutral in the full schedule, hence off by default).
struct GemmParams {
    int     M, N;              // N = outputs per gate (multiple of 16)
    int     nb_total;          // sum over segments of K/16
    int     gate_rows;         // row distance between gates inside w (GRU: h_dim); bias index stride
    GemmSeg seg[3];
    int     nseg;
    int     var_bit;
    int     sample;            // EPI_CODE: CodeSample
    int     gate_il;           // weights are gate-interleaved [n/16][k/16][gate][lane][4] (GRU launches)
    const float *bias0;        // group 0 bias [gates*N] (may be null)
    const float *bias1;        // group 1 bias (GRU only)
    DynPtr  y, y2, y3;         // outputs (y2/y3 optional); EPI_CODE with CS_SAMPLE: y2 = uniform noise INPUT
    DynPtr  aux;               // CODE: bits per frame (one per row); GRU: previous h; LINEAR/ELU: optional addend (natural [M][N])
                               // GRU: y3 (if valid) = pre-computed part of the input gates gi, natural [M][3*gate_rows]
    const float *part_i; const float *part_h; long long ldpart;   // GRU_PART: side-branch partial sums [M][3H]
    const float *mean; const float *stdv; // MEL epilogue
    const CallDesc *desc;      // null for stand-alone launches (all pointers static)
    unsigned long long *probe; // in-kernel timing slots, set only in the probe variant of a step graph
    int     node;              // index of this kernel inside its step (probe slot)
    int     tstep;             // static frame offset inside an unrolled multi-step graph: t = desc->t + tstep
    int     frames;            // > 1: the same layer for several independent frames in ONE launch (grid.y; every pointer static, see dp_static_frames)
};

int launch_gemm_skinny(const GemmParams &p, int epi, hipStream_t s, int mtw = 1);
int skinny_kernels_init();
int launch_step_advance(CallDesc *d, int by, hipStream_t s);
int launch_set_desc(CallDesc *d, const CallDesc &v, hipStream_t s);

// batched GEMM over all frames: y = act(x @ w^T + bias), M large.  out_mode (GemmOut) says where the rows go:
// see out_index() in k_gemm.hip.  frames_T = frames per utterance, mt16 = utterances rounded up to 16.
enum GemmOut { GO_NATURAL = 0, GO_FRAME_MAJOR_ROWS = 1, GO_PACKED_FRAMES = 2, GO_PACKED_FROM_UTT = 3 };
int launch_gemm_batched(const float *x, long long ldx, const float *w, long long ldw, const float *bias,
                        int M, int N, int K, int act, float *y, long long ldy, hipStream_t s,
                        int out_mode = GO_NATURAL, long long frames_T = 0, int mt16 = 0);
// yn = (y - mean) / std over rows of length n (bvrnn.py:173)
int launch_normalize_rows(const float *y, const float *mean, const float *stdv, long long rows, int n,
                          float *out, hipStream_t s);
int launch_fill(float *p, float v, long long n, hipStream_t s);
// per-frame KL term of BVRNN.forward (bvrnn.py:148-157): kld[t] = mean_b sum_j mask * elem(prob, prior)
int launch_kld_frames(const float *prob, const float *prior, const float *bits, int B, long long T, int Z, float *kld,
                      hipStream_t s);
int launch_copy_rows(const float *src, long long lds, float *dst, long long ldd, int rows, int n, hipStream_t s);
// natural [rows][n] (row stride ld) <-> fragment-packed [rows/16][n/16][64][4]; dir 0: pack, 1: unpack
int launch_repack_rows(const float *src, float *dst, long long ld_natural, int rows, int n, int dir, hipStream_t s);

// ------------------------------------------------------------------ persistent recurrence (k_flow.hip)
// All frames of BVRNN.encode / BVRNN.decode in one launch; layers are chained by "poisoned buffer" dataflow
// (see k_flow.hip).  Activations live in the flow region of the workspace: FlowBuf id, frame parity ->
// flow + (id * 2 + parity) * slot_bytes, each a fragment-packed [MT16][dim] matrix.
constexpr unsigned FLOW_POISON = 0xFFFFDEADu;      // a NaN bit pattern no layer may publish as data
constexpr int FLOW_STAMPS = 6;                     // stamp kinds per layer and frame (k_flow.hip: flow_stamp)
enum FlowEpi { FE_ELU = 0, FE_CODE = 1, FE_MEL = 2, FE_GRU = 3, FE_ELU_KEEP = 4 };   // FE_ELU_KEEP: ELU, and the result also goes to FlowArgs::keep
enum FlowBuf { FB_H = 0, FB_E1, FB_E2, FB_ZC, FB_Q1, FB_Q2, FB_Q3, FB_D1, FB_D2, FB_D3, FB_DN, FB_G1, FB_G2, FB_G3, FB_COUNT };
struct FlowLin {         // one K-segment of a layer as the persistent kernel sees it (32-/64-bit fields: scalar loads)
    const float *w;      // packed weights [n/16][k/16][lane][4], offset to the segment's first k-block
    const float *bias;   // [N] or null
    int wnb;             // k-blocks per weight row
    int pad_;
};
struct FlowArgs {
    // layers of the step in dependency order (encode: bvrnn.py:187-206, decode: bvrnn.py:222-227).  enc0h / dec0h: the
    // halves of enc.0 / dec.0 that multiply h (the other halves are batched over all frames beforehand: part0);
    // dec0z: the phi_z half of dec.0, used by encode only (there the bias of dec.0 travels in dec0h).
    FlowLin enc0h, enc1, enc2, pz0, pz1, pz2, dec0h, dec0z, dec1, dec2, dec3, px0, px1, px2;
    const float *w_hh, *w_ihx, *w_ihz;      // GRU, gate-interleaved [n/16][k/16][gate][lane][4]: W_hh, W_ih[:, :H], W_ih[:, H:]
    const float *b_ih, *b_hh;
    int hb, zb, xb;                         // h_dim / 16, z_dim / 16, num_mels / 16
    float *flow; unsigned slot_bytes;
    int B, MT, NTG;                         // utterances, utterance groups of 16, feature tiles covered by the grid
    long long T;
    const float *part0;                     // (B,T,H) pre-computed half of the first layer (+ its bias)
    const float *part_gru;                  // decode: (B,T,3H) pre-computed phi_z half of the GRU input gates (+ b_ih)
    float *codes, *prob; const float *bits; float *all_h; float *mel;
    const float *mean, *stdv;
    int var_bit;
    unsigned *status; unsigned spin_limit;
    unsigned long long *probe;              // bench only: [FLOW_STAMPS][T][nodes] s_memrealtime stamps of one wave (0 layer entry, 1 exit, 2.. diagnostics)
    int probe_nodes, probe_first;           // layers per frame, id of the first one
    int probe_wg, probe_wave;               // which workgroup / wave stamps (BVC_PROBE_WG / BVC_PROBE_WAVE, default 0 / 0)
    int dbg_hot_w;                          // experiments only (BVC_FLOW_HOTW=1): every weight request hits the same blocks (wrong results)
    int MG;                                 // utterance groups (chains) per workgroup; 1 = one chain (B <= 16 * CUs / feature tiles)
    int dbg_withhold;                       // tests only: workgroup 0 returns at once
    // the folded hop (bvcodec_abi.hip: build_bvrnn): dec.6 has no activation, so phi_x.0(norm(dec.6(u))) is ONE affine map
    // of u = ELU(dec.4(.)) followed by the ELU: pxc = (phi_x.0.W diag(1/std) dec.6.W, phi_x.0.W ((dec.6.b - mean) / std) + phi_x.0.b).
    // The kernel then runs dec.4 -> pxc -> phi_x.2 ... (one wide layer instead of the two narrow hops dec.6, phi_x.0); decode stores u for
    // all frames in `keep` (B,T,H), and dec.6 itself - the decoder's output - is one batched GEMM over `keep` behind the launch.
    FlowLin pxc;                            // w == null: the layers as the reference lists them
    float *keep;
};
int flow_kernels_init();
int flow_perh(int h_dim);
int launch_flow(const FlowArgs &a, FlowArgs *d_args, int perh, bool encode, bool fill, hipStream_t s, bool args_resident = false);
// sentinel fill of the flow region + initial state into its first buffer + the device copy of the arguments, in one kernel
int launch_flow_prepare(const FlowArgs &a, FlowArgs *d_args, unsigned *flow, long long n_flow, long long n_h0, const float *d_h0, int B, int H,
                        hipStream_t s);
int launch_flow_census(unsigned *ctr, int grid, unsigned spin_limit, hipStream_t s);   // ctr[0] arrivals, ctr[1] workgroups that gave up
int launch_fill_u32(unsigned *p, unsigned v, long long n, hipStream_t s);

// ------------------------------------------------------------------ front-end (k_frontend.hip)
struct FrontendTables {          // device pointers
    const float *window;         // [1024]
    const float2 *tw1;           // [8][64]  W_512^(lane*k0)
    const float2 *tw2;           // [8][8]   W_64^(n0*k1)
    const float2 *tws;           // [513]    e^{-2 pi i k / 1024}
    const int   *mel_start;      // [num_mels]
    const int   *mel_len;        // [num_mels]
    const int   *mel_off;        // [num_mels] offset into mel_w
    const float *mel_w;          // packed non-zero weights
    int num_mels;
    int kmax;                    // highest bin with non-zero weight (+1)
};
int launch_stft_logmel(const FrontendTables &t, const float *wav, int B, long long L, long long T,
                       int pad_left, float scale, float *mel, hipStream_t s);

int launch_resample_poly(const float *x, int B, long long Lin, const double *h, int ntaps, int up, int down,
                         long long n_pre_remove, float *y, long long n_out, hipStream_t s);
int launch_peak_normalize(float *x, int B, long long L, hipStream_t s);
int launch_pack_codes(const float *codes, long long frames, int z, int nbits, unsigned char *out, hipStream_t s);
int launch_unpack_codes(const unsigned char *in, long long frames, int z, int nbits, float *codes, hipStream_t s);

// ------------------------------------------------------------------ vocoder (k_vocoder.hip)
struct ConvLayer {               // one causal conv as implicit GEMM on fp32 MFMA
    int cin;                     // input channels (multiple of 4)
    int cout;                    // real output columns
    int ntiles;                  // ceil(cout/16)
    int ks;                      // taps
    int dil;
    const float *wp;             // packed B fragments [ks][cin/4][ntiles][64]
    const float *wp4;            // cin == cout, a multiple of 16, >= 32: [ks][cin/16][ntiles][64][4] - a lane's fragments of four consecutive k-steps as ONE
                                 // 16-byte granule (amp_pair_kernel's streamed weights: a quarter of the load instructions), else nullptr
    const float *wp2;            // cin == cout == 8 only: two-output-rows-per-tile form [ks+1][2][64] (amp_pair8_kernel), else nullptr
    const float *bias;           // [cout] (for ConvT: bias replicated per phase)
    const float *act_a;          // exp(alpha) per input channel or nullptr (no input activation)
    const float *act_ib;         // 1/(exp(beta)+1e-9)
};
enum ConvEpi { CE_STORE = 0, CE_RES = 1, CE_RES_ACC = 2, CE_RES_ACC_DIV = 3 };
// Streaming window: the tensors are (B, rows, C) buffers whose first rows are history; only rows from
// row_begin on are computed.  t_origin = global time of buffer row 0 (for the zero-before-start rule).
#ifdef BVC_PHASE_PROBE
int phase_probe_read(unsigned long long *out, int reset);   // debugging builds only (tools/phase_probe.py)
#endif
struct ConvWindow { long long in_bs, out_bs, row_begin, t_origin; };
// in (B, Lin, cin) channels-last; out (B, Lout, cout).  Output row r reads input rows
// r - (ks-1)*dil ... r  (rows outside [0,Lin) are zero).  res/acc have the layout of out.
int conv_kernels_init();
// stage-specific AMP kernels a launch may take (per model): C = 8 pairs on the two-rows-per-tile kernel, C = 16 pairs on the persistent
// kernel with swapped operands; without the bit the generic amp_pair_kernel runs (same bits)
enum : unsigned { AMPK_C8 = 1u, AMPK_C16 = 2u, AMPK_ALL = 3u };
unsigned amp_kernels_default();
int launch_snakebeta_test(const float *x, long long n, float a, float ib, float *y, hipStream_t s);
int launch_conv_mfma(const ConvLayer &c, const float *in, long long Lin, float *out, long long Lout,
                     int B, int epi, const float *res, const float *acc, float divisor, hipStream_t s,
                     const ConvWindow *win = nullptr);
// one fused AMPBlock1 iteration: out = x + conv2(S2(conv1_dil(S1(x)))) (+acc, /divisor per epi); c2.dil == 1
int launch_amp_pair(const ConvLayer &c1, const ConvLayer &c2, const float *x, long long L, float *out, int B, int epi,
                    const float *acc, float divisor, hipStream_t s, const ConvWindow *win = nullptr, unsigned kernels = AMPK_ALL);
// SnakeBeta -> causal conv C->1 (k taps) -> tanh -> / div -> first n_out samples
int launch_conv_post(const float *in, long long Lin, int C, int ks, const float *w, const float *bias,
                     const float *act_a, const float *act_ib, float div, float *wav, long long n_out,
                     int B, hipStream_t s, const ConvWindow *win = nullptr);

}  // namespace bvc
