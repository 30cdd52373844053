/* bvcodec.h - C ABI of libbvcodec_hip.so: the MI355X (gfx950) implementation of the
 * BVRNNCodecModel encode/decode hot path.
 *
 * The reference (BenjSta/bernoulli-var-speech-codec) is pure Python/PyTorch and has NO FFI of its
 * own; its boundary for this path is the Python class BVRNNCodecModel (bvrnn_codec_model.py:19-76).
 * This header is the boundary the build adds underneath that class: each entry point names the
 * reference call it replaces (file:line relative to the reference checkout).  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C, no torch/HIP types: `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - every pointer named d_* is DEVICE memory owned by the caller, contiguous float32;
 *    every pointer named h_* is HOST memory;
 *  - the compute entry points (bvc_stft_logmel, bvc_bvrnn_*, bvc_bigvgan, bvc_encode, bvc_decode,
 *    bvc_vocoder_stream_push, bvc_pack/unpack_codes, bvc_resample_poly, bvc_peak_normalize) are
 *    asynchronous on `stream` and use only the caller-provided workspace.  They do not allocate or
 *    synchronise, with these exceptions: the first calls per process create a handful of HIP events (the
 *    persistent launches' ticket, the end-of-call mark of the "auto" schedule, one per bvc_flow_fence slot in
 *    use), the launch-per-layer schedule captures and instantiates a hipGraph on a stream of the library's own
 *    on the first call per (batch, workspace), and the call that follows a reported time-out re-runs the
 *    residency census (it synchronises a stream of the library's own, and the device if the census fails);
 *  - capturing into a graph of your own: allowed for every compute entry point.  While `stream` is
 *    being captured a call records no event and waits on none, and its recurrence takes the
 *    launch-per-layer kernels, launched directly into your capture (the persistent recurrence kernel is
 *    never captured: see "recurrence" below); the status word is still read when the call is ISSUED, not
 *    on replay;
 *  - bvc_model_status, bvc_probe_end, bvc_kprobe_* and the bvc_test_* helpers synchronise the device
 *    (bvc_model_poll_status does not);
 *  - return value: 0 = BVC_OK, negative = error code; bvc_last_error() gives the text
 *    (thread-local); no C++ exception crosses the boundary;
 *  - one in-flight call per (model, workspace): calls on different streams with different
 *    workspaces may overlap (persistent recurrence kernels of overlapping calls run one after the
 *    other, everything else concurrently; with the default "recurrence" = auto overlapping calls take
 *    the launch-per-layer schedule, whose kernels interleave).  Several host threads may issue calls on ONE model
 *    at once, each with its own workspace and stream (the model's graph cache, the persistent launches' ticket and
 *    the census are guarded inside the library); a model's weights are immutable after creation, and its options
 *    (bvc_model_set_option) must not be changed while another thread is issuing calls on it.
 */
#ifndef BVCODEC_H
#define BVCODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVC_ABI_VERSION 3   /* 2: + bvc_model_get_option, bvc_flow_fence, bvc_kprobe_read_span; recurrence option takes 2 (auto); status word reported by every compute entry
                             * 3: + bvc_model_poll_status, bvc_forward */

enum {
    BVC_OK = 0,
    BVC_EINVAL = -1,      /* bad argument / unsupported shape */
    BVC_ENOMEM = -2,      /* workspace too small or device allocation failed */
    BVC_EHIP = -3,        /* HIP runtime error (see bvc_last_error) */
    BVC_EMISSING = -4,    /* a required weight tensor is missing or has the wrong size */
    BVC_ENODEVICE = -5,   /* no gfx950 device visible */
    BVC_ETIMEOUT = -6     /* a persistent recurrence kernel gave up waiting for its peers (bvc_model_status) */
};

/* Static description of the codec; mirrors the TOML keys the reference facade reads
 * (configs/config_varBitRate.toml:21-29,35-37,39-56; bvrnn_codec_model.py:27-36,49-59). */
typedef struct bvc_config {
    int32_t num_mels;          /* 80 */
    int32_t h_dim;             /* 1024 */
    int32_t z_dim;             /* 64 */
    int32_t var_bit;           /* 1: variable bits/frame mask (bvrnn.py:180-182,193-194) */
    int32_t n_fft;             /* 1024 (== win) */
    int32_t hop;               /* 256 */
    int32_t pad_left;          /* mel_pad_left = 256 */
    int32_t sample_rate;       /* 22050 */
    float   fmin, fmax;        /* 0, 8000 */
    int32_t upsample_initial_channel;      /* 128 */
    int32_t n_up;                          /* 4 */
    int32_t up_rates[8];                   /* 8,8,2,2 */
    int32_t up_kernels[8];                 /* 16,16,4,4 (must be 2*rate) */
    int32_t n_resk;                        /* 3 */
    int32_t res_kernels[4];                /* 3,7,11 */
    int32_t res_dilations[4][3];           /* 1,3,5 each */
} bvc_config;

/* One named HOST tensor (float32, contiguous, PyTorch layout).  Names are the reference's
 * state_dict keys: BVRNN (bvrnn.py:30-83) "mean_mel","std_mel","phi_x.0.weight",...,
 * "rnn.bias_hh_l0"; generator (models.py:132-205) with weight-norm ALREADY FOLDED by the caller:
 * "conv_pre.weight","conv_pre.bias","ups.0.1.weight",...,"resblocks.0.convs1.0.weight",...,
 * "resblocks.0.activations.0.alpha",...,"activation_post.alpha","conv_post.weight",...;
 * plus "mel_basis" (num_mels x (n_fft/2+1), the librosa.filters.mel matrix of meldataset.py:68). */
typedef struct bvc_tensor {
    const char  *name;
    const float *h_data;
    int64_t      numel;
} bvc_tensor;

typedef struct bvc_model bvc_model;     /* opaque: device-resident, re-laid-out weights */

int          bvc_abi_version(void);
const char  *bvc_last_error(void);

/* Replaces BVRNNCodecModel.__init__'s module construction + load_state_dict
 * (bvrnn_codec_model.py:30-42): uploads and re-lays-out the weights on the current device. */
int  bvc_model_create(const bvc_config *cfg, const bvc_tensor *tensors, int32_t n_tensors,
                      bvc_model **out);
void bvc_model_destroy(bvc_model *m);

/* Run-time options of a model (not thread-safe; set them while no call is in flight).
 *   "recurrence": how the frame loop of BVRNN.encode / BVRNN.decode (bvrnn.py:186-206, 222-227) is scheduled.
 *                 0 = one persistent kernel launch per call (fastest for one batch at a time),
 *                 1 = one launch per layer, hipGraph-replayed (more throughput when batches are in flight on several streams),
 *                 2 = automatic (default; the start-up default follows BVC_RECURRENCE=persistent|layers|auto): persistent while
 *                     calls come one at a time; while a call starts before the previous one - issued on ANOTHER stream - has
 *                     finished, launch per layer, for all streams alike.
 *                 Whatever the option says, a call takes the launch-per-layer kernels when its stream is being captured
 *                 into a graph (a persistent launch is serialised against other persistent launches by a host-side ticket,
 *                 which a replay would skip), when the batch is beyond the persistent kernel's limits, or when the
 *                 residency census at bvc_model_create found that a full persistent grid is not co-resident on this device.
 *   "vocoder_full_tiles": 1 (default) = the eight-channel generator stage runs on the kernel that packs two output rows into
 *                 one MFMA tile, 0 = on the generic kernel (half of every tile is channel padding).  Same bits either way;
 *                 a validation switch (per model, like every option).
 *   "vocoder_c16_kernel": 1 (default) = the sixteen-channel generator stage runs offline on its persistent kernel (weights in
 *                 registers, next tile's rows under the current tile's convs, 16-byte epilogues), 0 = on the generic kernel.
 *                 Same bits either way; per model.
 *   "decode_fold": 1 (default) = the persistent DECODE kernel runs phi_x.0((dec.6(u) - mean) / std) - three maps with no
 *                 non-linearity between them (bvrnn.py:80, :226) - as ONE affine map of u (folded in float64 at model creation):
 *                 one wide layer instead of two narrow hops per frame; dec.6(u), the decoder's output, is then one batched GEMM
 *                 over the kept u of all frames.  mel^ / h_T agree with the layer-by-layer program to rounding (2e-5 / 5e-6 in
 *                 the tests); 0 = the layers as the reference lists them.
 *   "encode_fold": 1 (default) = the same fold in the persistent ENCODE kernel (dec.6's output is not needed there).  The
 *                 folded layer feeds the next state and so the next codes: the same function with another rounding, like
 *                 another order of summation - the goldens and the full-size parity runs give the same bits either way
 *                 (tests/test_gpu_robustness.py compares both settings); 0 = the layers as the reference lists them.
 *   "flow_spin_limit" (polls before a wait inside the persistent kernel gives up; default 4,000,000, more than a second),
 *   "flow_debug_withhold" (1: workgroup 0 of every persistent launch does nothing, so its consumers time out),
 *   "flow_debug_nofill" (1: the plain layer program without filler quanta): test switches.
 * bvc_model_get_option reads "recurrence", "decode_fold", "encode_fold", "flow_resident" (1: the census found a full persistent grid co-resident),
 * "flow_supported" (1: h_dim / z_dim / num_mels are laid out for the persistent kernel), "compute_units". */
int bvc_model_set_option(bvc_model *m, const char *name, int32_t value);
int bvc_model_get_option(const bvc_model *m, const char *name, int32_t *value);

/* The persistent recurrence kernel's workgroups hand activations to each other, so all of them must be resident together;
 * every wait in it is bounded.  If a wait ever times out (a workgroup that never became resident: another process on the
 * device, a CU mask), the kernel still ends, with invalid results, and stores a code (frame << 4 | layer, top bit set) in a
 * status word in host-mapped memory.  EVERY compute entry point (bvc_encode, bvc_decode, bvc_bvrnn_*, bvc_bigvgan,
 * bvc_stft_logmel) reads that word first, without synchronising, and returns BVC_ETIMEOUT - once, clearing it - if an
 * earlier call of this model timed out.  bvc_model_status synchronises the device first, so it also sees calls still in
 * flight; it returns BVC_ETIMEOUT and the code, and clears it; BVC_OK and 0 otherwise. */
int bvc_model_status(const bvc_model *m, uint32_t *code);
/* The same check WITHOUT synchronising: for a caller that has just synchronised by its own means - a blocking device-to-host copy
 * of a call's output, hipStreamSynchronize - and wants to know whether THAT call was valid instead of learning it from the next
 * one.  (bvcodec/model.py calls it behind every copy of an output to a CPU tensor.)  After any report of a time-out the residency
 * census of bvc_model_create runs again before the model's next persistent launch: a tenant that arrived on the device later
 * moves the model to the launch-per-layer schedule instead of letting every call time out. */
int bvc_model_poll_status(const bvc_model *m, uint32_t *code);

/* A persistent recurrence launch needs every compute unit of the device.  Work of the caller's own that holds compute units
 * for an unbounded time - above all an RCCL collective, whose kernel waits for its peers - must not be running beside it.
 * bvc_flow_fence(stream) marks the work issued on `stream` so far: the next persistent launch of this process on the current
 * device (any model, any stream) starts only after it has finished.  Cheap (one event record); not needed when the
 * collective and the codec calls share one stream.  bvcodec/dist.py calls it behind every gather. */
int bvc_flow_fence(void *stream);

/* Frames for L samples: floor(L / hop)  (torch.stft center=False after the reflect pad,
 * meldataset.py:72-85).  Returns < 0 when L <= win - hop (reflect pad impossible). */
int64_t bvc_num_frames(const bvc_model *m, int64_t L);
/* Un-trimmed vocoder output length for T frames: 256*T + 294 for the shipped config
 * (models.py:216-217 applied four times with padding=0). */
int64_t bvc_vocoder_length(const bvc_model *m, int64_t T);
/* Bytes of caller-provided device workspace needed by any entry point at batch B, T frames. */
size_t  bvc_workspace_bytes(const bvc_model *m, int32_t B, int64_t T);

/* mel_spectrogram(y*scale, ...) of meldataset.py:60-95 as called at bvrnn_codec_model.py:49-56,
 * including the .permute(0,2,1): d_wav (B,L) -> d_mel (B,T,num_mels), natural-log mel. */
int bvc_stft_logmel(const bvc_model *m, const float *d_wav, int32_t B, int64_t L, float scale,
                    float *d_mel, void *stream);

/* BVRNN.encode (bvrnn.py:163-209).  d_mel (B,T,num_mels); d_bits (B,T) bits per frame (ignored
 * when var_bit=0); d_h0 (B,h_dim) or NULL for zeros.  Outputs: d_codes (B,T,z_dim) in {0,1,0.5};
 * optional d_all_h (B,T,h_dim) = state BEFORE each frame (bvrnn.py:205); optional d_hT (B,h_dim) =
 * state after the last frame; optional d_prob (B,T,z_dim) = sigmoid output before rounding. */
int bvc_bvrnn_encode(const bvc_model *m, const float *d_mel, const float *d_bits, const float *d_h0,
                     int32_t B, int64_t T, float *d_codes, float *d_all_h, float *d_hT,
                     float *d_prob, void *d_ws, size_t ws_bytes, void *stream);

/* BVRNN.decode (bvrnn.py:211-229).  d_codes (B,T,z_dim) -> d_mel (B,T,num_mels), d_hT optional. */
int bvc_bvrnn_decode(const bvc_model *m, const float *d_codes, const float *d_h0, int32_t B,
                     int64_t T, float *d_mel, float *d_hT, void *d_ws, size_t ws_bytes,
                     void *stream);

/* BVRNN.forward (bvrnn.py:86-160), forward VALUES only (no autograd): the training-time pass with the
 * stochastic Bernoulli sampler round(u - 0.5 + p), the prior net and the KL term.
 *   d_mel (B,T,num_mels), d_bits (B,T) (NULL unless var_bit);
 *   h_use_gen: HOST array of T bytes, use_gen[t] = (random_num_t < p_use_gen), bvrnn.py:111-120: frame t is
 *     conditioned on h2 (the state fed with generated features) when set, else on h (teacher-forced);
 *   update_h = (p_use_gen < 1), update_h2 = (p_use_gen > 0)  (bvrnn.py:142-145);
 *   d_noise (B,T,z_dim) uniform [0,1) samples, or NULL for greedy=True (bvrnn.py:123-126);
 *   outputs: d_dec (B,T,num_mels) = all_dec_mean, d_kld (T) = per-frame KLD terms (the reference returns their
 *   mean, bvrnn.py:160); optional d_z (forward value of z_t, i.e. round(.) - p + p, masked), d_prob (enc_t),
 *   d_prior (prior_t), each (B,T,z_dim).  Needs the prior.{0,2,4}.{weight,bias} tensors at bvc_model_create. */
int bvc_bvrnn_forward(const bvc_model *m, const float *d_mel, const float *d_bits,
                      const uint8_t *h_use_gen, int32_t update_h, int32_t update_h2,
                      const float *d_noise, int32_t B, int64_t T, float *d_dec, float *d_kld,
                      float *d_z, float *d_prob, float *d_prior, void *d_ws, size_t ws_bytes,
                      void *stream);

/* BigVGAN.forward(x, length) (models.py:207-238) followed by `/ out_scale_div`
 * (bvrnn_codec_model.py:71: .squeeze(1) / SCALING).  d_mel is TIME-major (B,T,num_mels), i.e. what
 * bvc_bvrnn_decode emits (the reference permutes to (B,80,T) first).  d_wav (B, n_out) with
 * n_out = min(length, bvc_vocoder_length(T)). */
int bvc_bigvgan(const bvc_model *m, const float *d_mel, int32_t B, int64_t T, int64_t length,
                float out_scale_div, float *d_wav, void *d_ws, size_t ws_bytes, void *stream);

/* Incremental BigVGAN for streaming (not in the reference, whose BigVGAN.forward, models.py:207-238,
 * always sees the whole utterance).  The state keeps the last rows of every activation tensor of the
 * generator for B parallel streams; bvc_vocoder_stream_push runs the generator over only the k new mel
 * frames d_mel (B,k,num_mels) and writes exactly their samples d_wav (B, k*prod(upsample_rates)),
 * equal to the corresponding slice of bvc_bigvgan over the whole utterance.  One in-flight push per
 * state; state buffers are allocated by create (device memory), zeroed by create/reset. */
typedef struct bvc_vocoder_stream bvc_vocoder_stream;
int bvc_vocoder_stream_create(const bvc_model *m, int32_t B, int32_t max_frames_per_push,
                              bvc_vocoder_stream **out);
void bvc_vocoder_stream_destroy(bvc_vocoder_stream *st);
int bvc_vocoder_stream_reset(bvc_vocoder_stream *st, void *stream);
int bvc_vocoder_stream_push(bvc_vocoder_stream *st, const float *d_mel, int32_t k, float out_scale_div,
                            float *d_wav, void *stream);

/* Whole-hop streaming codec (BASELINE.json configs[4]; not in the reference, which has no streaming mode): B parallel
 * streams, `hop_samples` new samples per stream and tick (e.g. 441 = 20 ms).  A tick runs the front-end for the frames the
 * hop completes (frame t needs samples up to 256 t + 768: 34.8 ms of look-ahead, README.md:19), BVRNN.encode and
 * BVRNN.decode with carried GRU states and the incremental vocoder, i.e. encode + decode of exactly those frames, and equals
 * the offline bvc_encode / bvc_decode of the whole signal on them.  The caller writes the hop into d_in (B, hop_samples)
 * before the tick and finds d_codes (B, n_frames, z_dim) and d_wav (B, n_frames * 256) afterwards (buffers owned by the
 * state, fixed addresses).  Schedule: where the persistent recurrence kernel is available (option "recurrence" not 1, the
 * residency census passed, batch within its range) a tick is launched eagerly with ONE persistent launch per recurrence
 * (BVC_STREAM_FLOW=0 turns that off); otherwise, from the 33rd frame on, a tick of launch-per-layer kernels is replayed from a
 * hipGraph captured on first use (one per frame count and vocoder parity; BVC_STREAM_NO_GRAPH=1 keeps eager launches).  The
 * bits are the same on every schedule.  scale / out_scale_div as in bvc_encode / bvc_decode.  One in-flight tick per state;
 * create allocates, tick does not (except the graph instantiation). */
typedef struct bvc_stream_codec bvc_stream_codec;
int  bvc_stream_codec_create(const bvc_model *m, int32_t B, int32_t hop_samples, float bits_per_frame, float scale,
                             float out_scale_div, bvc_stream_codec **out);
void bvc_stream_codec_destroy(bvc_stream_codec *st);
int  bvc_stream_codec_buffers(bvc_stream_codec *st, float **d_in, float **d_codes, float **d_wav, int32_t *max_frames_per_tick);
int  bvc_stream_codec_tick(bvc_stream_codec *st, int32_t *n_frames, void *stream);

/* BVRNNCodecModel.encode (bvrnn_codec_model.py:44-62): scale, log-mel, bits/frame =
 * bits_per_frame for every (b,t), zero initial state, BVRNN.encode.  d_wav (B,L) -> d_codes. */
int bvc_encode(const bvc_model *m, const float *d_wav, int32_t B, int64_t L, float scale,
               float bits_per_frame, float *d_codes, void *d_ws, size_t ws_bytes, void *stream);

/* BVRNNCodecModel.decode (bvrnn_codec_model.py:64-71): zero state, BVRNN.decode, vocoder, /scale. */
int bvc_decode(const bvc_model *m, const float *d_codes, int32_t B, int64_t T, int64_t length,
               float out_scale_div, float *d_wav, void *d_ws, size_t ws_bytes, void *stream);

/* BVRNNCodecModel.forward (bvrnn_codec_model.py:73-76: decode(encode(x, bitrate), x.shape[1])) WITHOUT the second recurrence.  The
 * encoder's frame loop already runs the decoder on every frame (bvrnn.py:198-204) from exactly the states BVRNN.decode would visit
 * again from the same codes (bvrnn.py:222-227), so its outputs are handed to the vocoder directly: one recurrence launch instead of two,
 * no all-frame phi_z GEMMs.  Not a replacement for bvc_encode + bvc_decode (which stay the reference's two operators and the path the
 * benchmark's headline times): the two differ in the ORDER in which dec.0 and the GRU's input gates sum their halves, i.e. by rounding
 * (tests: waveform within 1e-5 RMS of bvc_decode(bvc_encode(x)), codes identical).  d_codes (B, T, z_dim) optional. */
int bvc_forward(const bvc_model *m, const float *d_wav, int32_t B, int64_t L, float scale, float bits_per_frame, int64_t length,
                float out_scale_div, float *d_codes, float *d_wav_out, void *d_ws, size_t ws_bytes, void *stream);

/* Pre-processing of the reference's example.py:15-17.
 * bvc_resample_poly: y = upfirdn(h, x, up, down)[n_pre_remove : n_pre_remove + n_out] with zero padding, i.e.
 * scipy.signal.resample_poly once the caller has designed / padded the filter h (float64, device memory) as
 * scipy does; d_x (B, L_in) -> d_y (B, n_out).  bvc_peak_normalize: x[b,:] /= max|x[b,:]| in place. */
int bvc_resample_poly(const float *d_x, int32_t B, int64_t L_in, const double *d_h, int32_t ntaps, int32_t up,
                      int32_t down, int64_t n_pre_remove, float *d_y, int64_t n_out, void *stream);
int bvc_peak_normalize(float *d_x, int32_t B, int64_t L, void *stream);

/* Wire format for the codes (new: the reference keeps them as float32 {0,1,0.5}, bvrnn.py:191-196, and
 * defines no bit stream).  A frame is ceil(nbits/8) bytes, bit i at byte i/8 position i%8 (LSB first);
 * nbits = min(z_dim, bits per frame) active bits are kept, the masked ones (0.5) are re-created on
 * unpack.  d_bytes (B, T, ceil(nbits/8)) uint8; unpack(pack(codes)) == codes for codes from encode(). */
int bvc_pack_codes(const float *d_codes, int32_t B, int64_t T, int32_t z_dim, int32_t nbits, uint8_t *d_bytes,
                   void *stream);
int bvc_unpack_codes(const uint8_t *d_bytes, int32_t B, int64_t T, int32_t z_dim, int32_t nbits, float *d_codes,
                     void *stream);

/* ---- building blocks exported for the parity tests (tests/ only; same kernels the path uses) */
/* y[M,N] = act(x[M,K] @ w[N,K]^T + bias), act: 0 none, 1 ELU.  Recurrent-step kernel. */
int bvc_test_linear(const float *d_x, const float *d_w, const float *d_bias, int32_t M, int32_t N,
                    int32_t K, int32_t act, float *d_y, void *stream);
/* same contract through the batched (all-frames) GEMM kernel */
int bvc_test_linear_batched(const float *d_x, const float *d_w, const float *d_bias, int32_t M,
                            int32_t N, int32_t K, int32_t act, float *d_y, void *stream);
/* dump of one vocoder intermediate, channels-last: which = 0 conv_pre, 1+2i up_i, 2+2i stage_i.
 * Runs the vocoder up to that point.  d_out (B, len, C); returns len*C via *out_numel_per_batch. */
int bvc_test_vocoder_tap(const bvc_model *m, const float *d_mel, int32_t B, int64_t T, int32_t which,
                         float *d_out, int64_t *out_numel_per_batch, void *d_ws, size_t ws_bytes,
                         void *stream);

/* y[i] = SnakeBeta(x[i]) = x + sin(x*exp(alpha))^2 / (exp(beta) + 1e-9)  (activations.py:107-120), through the
 * device routines of the generator kernels (their sin^2 is a hand-written range reduction, not ocml's sinf). */
int bvc_test_snakebeta(const float *d_x, int64_t n, float alpha, float beta, float *d_y, void *stream);

/* ---- bench instrumentation: in-situ hipEvent timing of one kernel family inside the real schedule.
 * kind: 1 recurrent linear layer, 2 GRU cell, 3 vocoder conv, 4 batched phi_x GEMM, 5 STFT/mel,
 * 6 conv_post.  Every `sample_every`-th launch of that family is bracketed by an event pair on its
 * own stream until `max_samples` pairs are used; bvc_probe_end synchronises the device and returns
 * the mean / min elapsed microseconds.  Not thread-safe; off by default. */
int bvc_probe_begin(int32_t kind, int32_t sample_every, int32_t max_samples);
int bvc_probe_end(double *mean_us, double *min_us, int32_t *n_samples);
/* The recurrent BVRNN kernels are replayed from a hipGraph, where event pairs cannot be inserted:
 * with bvc_kprobe_enable(1) every workgroup of those kernels stamps wall_clock64() (100 MHz) at its
 * start and end into a device buffer; bvc_kprobe_read returns the mean/min kernel duration
 * (first workgroup start -> last workgroup end) over all frames of the LAST bvrnn encode/decode call
 * for the step-kernel indices [node_lo, node_hi).  Encode step: 0 enc.0, 1 enc.2, 2 enc.4(+sigmoid/
 * round/mask), 3-5 phi_z, 6-9 dec, 10-12 phi_x, 13 GRU.  Decode step: 0-3 dec, 4-6 phi_x, 7 GRU. */
int bvc_kprobe_enable(int32_t on);
int bvc_kprobe_read(int32_t node_lo, int32_t node_hi, double *mean_us, double *min_us, int32_t *n_samples);
/* Persistent recurrence only: one wave (workgroup BVC_PROBE_WG, wave BVC_PROBE_WAVE; default 0 / 0) stamps per layer and frame:
 * 0 layer entered, 1 output published; a library built with -DBVC_FLOW_DIAG=1 also 2 flags seen, 3 products done, 4 reduction
 * barrier passed, 5 layer left.  Mean / min of (stamp `to` - stamp `from`) over the frames of the LAST call for the layers
 * [node_lo, node_hi); from = -1 measures from the stamp 1 of the nearest earlier layer that ran (with the folded hop the program has no dec.6
 * node: its slot stays empty and its row reads 0). */
int bvc_kprobe_read_span(int32_t from, int32_t to, int32_t node_lo, int32_t node_hi, double *mean_us, double *min_us,
                         int32_t *n_samples);

#ifdef __cplusplus
}
#endif
#endif /* BVCODEC_H */
