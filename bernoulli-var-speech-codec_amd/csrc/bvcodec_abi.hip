// C ABI of libbvcodec_hip.so (include/bvcodec.h): model creation (weight upload + re-layout into
// MFMA fragment order), workspace carving and the per-call kernel schedules of the
// BVRNNCodecModel encode/decode path.  Host-side only; the kernels are in k_*.hip.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>

#include "bvc_internal.h"

namespace bvc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- sampled hipEvent probes around kernel launches (bench instrumentation, off by default)
struct ProbeState {
    int kind = PK_NONE, every = 1, counter = 0, used = 0;
    std::vector<hipEvent_t> ev;          // pairs
};
static ProbeState g_probe;
static thread_local bool g_capturing = false;   // no event probes while THIS thread captures a stream (captures are thread-local)
static thread_local bool g_stream_tick = false;   // inside bvc_stream_codec_tick: a launch-per-layer recurrence is launched eagerly (the tick itself is the graph)
static thread_local bool g_tick_flow = false;     // ... of a tick that is NOT a graph: its recurrences may take the persistent kernel

// in-kernel timestamp probes for the graph-replayed recurrent kernels (wall_clock64, 100 MHz)
struct KProbe {
    bool enabled = false;
    unsigned long long *dev = nullptr;
    size_t capacity = 0;                   // in u64
    long long T = 0; int nodes = 0;        // geometry of the last probed call
};
static KProbe g_kprobe;

ProbeScope::ProbeScope(int kind, hipStream_t stream) : s(stream), slot(-1) {
    if (g_probe.kind != kind || g_capturing) return;
    if ((g_probe.counter++ % g_probe.every) != 0) return;
    if ((size_t)(g_probe.used + 1) * 2 > g_probe.ev.size()) return;
    slot = g_probe.used++;
    (void)hipEventRecord(g_probe.ev[2 * slot], s);
}
ProbeScope::~ProbeScope() {
    if (slot >= 0) (void)hipEventRecord(g_probe.ev[2 * slot + 1], s);
}

struct Linear { const float *w = nullptr, *wp = nullptr, *b = nullptr; int in = 0, out = 0; };   // w natural, wp fragment-packed

struct AmpPair { ConvLayer c1, c2; };

}  // namespace bvc

using namespace bvc;

struct bvc_model {
    bvc_config cfg;
    std::vector<void *> allocs;
    // front-end
    FrontendTables fe;
    // BVRNN
    const float *mean_mel = nullptr, *std_mel = nullptr;
    Linear phi_x[3], phi_z[3], enc[3], dec[4];
    Linear prior[3];            // only used by bvc_bvrnn_forward; optional (has_prior)
    bool has_prior = false;
    const float *w_ih = nullptr, *w_hh = nullptr, *b_ih = nullptr, *b_hh = nullptr;     // w_*: fragment-packed
    const float *w_ih_il = nullptr, *w_hh_il = nullptr;   // gate-interleaved packing (pack_gru_interleaved): the GRU launches
    const float *w_ih_nat = nullptr;      // natural [3H][2H] copy: the phi_z half is applied to all frames at once in decode
    // vocoder
    ConvLayer conv_pre;
    std::vector<ConvLayer> ups;                       // n_up
    std::vector<std::vector<std::vector<AmpPair>>> amp;   // [stage][kernel][dilation]
    std::vector<int> stage_ch;                        // channels after each upsampler
    const float *post_a = nullptr, *post_ib = nullptr, *post_w = nullptr, *post_b = nullptr;
    int post_c = 0, post_ks = 7;
    // captured recurrent steps (hipGraph), keyed by (kind, batch, workspace)
    // (launch-per-layer schedule only) most recently used first; `idle` is recorded behind the entry's last replay, so an
    // entry is only destroyed once the GPU is done with it
    struct StepGraph { int kind; int B; void *ws; void *probe; hipGraphExec_t exec1, execN; hipEvent_t idle; };
    mutable std::list<StepGraph> graphs;
    mutable std::mutex graph_mu;
    mutable hipStream_t cap_stream = nullptr, side_stream = nullptr;
    mutable std::vector<hipEvent_t> cap_events;
    bool side_branch = false;   // measured SLOWER on MI355X (cross-branch graph dependencies + no spare L2->CU bandwidth): opt-in
    bool use_graph = true;
    bool fused_amp = true;
    unsigned amp_kernels = AMPK_ALL;   // stage-specific generator kernels in use (options vocoder_full_tiles / vocoder_c16_kernel)
    bool precomp_pz = true;     // decode: the phi_z halves of dec.0 and of the GRU input product are batched over all frames
    int mtw = 1;                // 16-row tiles per workgroup in the recurrent kernels (BVC_MTW = 1 | 2 | 4)
    // persistent recurrence (k_flow.hip): hop tables of encode / decode, resident in device memory
    // recurrence schedule: RS_PERSISTENT one launch per call (k_flow.hip), RS_LAYERS one launch per layer (hipGraph replay),
    // RS_AUTO (default) persistent while calls come one at a time, layers while calls of several streams overlap
    int recurrence = 2;         // BVC_RECURRENCE=persistent|layers|auto, bvc_model_set_option("recurrence")
    mutable std::atomic<bool> flow_resident{false}; // the residency census found a full persistent grid co-resident on this device
    mutable std::atomic<bool> census_due{false};    // a recurrence time-out was seen: the census runs again before the next persistent launch (another
                                        // tenant may have arrived after bvc_model_create: the model then moves to the layer schedule)
    mutable hipStream_t census_stream = nullptr;
    mutable unsigned *census_ctr = nullptr;
    int flow_perh = 0;          // k-blocks per wave of an h_dim-sized segment (0: h_dim not supported by the persistent kernel)
    // sticky status word of the persistent kernels, in host-mapped pinned memory: a kernel whose wait timed out stores its
    // code there; every compute entry point reads it WITHOUT synchronising (h_status) and reports BVC_ETIMEOUT once
    volatile unsigned *h_status = nullptr;
    unsigned *d_status = nullptr;       // device address of the same word
    int cu_count = 0;                   // compute units of the device: a persistent launch needs one per workgroup
    unsigned flow_spin_limit = 4000000u;   // polls before a wait gives up (> 1 s: only a workgroup that never became resident gets there)
    int flow_debug_withhold = 0;        // tests only: workgroup 0 of a persistent launch returns at once (its peers time out)
    int flow_debug_nofill = 0;          // tests only: no filler quanta (the plain layer program)
    // decode_fold / encode_fold (default 1): the persistent kernels run phi_x.0(norm(dec.6(u))) - three maps without a non-linearity
    // between them (bvrnn.py:80, :204 / :226) - as the one affine map px0_dec3 (folded in float64 at model creation): one wide layer
    // instead of two narrow hops per frame.  Decode computes dec.6 itself, the decoder's output, as one batched GEMM behind the
    // launch; encode does not need it (BVRNN.encode returns codes and states only).  In encode the folded layer feeds the next
    // state and with it the next codes: same function, another rounding (like another order of summation) - every golden and the
    // full-size parity runs give the same bits and the same largest probability deviation (1.2e-7) with and without it.
    Linear px0_dec3{};
    int decode_fold = 1;
    int encode_fold = 1;

    ~bvc_model() {
        for (auto &g : graphs) { (void)hipGraphExecDestroy(g.exec1); (void)hipGraphExecDestroy(g.execN); if (g.idle) (void)hipEventDestroy(g.idle); }
        if (census_stream) (void)hipStreamDestroy(census_stream);
        if (census_ctr) (void)hipFree(census_ctr);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        for (auto e : cap_events) (void)hipEventDestroy(e);
        if (h_status) (void)hipHostFree(const_cast<unsigned *>(h_status));
        for (void *p : allocs) (void)hipFree(p);
    }
};

namespace {

typedef std::map<std::string, const bvc_tensor *> TensorMap;

template <typename T>
int upload(bvc_model *m, const std::vector<T> &host, const T **dev) {
    void *d = nullptr;
    const size_t bytes = host.size() * sizeof(T);
    BVC_HIP_TRY(hipMalloc(&d, bytes ? bytes : 16));
    m->allocs.push_back(d);
    if (bytes) BVC_HIP_TRY(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(d);
    return BVC_OK;
}

int upload_raw(bvc_model *m, const float *h, int64_t n, const float **dev) {
    void *d = nullptr;
    BVC_HIP_TRY(hipMalloc(&d, (size_t)n * sizeof(float)));
    m->allocs.push_back(d);
    BVC_HIP_TRY(hipMemcpy(d, h, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    *dev = static_cast<const float *>(d);
    return BVC_OK;
}

const bvc_tensor *find(const TensorMap &tm, const std::string &name, int64_t numel) {
    auto it = tm.find(name);
    if (it == tm.end()) { set_error("missing tensor '%s'", name.c_str()); return nullptr; }
    if (it->second->numel != numel || !it->second->h_data) {
        set_error("tensor '%s' has %lld elements, expected %lld", name.c_str(),
                  (long long)it->second->numel, (long long)numel);
        return nullptr;
    }
    return it->second;
}

// Linear weight W[N][K] (row-major) -> MFMA B-operand fragment order [N/16][K/16][lane][4]:
// lane = ((k%16)/4)*16 + n%16 holds W[n][k..k+3]; one (n-tile, k-block) pair is 1 KiB contiguous.
std::vector<float> pack_linear(const float *W, int N, int K) {
    std::vector<float> p((size_t)N * K);
    const int nb = K / 16;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k)
            p[((((size_t)(n >> 4) * nb + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) << 2) + (k & 3)] = W[(size_t)n * K + k];
    return p;
}

// GRU weight W[3H][K] (gates r, z, n stacked, PyTorch order) -> [H/16][K/16][gate][lane][4]: the three gates' fragments of
// one (feature tile, k-block) are 3 KiB contiguous.  With the gates 4-8 MB apart (pack_linear) a wave's three loads per
// k-block hit the same L2 channel; interleaved, the GRU launch is 10 % shorter alone and 24 % in the aggregate of three
// concurrent chains (tools/gru_splitk_bench.hip).
std::vector<float> pack_gru_interleaved(const float *W, int H, int K) {
    std::vector<float> p((size_t)3 * H * K);
    const int nb = K / 16;
    for (int q = 0; q < 3; ++q)
        for (int n = 0; n < H; ++n)
            for (int k = 0; k < K; ++k)
                p[(((((size_t)(n >> 4) * nb + (k >> 4)) * 3 + q) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) << 2) + (k & 3)] =
                    W[((size_t)q * H + n) * K + k];
    return p;
}

int load_linear(bvc_model *m, const TensorMap &tm, const std::string &name, int in, int out, Linear *l) {
    const bvc_tensor *w = find(tm, name + ".weight", (int64_t)in * out);
    if (!w) return BVC_EMISSING;
    const bvc_tensor *b = find(tm, name + ".bias", out);
    if (!b) return BVC_EMISSING;
    l->in = in; l->out = out;
    int rc;
    if ((rc = upload_raw(m, w->h_data, w->numel, &l->w))) return rc;          // natural: batched GEMM
    if ((rc = upload(m, pack_linear(w->h_data, out, in), &l->wp))) return rc;  // packed: recurrent kernels
    return upload_raw(m, b->h_data, b->numel, &l->b);
}

// Conv1d weight W[cout][cin][ks] -> MFMA B fragments [ks][cin/4][ntiles][64]
std::vector<float> pack_conv(const float *W, int cout, int cin, int ks) {
    const int c4 = cin / 4, ntiles = (cout + 15) / 16;
    std::vector<float> p((size_t)ks * c4 * ntiles * 64, 0.0f);
    for (int j = 0; j < ks; ++j)
        for (int cg = 0; cg < c4; ++cg)
            for (int nt = 0; nt < ntiles; ++nt)
                for (int l = 0; l < 64; ++l) {
                    const int co = nt * 16 + (l & 15), ci = cg * 4 + (l >> 4);
                    if (co < cout)
                        p[(((size_t)j * c4 + cg) * ntiles + nt) * 64 + l] = W[((size_t)co * cin + ci) * ks + j];
                }
    return p;
}

// Conv1d weight W[cout][cin][ks] -> [ks][cin/16][ntiles][64][4]: the fragments of pack_conv for four consecutive k-steps side by side
std::vector<float> pack_conv_k4(const float *W, int cout, int cin, int ks) {
    const int g4 = cin / 16, ntiles = (cout + 15) / 16;
    std::vector<float> p((size_t)ks * g4 * ntiles * 64 * 4, 0.0f);
    for (int j = 0; j < ks; ++j)
        for (int q = 0; q < g4; ++q)
            for (int nt = 0; nt < ntiles; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int u = 0; u < 4; ++u) {
                        const int co = nt * 16 + (l & 15), ci = (q * 4 + u) * 4 + (l >> 4);
                        if (co < cout)
                            p[((((size_t)j * g4 + q) * ntiles + nt) * 64 + l) * 4 + u] = W[((size_t)co * cin + ci) * ks + j];
                    }
    return p;
}

// Conv1d weight W[8][8][ks] -> B fragments [ks+1][2][64] of the two-rows-per-tile form (k_vocoder.hip, amp_pair8_kernel):
// column n = p*8 + co of k-step k holds W[co][ci][k - p] (zero outside the kernel)
std::vector<float> pack_conv_two_rows(const float *W, int ks) {
    std::vector<float> p((size_t)(ks + 1) * 2 * 64, 0.0f);
    for (int k = 0; k <= ks; ++k)
        for (int cg = 0; cg < 2; ++cg)
            for (int l = 0; l < 64; ++l) {
                const int n = l & 15, pr = n >> 3, co = n & 7, ci = cg * 4 + (l >> 4), j = k - pr;
                if (j >= 0 && j < ks) p[((size_t)k * 2 + cg) * 64 + l] = W[((size_t)co * 8 + ci) * ks + j];
            }
    return p;
}

// ConvTranspose1d weight W[cin][cout][2u] -> 2-tap conv with u*cout columns (polyphase form)
std::vector<float> convt_as_conv(const float *W, int cin, int cout, int u) {
    const int k = 2 * u, ncol = u * cout;
    std::vector<float> v((size_t)ncol * cin * 2);
    for (int p = 0; p < u; ++p)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                const size_t n = (size_t)p * cout + co;
                v[(n * cin + ci) * 2 + 0] = W[((size_t)ci * cout + co) * k + p + u];   // tap on in[q-1]
                v[(n * cin + ci) * 2 + 1] = W[((size_t)ci * cout + co) * k + p];       // tap on in[q]
            }
    return v;
}

int make_conv(bvc_model *m, const float *W, const float *bias, int nbias_rep, int cout, int cin, int ks, int dil,
              const float *alpha, const float *beta, ConvLayer *c) {
    c->cin = cin; c->cout = cout; c->ntiles = (cout + 15) / 16; c->ks = ks; c->dil = dil;
    c->act_a = c->act_ib = nullptr;
    int rc;
    std::vector<float> wp = pack_conv(W, cout, cin, ks);
    if ((rc = upload(m, wp, &c->wp))) return rc;
    c->wp2 = nullptr;
    c->wp4 = nullptr;
    if (cin == cout && cin >= 32 && cin % 16 == 0 && alpha && (rc = upload(m, pack_conv_k4(W, cout, cin, ks), &c->wp4))) return rc;
    if (cin == 8 && cout == 8 && (rc = upload(m, pack_conv_two_rows(W, ks), &c->wp2))) return rc;
    std::vector<float> b((size_t)cout);
    const int per = cout / nbias_rep;
    for (int i = 0; i < cout; ++i) b[i] = bias[i % per];
    if ((rc = upload(m, b, &c->bias))) return rc;
    if (alpha) {
        std::vector<float> a(cin), ib(cin);
        for (int i = 0; i < cin; ++i) {
            a[i] = (float)std::exp((double)alpha[i]);                      // torch.exp(alpha)
            const float eb = (float)std::exp((double)beta[i]);
            ib[i] = 1.0f / (eb + 0.000000001f);                            // activations.py:116
        }
        if ((rc = upload(m, a, &c->act_a))) return rc;
        if ((rc = upload(m, ib, &c->act_ib))) return rc;
    }
    return BVC_OK;
}

int build_frontend(bvc_model *m, const TensorMap &tm) {
    const bvc_config &c = m->cfg;
    const int nfft = c.n_fft, nbins = nfft / 2 + 1;
    const double PI = 3.14159265358979323846;
    std::vector<float> win(nfft);
    auto itw = tm.find("hann_window");
    if (itw != tm.end() && itw->second->numel == nfft) {
        memcpy(win.data(), itw->second->h_data, sizeof(float) * nfft);
    } else {
        for (int n = 0; n < nfft; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / nfft));
    }
    std::vector<float2> tw1(8 * 64), tw2(64), tws(nbins);
    for (int k = 0; k < 8; ++k)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * PI * (double)(l * k) / 512.0;
            tw1[k * 64 + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k = 0; k < 8; ++k)
        for (int n = 0; n < 8; ++n) {
            const double a = -2.0 * PI * (double)(n * k) / 64.0;
            tw2[k * 8 + n] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k = 0; k < nbins; ++k) {
        const double a = -2.0 * PI * (double)k / 1024.0;
        tws[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    const bvc_tensor *mb = find(tm, "mel_basis", (int64_t)c.num_mels * nbins);
    if (!mb) return BVC_EMISSING;
    std::vector<int> st(c.num_mels), ln(c.num_mels), off(c.num_mels);
    std::vector<float> w;
    int kmax = 1;
    for (int j = 0; j < c.num_mels; ++j) {
        const float *row = mb->h_data + (size_t)j * nbins;
        int lo = -1, hi = -1;
        for (int k = 0; k < nbins; ++k)
            if (row[k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { lo = 0; hi = -1; }
        st[j] = lo; ln[j] = hi - lo + 1; off[j] = (int)w.size();
        for (int k = lo; k <= hi; ++k) w.push_back(row[k]);
        if (hi + 1 > kmax) kmax = hi + 1;
    }
    FrontendTables &t = m->fe;
    int rc;
    if ((rc = upload(m, win, &t.window))) return rc;
    if ((rc = upload(m, tw1, &t.tw1))) return rc;
    if ((rc = upload(m, tw2, &t.tw2))) return rc;
    if ((rc = upload(m, tws, &t.tws))) return rc;
    if ((rc = upload(m, st, &t.mel_start))) return rc;
    if ((rc = upload(m, ln, &t.mel_len))) return rc;
    if ((rc = upload(m, off, &t.mel_off))) return rc;
    if ((rc = upload(m, w, &t.mel_w))) return rc;
    t.num_mels = c.num_mels;
    t.kmax = kmax;
    return BVC_OK;
}

int build_bvrnn(bvc_model *m, const TensorMap &tm) {
    const int X = m->cfg.num_mels, H = m->cfg.h_dim, Z = m->cfg.z_dim;
    int rc;
    const bvc_tensor *t;
    if (!(t = find(tm, "mean_mel", X))) return BVC_EMISSING;
    if ((rc = upload_raw(m, t->h_data, X, &m->mean_mel))) return rc;
    if (!(t = find(tm, "std_mel", X))) return BVC_EMISSING;
    if ((rc = upload_raw(m, t->h_data, X, &m->std_mel))) return rc;
    const int px_in[3] = {X, H, H}, pz_in[3] = {Z, H, H}, en_in[3] = {2 * H, H, H}, en_out[3] = {H, H, Z};
    const int de_in[4] = {2 * H, H, H, H}, de_out[4] = {H, H, H, X};
    for (int i = 0; i < 3; ++i) {
        const std::string idx = std::to_string(2 * i);
        if ((rc = load_linear(m, tm, "phi_x." + idx, px_in[i], H, &m->phi_x[i]))) return rc;
        if ((rc = load_linear(m, tm, "phi_z." + idx, pz_in[i], H, &m->phi_z[i]))) return rc;
        if ((rc = load_linear(m, tm, "enc." + idx, en_in[i], en_out[i], &m->enc[i]))) return rc;
    }
    for (int i = 0; i < 4; ++i)
        if ((rc = load_linear(m, tm, "dec." + std::to_string(2 * i), de_in[i], de_out[i], &m->dec[i]))) return rc;
    if (tm.count("prior.0.weight")) {        // training-time prior net (bvrnn.py:68-73): needed by bvc_bvrnn_forward only
        const int pr_out[3] = {H, H, Z};
        for (int i = 0; i < 3; ++i)
            if ((rc = load_linear(m, tm, "prior." + std::to_string(2 * i), H, pr_out[i], &m->prior[i]))) return rc;
        m->has_prior = true;
    }
    {   // px0_dec3: W = phi_x.0.W diag(1/std) dec.6.W  (H x H),  b = phi_x.0.W ((dec.6.b - mean) / std) + phi_x.0.b, in float64
        const float *wp0 = tm.at("phi_x.0.weight")->h_data, *bp0 = tm.at("phi_x.0.bias")->h_data;      // [H][X], [H]
        const float *wd6 = tm.at("dec.6.weight")->h_data, *bd6 = tm.at("dec.6.bias")->h_data;          // [X][H], [X]
        const float *mean = tm.at("mean_mel")->h_data, *stdv = tm.at("std_mel")->h_data;
        std::vector<float> wc((size_t)H * H), bc((size_t)H);
        std::vector<double> row((size_t)H);
        for (int n = 0; n < H; ++n) {
            std::fill(row.begin(), row.end(), 0.0);
            double b = (double)bp0[n];
            for (int j = 0; j < X; ++j) {
                const double f = (double)wp0[(size_t)n * X + j] / (double)stdv[j];
                b += f * ((double)bd6[j] - (double)mean[j]);
                const float *wr = wd6 + (size_t)j * H;
                for (int k = 0; k < H; ++k) row[k] += f * (double)wr[k];
            }
            for (int k = 0; k < H; ++k) wc[(size_t)n * H + k] = (float)row[k];
            bc[n] = (float)b;
        }
        if (getenv("BVC_DECODE_FOLD") && getenv("BVC_DECODE_FOLD")[0] == '0') m->decode_fold = 0;     // A/B runs (tools/flow_variants.py)
        if (getenv("BVC_ENCODE_FOLD") && getenv("BVC_ENCODE_FOLD")[0] == '0') m->encode_fold = 0;
        m->px0_dec3.in = H; m->px0_dec3.out = H;
        m->px0_dec3.w = nullptr;                                   // (only the recurrent kernels use it)
        if ((rc = upload(m, pack_linear(wc.data(), H, H), &m->px0_dec3.wp))) return rc;
        if ((rc = upload(m, bc, &m->px0_dec3.b))) return rc;
    }
    if (!(t = find(tm, "rnn.weight_ih_l0", (int64_t)3 * H * 2 * H))) return BVC_EMISSING;
    if ((rc = upload(m, pack_linear(t->h_data, 3 * H, 2 * H), &m->w_ih))) return rc;
    if ((rc = upload(m, pack_gru_interleaved(t->h_data, H, 2 * H), &m->w_ih_il))) return rc;
    if ((rc = upload_raw(m, t->h_data, t->numel, &m->w_ih_nat))) return rc;
    if (!(t = find(tm, "rnn.weight_hh_l0", (int64_t)3 * H * H))) return BVC_EMISSING;
    if ((rc = upload(m, pack_linear(t->h_data, 3 * H, H), &m->w_hh))) return rc;
    if ((rc = upload(m, pack_gru_interleaved(t->h_data, H, H), &m->w_hh_il))) return rc;
    if (!(t = find(tm, "rnn.bias_ih_l0", 3 * H))) return BVC_EMISSING;
    if ((rc = upload_raw(m, t->h_data, t->numel, &m->b_ih))) return rc;
    if (!(t = find(tm, "rnn.bias_hh_l0", 3 * H))) return BVC_EMISSING;
    if ((rc = upload_raw(m, t->h_data, t->numel, &m->b_hh))) return rc;
    return BVC_OK;
}

int build_vocoder(bvc_model *m, const TensorMap &tm) {
    const bvc_config &c = m->cfg;
    int rc;
    const bvc_tensor *w, *b;
    const int c0 = c.upsample_initial_channel;
    if (!(w = find(tm, "conv_pre.weight", (int64_t)c0 * c.num_mels * 7))) return BVC_EMISSING;
    if (!(b = find(tm, "conv_pre.bias", c0))) return BVC_EMISSING;
    if ((rc = make_conv(m, w->h_data, b->h_data, 1, c0, c.num_mels, 7, 1, nullptr, nullptr, &m->conv_pre))) return rc;
    int ch = c0;
    m->ups.resize(c.n_up);
    m->amp.resize(c.n_up);
    m->stage_ch.resize(c.n_up);
    for (int i = 0; i < c.n_up; ++i) {
        const int u = c.up_rates[i], cin = ch, cout = ch / 2;
        const std::string nm = "ups." + std::to_string(i) + ".1";
        if (!(w = find(tm, nm + ".weight", (int64_t)cin * cout * 2 * u))) return BVC_EMISSING;
        if (!(b = find(tm, nm + ".bias", cout))) return BVC_EMISSING;
        std::vector<float> wv = convt_as_conv(w->h_data, cin, cout, u);
        if ((rc = make_conv(m, wv.data(), b->h_data, u, u * cout, cin, 2, 1, nullptr, nullptr, &m->ups[i]))) return rc;
        ch = cout;
        m->stage_ch[i] = ch;
        m->amp[i].resize(c.n_resk);
        for (int j = 0; j < c.n_resk; ++j) {
            const int ks = c.res_kernels[j];
            const std::string pre = "resblocks." + std::to_string(i * c.n_resk + j);
            m->amp[i][j].resize(3);
            for (int d = 0; d < 3; ++d) {
                const bvc_tensor *a1, *b1, *a2, *b2, *w1, *bb1, *w2, *bb2;
                const std::string ds = std::to_string(d);
                if (!(a1 = find(tm, pre + ".activations." + std::to_string(2 * d) + ".alpha", ch))) return BVC_EMISSING;
                if (!(b1 = find(tm, pre + ".activations." + std::to_string(2 * d) + ".beta", ch))) return BVC_EMISSING;
                if (!(a2 = find(tm, pre + ".activations." + std::to_string(2 * d + 1) + ".alpha", ch))) return BVC_EMISSING;
                if (!(b2 = find(tm, pre + ".activations." + std::to_string(2 * d + 1) + ".beta", ch))) return BVC_EMISSING;
                if (!(w1 = find(tm, pre + ".convs1." + ds + ".weight", (int64_t)ch * ch * ks))) return BVC_EMISSING;
                if (!(bb1 = find(tm, pre + ".convs1." + ds + ".bias", ch))) return BVC_EMISSING;
                if (!(w2 = find(tm, pre + ".convs2." + ds + ".weight", (int64_t)ch * ch * ks))) return BVC_EMISSING;
                if (!(bb2 = find(tm, pre + ".convs2." + ds + ".bias", ch))) return BVC_EMISSING;
                AmpPair &ap = m->amp[i][j][d];
                if ((rc = make_conv(m, w1->h_data, bb1->h_data, 1, ch, ch, ks, c.res_dilations[j][d], a1->h_data,
                                    b1->h_data, &ap.c1))) return rc;
                if ((rc = make_conv(m, w2->h_data, bb2->h_data, 1, ch, ch, ks, 1, a2->h_data, b2->h_data, &ap.c2)))
                    return rc;
            }
        }
    }
    m->post_c = ch;
    const bvc_tensor *pa, *pb;
    if (!(pa = find(tm, "activation_post.alpha", ch))) return BVC_EMISSING;
    if (!(pb = find(tm, "activation_post.beta", ch))) return BVC_EMISSING;
    std::vector<float> a(ch), ib(ch);
    for (int i = 0; i < ch; ++i) {
        a[i] = (float)std::exp((double)pa->h_data[i]);
        ib[i] = 1.0f / ((float)std::exp((double)pb->h_data[i]) + 0.000000001f);
    }
    if ((rc = upload(m, a, &m->post_a))) return rc;
    if ((rc = upload(m, ib, &m->post_ib))) return rc;
    if (!(w = find(tm, "conv_post.weight", (int64_t)ch * 7))) return BVC_EMISSING;
    if (!(b = find(tm, "conv_post.bias", 1))) return BVC_EMISSING;
    if ((rc = upload_raw(m, w->h_data, w->numel, &m->post_w))) return rc;
    if ((rc = upload_raw(m, b->h_data, 1, &m->post_b))) return rc;
    return BVC_OK;
}

int check_config(const bvc_config *c) {
    if (!c) { set_error("null config"); return BVC_EINVAL; }
    if (c->n_fft != 1024 || c->hop != 256) { set_error("front-end kernel needs n_fft=1024, hop=256"); return BVC_EINVAL; }
    if (c->pad_left < 0 || c->pad_left > c->n_fft - c->hop) { set_error("pad_left out of range"); return BVC_EINVAL; }
    if (c->num_mels % 16 || c->h_dim % 16 || c->z_dim % 16 || c->num_mels > 128) {
        set_error("num_mels/h_dim/z_dim must be multiples of 16 (num_mels <= 128)"); return BVC_EINVAL; }
    if (c->n_up < 1 || c->n_up > 8 || c->n_resk < 1 || c->n_resk > 4) { set_error("bad n_up / n_resk"); return BVC_EINVAL; }
    int ch = c->upsample_initial_channel;
    if (ch != 128 && ch != 64 && ch != 32 && ch != 16) { set_error("unsupported upsample_initial_channel %d", ch); return BVC_EINVAL; }
    for (int i = 0; i < c->n_up; ++i) {
        if (c->up_kernels[i] != 2 * c->up_rates[i]) { set_error("upsample kernel must be 2*rate"); return BVC_EINVAL; }
        ch /= 2;
        if (ch < 8) { set_error("too many upsampling stages for %d initial channels", c->upsample_initial_channel); return BVC_EINVAL; }
    }
    if (ch != 8) { set_error("final channel count must be 8 (got %d)", ch); return BVC_EINVAL; }
    if (c->num_mels != 80) { set_error("conv_pre kernel is built for num_mels=80"); return BVC_EINVAL; }
    return BVC_OK;
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- workspace layout ---------------------------------------------------------------------------
struct Workspace {
    // encode
    float *yn, *pxA, *pxB, *pxC;
    float *step[16];            // per-step [B, max(H, ...)] scratch vectors
    float *hbuf;                // [2][B][H] GRU state ping-pong (parity of the frame counter)
    float *part_i, *part_h, *part_d;   // side-branch partial sums: W_ih[:,H:] phi_z + b_ih, W_hh h + b_hh, dec.0[:,H:] h
    CallDesc *desc;             // per-call dynamic state read by the captured step kernels
    float *mel, *bits;          // facade-level buffers
    float *part_dec0, *part_gru; // decode: dec.0[:, :H] phi_z + b (B,T,H) and W_ih[:, H:] phi_z + b_ih (B,T,3H), all frames
    float *flow;                // persistent recurrence: FB_COUNT x 2 fragment-packed [mt16][dmax] activation buffers
    size_t flow_slot;           // floats per flow buffer
    FlowArgs *flow_args;        // device copy of the persistent kernel's arguments
    // vocoder
    float *y0, *X, *P, *Q, *U, *XS;
    size_t total;
};

int64_t stage_len(const bvc_model *m, int64_t T, int stage) {    // length after upsampler `stage`
    int64_t L = T;
    for (int i = 0; i <= stage; ++i) L = (L + 1) * m->cfg.up_rates[i];
    return L;
}

void carve(const bvc_model *m, int B, int64_t T, char *base, Workspace *w) {
    const bvc_config &c = m->cfg;
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float *p = reinterpret_cast<float *>(base + off);
        off += align_up(nfloats * sizeof(float));
        return p;
    };
    const size_t BT = (size_t)B * (size_t)T;
    const int H = c.h_dim;
    const int vmax = H > c.num_mels ? H : c.num_mels;
    const size_t mt16 = (size_t)((B + 15) / 16) * 16;          // fragment-packed matrices hold whole 16-row tiles
    // buffers referenced by the captured step graphs come first: their offsets depend on B only, so a
    // graph captured for (B, workspace) stays valid for every T
    for (int i = 0; i < 16; ++i) w->step[i] = take(mt16 * vmax);
    w->hbuf = take(2 * mt16 * H);
    w->part_i = take(mt16 * 3 * H);
    w->part_h = take(mt16 * 3 * H);
    w->part_d = take(mt16 * H);
    w->desc = reinterpret_cast<CallDesc *>(take(64));
    {
        int dmax = H > c.num_mels ? H : c.num_mels;
        if (c.z_dim > dmax) dmax = c.z_dim;
        w->flow_slot = mt16 * (size_t)dmax;
        w->flow = take((size_t)FB_COUNT * 2 * w->flow_slot);
        w->flow_args = reinterpret_cast<FlowArgs *>(take((sizeof(FlowArgs) + 3) / 4));
    }
    w->yn = take(BT * c.num_mels);
    w->pxA = take(mt16 * (size_t)T * H);                       // final phi_x / phi_z: frame-packed
    w->pxB = take(mt16 * (size_t)T * H);                       // intermediates of the batched MLPs: frame-major rows
    w->pxC = take(mt16 * (size_t)T * H);
    w->mel = take(BT * c.num_mels);
    w->bits = take(BT);
    w->part_dec0 = take(BT * H);
    w->part_gru = take(BT * 3 * H);
    size_t maxel = 0;
    for (int i = 0; i < c.n_up; ++i) {
        const size_t e = (size_t)stage_len(m, T, i) * m->stage_ch[i];
        if (e > maxel) maxel = e;
    }
    w->y0 = take((size_t)B * T * c.upsample_initial_channel);
    w->X = take((size_t)B * maxel);
    w->P = take((size_t)B * maxel);
    w->Q = take((size_t)B * maxel);
    w->U = take((size_t)B * maxel);
    w->XS = take((size_t)B * maxel);
    w->total = off;
}

int check_ws(const bvc_model *m, int B, int64_t T, void *d_ws, size_t ws_bytes, Workspace *w) {
    if (!m) { set_error("null model"); return BVC_EINVAL; }
    if (B <= 0 || T <= 0) { set_error("B and T must be positive (B=%d, T=%lld)", B, (long long)T); return BVC_EINVAL; }
    carve(m, B, T, static_cast<char *>(d_ws), w);
    if (!d_ws || ws_bytes < w->total) {
        set_error("workspace too small: %zu bytes given, %zu needed", ws_bytes, w->total);
        return BVC_ENOMEM;
    }
    return BVC_OK;
}

inline GemmSeg mkseg(DynPtr x, const float *w, int wnb, int K, int grp) { return GemmSeg{w, wnb, K, x, grp, 0}; }

// keeps nb_total consistent with the segments
inline void finish(GemmParams &p) { p.nb_total = 0; for (int i = 0; i < p.nseg; ++i) p.nb_total += p.seg[i].K / 16; }

GemmParams lin_params(const Linear &l, DynPtr x, int M, DynPtr y) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.nseg = 1;
    p.seg[0] = mkseg(x, l.wp, l.in / 16, l.in, 0);
    p.M = M; p.N = l.out; p.gate_rows = 0;
    p.bias0 = l.b;
    p.y = y;
    finish(p);
    return p;
}

// linear over the concatenation [x1 | x2] (torch.cat at bvrnn.py:189,202)
GemmParams lin2_params(const Linear &l, DynPtr x1, int K1, DynPtr x2, int K2, int M, DynPtr y) {
    GemmParams p = lin_params(l, x1, M, y);
    p.nseg = 2;
    p.seg[0] = mkseg(x1, l.wp, l.in / 16, K1, 0);
    p.seg[1] = mkseg(x2, l.wp + (size_t)(K1 / 16) * 256, l.in / 16, K2, 0);
    finish(p);
    return p;
}

// One operation of a step: a kernel on the main or the side branch, or an event record / wait that
// forks and joins the two branches (they become graph dependencies under stream capture).
enum { OP_KERNEL = 0, OP_RECORD = 1, OP_WAIT = 2 };
enum { BR_MAIN = 0, BR_SIDE = 1 };
struct StepNode { int op; int branch; int event; GemmParams p; int epi; };
enum { STEP_ENCODE = 0, STEP_DECODE = 1, STEP_DECODE_PRE = 2 };   // _PRE: phi_z halves of dec.0 / GRU arrive pre-computed
enum { STEP_KIND_MASK = 0xF, STEP_FOLD = 0x10 };                  // | STEP_FOLD: the folded hop (step_fold below)
constexpr int64_t SMALL_T_FRAMES = 4;          // up to this many frames per call the all-frame MLPs run frame by frame on the recurrent-layer kernel
enum { EV_START = 0, EV_DEC0H = 1, EV_PZ = 2, EV_GATES = 3, EV_COUNT = 4 };

// The operation sequence of ONE frame.  Every pointer is either workspace-static, frame-indexed through
// the call descriptor, or parity-indexed (GRU state), so the same sequence serves every frame and
// every call: it is captured once into a hipGraph.  Internal activations are kept in MFMA fragment
// order (packed=1) so every operand load is a coalesced 1 KiB read.
//   encode (bvrnn.py:187-206): enc -> sigmoid/round/mask -> phi_z -> dec -> phi_x(norm) -> GRU
//   decode (bvrnn.py:222-227): [phi_z batched over all frames beforehand] dec -> phi_x(norm) -> GRU
// Side branch: the halves of the split dot products that do not depend on the current frame's chain -
// dec.0[:, H:] h, W_hh h + b_hh, W_ih[:, H:] phi_z + b_ih - run concurrently with the chain (which is
// latency-bound), so the GRU kernel on the critical path only streams W_ih[:, :H].
std::vector<StepNode> build_step(const bvc_model *m, const Workspace &w, int B, int kind_and_fold) {
    const int kind = kind_and_fold & STEP_KIND_MASK;
    const bool fold = (kind_and_fold & STEP_FOLD) != 0;       // dec.6 -> norm -> phi_x.0 as one layer (bvc_model::px0_dec3)
    const int H = m->cfg.h_dim, Z = m->cfg.z_dim, X = m->cfg.num_mels;
    std::vector<StepNode> plan;
    const long long MH = (long long)((B + 15) / 16) * 16 * H;
    const DynPtr h_cur = dp_parity(w.hbuf, H, MH, 0, 1);
    const DynPtr h_next = dp_parity(w.hbuf + MH, H, -MH, 0, 1);
    float *e1 = w.step[0], *e2 = w.step[1];
    float *pz1 = w.step[2], *pz2 = w.step[3], *pz3 = w.step[4];
    float *d1 = w.step[5], *d2 = w.step[6], *d3 = w.step[7], *dn = w.step[8];
    float *g1 = w.step[9], *g2 = w.step[10], *g3 = w.step[11];
    auto S = [&](float *p, int ld) { return dp_static(p, ld, 1); };
    const bool side = m->side_branch;
    int node = 0;
    auto K = [&](int branch, GemmParams p, int epi) {
        p.desc = w.desc; p.node = node++;
        p.probe = g_kprobe.enabled ? g_kprobe.dev : nullptr;
        finish(p);
        plan.push_back(StepNode{OP_KERNEL, branch, -1, p, epi});
    };
    auto REC = [&](int branch, int ev) { GemmParams z; memset(&z, 0, sizeof(z)); if (side) plan.push_back(StepNode{OP_RECORD, branch, ev, z, 0}); };
    auto WAIT = [&](int branch, int ev) { GemmParams z; memset(&z, 0, sizeof(z)); if (side) plan.push_back(StepNode{OP_WAIT, branch, ev, z, 0}); };
    // --- side-branch kernels (plain linears into natural [B][.] partial buffers)
    auto side_dec0h = [&]() {       // dec.0.weight[:, H:] @ h            (no bias: added on the main branch)
        GemmParams p = lin_params(m->dec[0], h_cur, B, dp_static(w.part_d, H));
        p.seg[0] = mkseg(h_cur, m->dec[0].wp + (size_t)(H / 16) * 256, 2 * H / 16, H, 0);
        p.bias0 = nullptr;
        K(BR_SIDE, p, EPI_LINEAR);
    };
    auto side_hh = [&]() {          // W_hh @ h + b_hh
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.nseg = 1;
        p.seg[0] = mkseg(h_cur, m->w_hh, H / 16, H, 0);
        p.M = B; p.N = 3 * H; p.bias0 = m->b_hh;
        p.y = dp_static(w.part_h, 3 * H);
        K(BR_SIDE, p, EPI_LINEAR);
    };
    auto side_ihz = [&](DynPtr pz) { // W_ih[:, H:] @ phi_z + b_ih
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.nseg = 1;
        p.seg[0] = mkseg(pz, m->w_ih + (size_t)(H / 16) * 256, 2 * H / 16, H, 0);
        p.M = B; p.N = 3 * H; p.bias0 = m->b_ih;
        p.y = dp_static(w.part_i, 3 * H);
        K(BR_SIDE, p, EPI_LINEAR);
    };

    DynPtr pz_final = (kind == STEP_ENCODE) ? S(pz3, H) : dp_frame(DS_PZ, H, 0, 1);
    if (kind == STEP_ENCODE) {
        {   // enc.0([phi_x, h]) = (enc.0[:, :H] phi_x + b) [all frames beforehand: encode_prologue] + enc.0[:, H:] h, as in the persistent kernel
            GemmParams p = lin_params(m->enc[0], h_cur, B, S(e1, H));
            p.seg[0] = mkseg(h_cur, m->enc[0].wp + (size_t)(H / 16) * 256, 2 * H / 16, H, 0);
            p.bias0 = nullptr;
            p.aux = dp_frame(DS_PARTD, H);
            K(BR_MAIN, p, EPI_ELU);
        }
        K(BR_MAIN, lin_params(m->enc[1], S(e1, H), B, S(e2, H)), EPI_ELU);
        {
            GemmParams p = lin_params(m->enc[2], S(e2, H), B, dp_frame(DS_CODES, Z));
            p.var_bit = m->cfg.var_bit;
            p.aux = dp_frame(DS_BITS, 1);
            p.y3 = dp_frame(DS_PROB, Z);
            K(BR_MAIN, p, EPI_CODE);
        }
        K(BR_MAIN, lin_params(m->phi_z[0], dp_frame(DS_CODES, Z), B, S(pz1, H)), EPI_ELU);
        K(BR_MAIN, lin_params(m->phi_z[1], S(pz1, H), B, S(pz2, H)), EPI_ELU);
        K(BR_MAIN, lin_params(m->phi_z[2], S(pz2, H), B, S(pz3, H)), EPI_ELU);
        REC(BR_MAIN, EV_PZ);
    }
    const int n_dec0 = node;
    if (side) {      // dec.0 on the critical path only sees phi_z; the h half arrives from the side branch
        GemmParams p = lin_params(m->dec[0], pz_final, B, S(d1, H));
        p.seg[0] = mkseg(pz_final, m->dec[0].wp, 2 * H / 16, H, 0);
        p.aux = dp_static(w.part_d, H);
        WAIT(BR_MAIN, EV_DEC0H);
        K(BR_MAIN, p, EPI_ELU);
    } else if (kind == STEP_DECODE_PRE) {
        // dec.0([phi_z, h]) = (dec.0[:, :H] phi_z + b) [all frames, batched] + dec.0[:, H:] h
        GemmParams p = lin_params(m->dec[0], h_cur, B, S(d1, H));
        p.seg[0] = mkseg(h_cur, m->dec[0].wp + (size_t)(H / 16) * 256, 2 * H / 16, H, 0);
        p.bias0 = nullptr;
        p.aux = dp_frame(DS_PARTD, H);
        K(BR_MAIN, p, EPI_ELU);
    } else {
        // both halves in the step: the h half first, then the phi_z half, chunk by chunk into the same accumulators (the order of the
        // persistent kernel, whose filler quanta have dec.0[:, H:] h summed before phi_z exists)
        GemmParams p = lin_params(m->dec[0], h_cur, B, S(d1, H));
        p.nseg = 2;
        p.seg[0] = mkseg(h_cur, m->dec[0].wp + (size_t)(H / 16) * 256, 2 * H / 16, H, 0);
        p.seg[1] = mkseg(pz_final, m->dec[0].wp, 2 * H / 16, H, 0);
        K(BR_MAIN, p, EPI_ELU);
    }
    (void)n_dec0;
    K(BR_MAIN, lin_params(m->dec[1], S(d1, H), B, S(d2, H)), EPI_ELU);
    if (fold) {
        // one launch less per frame: u = ELU(dec.4) (decode: also kept for all frames - dec.6(u), the decoder's output, is one batched
        // GEMM behind the recurrence), then phi_x.0(norm(dec.6(u))) as the one folded layer
        GemmParams p = lin_params(m->dec[2], S(d2, H), B, S(d3, H));
        p.y2 = dp_frame(kind != STEP_ENCODE ? DS_KEEP : DS_KEEP_ENC, H);       // (encode: a null slot unless the fused forward wants mel^)
        K(BR_MAIN, p, EPI_ELU);
        K(BR_MAIN, lin_params(m->px0_dec3, S(d3, H), B, S(g1, H)), EPI_ELU);
    } else {
        K(BR_MAIN, lin_params(m->dec[2], S(d2, H), B, S(d3, H)), EPI_ELU);
        GemmParams p = lin_params(m->dec[3], S(d3, H), B, dp_frame(DS_MEL, X));      // (encode: a null slot unless the fused forward wants mel^)
        p.y2 = S(dn, X); p.mean = m->mean_mel; p.stdv = m->std_mel;
        K(BR_MAIN, p, EPI_MEL);
        K(BR_MAIN, lin_params(m->phi_x[0], S(dn, X), B, S(g1, H)), EPI_ELU);
    }
    K(BR_MAIN, lin_params(m->phi_x[1], S(g1, H), B, S(g2, H)), EPI_ELU);
    K(BR_MAIN, lin_params(m->phi_x[2], S(g2, H), B, S(g3, H)), EPI_ELU);
    {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.M = B; p.N = H; p.gate_rows = H;
        p.y = h_next;
        p.y2 = (kind == STEP_ENCODE) ? dp_frame(DS_ALLH, H, 1) : dp_null();   // all_h[:, t+1] (bvrnn.py:205)
        p.aux = h_cur;
        if (side) {
            p.nseg = 1;
            p.seg[0] = mkseg(S(g3, H), m->w_ih, 2 * H / 16, H, 0);                    // W_ih[:, :H] @ phi_x_gen
            p.part_i = w.part_i; p.part_h = w.part_h; p.ldpart = 3LL * H;
            WAIT(BR_MAIN, EV_GATES);
            K(BR_MAIN, p, EPI_GRU_PART);
        } else if (kind == STEP_DECODE_PRE) {
            p.nseg = 2;                                                               // W_ih[:, H:] phi_z + b_ih comes in through y3
            p.gate_il = 1;
            p.seg[0] = mkseg(S(g3, H), m->w_ih_il, 2 * H / 16, H, 0);
            p.seg[1] = mkseg(h_cur, m->w_hh_il, H / 16, H, 1);
            p.bias0 = nullptr; p.bias1 = m->b_hh;
            p.y3 = dp_frame(DS_PARTG, 3 * H);
            K(BR_MAIN, p, EPI_GRU);
        } else {
            p.nseg = 3;
            p.gate_il = 1;
            // cat([phi_x_gen, phi_z]) bvrnn.py:206; the phi_z third first (its input exists first: the persistent kernel sums it ahead)
            p.seg[0] = mkseg(pz_final, m->w_ih_il + (size_t)(H / 16) * 3 * 256, 2 * H / 16, H, 0);
            p.seg[1] = mkseg(S(g3, H), m->w_ih_il, 2 * H / 16, H, 0);
            p.seg[2] = mkseg(h_cur, m->w_hh_il, H / 16, H, 1);
            p.bias0 = m->b_ih; p.bias1 = m->b_hh;
            K(BR_MAIN, p, EPI_GRU);
        }
    }
    if (side) {
        // side-branch operations, inserted at the positions where their inputs exist: step start for the
        // two h products; after phi_z for the W_ih half (encode) or step start (decode: phi_z is batched)
        std::vector<StepNode> main_ops;
        main_ops.swap(plan);
        REC(BR_MAIN, EV_START);
        WAIT(BR_SIDE, EV_START);
        side_dec0h();
        REC(BR_SIDE, EV_DEC0H);
        side_hh();
        if (kind == STEP_DECODE) { side_ihz(pz_final); REC(BR_SIDE, EV_GATES); }
        for (const StepNode &n : main_ops) {
            plan.push_back(n);
            if (n.op == OP_RECORD && n.event == EV_PZ) {        // encode: phi_z ready
                WAIT(BR_SIDE, EV_PZ);
                side_ihz(pz_final);
                REC(BR_SIDE, EV_GATES);
            }
        }
    }
    return plan;
}

int count_kernels(const std::vector<StepNode> &plan) {
    int n = 0;
    for (const StepNode &s : plan) n += (s.op == OP_KERNEL);
    return n;
}

// Launch `nsteps` consecutive frames; the frame counter is advanced ONCE at the end: frame k of the
// group runs with the static offset tstep = k baked into its kernel arguments.  With side == nullptr
// everything runs in plan order on `s` (the plan order respects every dependency).
int launch_steps(const bvc_model *m, const std::vector<StepNode> &plan, const Workspace &w, int nsteps, hipStream_t s,
                 hipStream_t side) {
    int rc;
    for (int k = 0; k < nsteps; ++k)
        for (const StepNode &n : plan) {
            hipStream_t st = (n.branch == BR_SIDE && side) ? side : s;
            if (n.op == OP_KERNEL) {
                GemmParams p = n.p;
                p.tstep = k;
                if ((rc = launch_gemm_skinny(p, n.epi, st, m->mtw))) return rc;
            } else if (side) {
                if (n.op == OP_RECORD) BVC_HIP_TRY(hipEventRecord(m->cap_events[n.event], st));
                else                   BVC_HIP_TRY(hipStreamWaitEvent(st, m->cap_events[n.event], 0));
            }
        }
    return launch_step_advance(w.desc, nsteps, s);
}

constexpr int GRAPH_STEPS = 8;

constexpr size_t GRAPH_CACHE_ENTRIES = 64;      // (kind, batch, workspace) triples kept per model; least recently used goes first

// Returns the cached graph pair for (kind, B, workspace), capturing it on first use.  Caller holds m->graph_mu.
int get_step_graph(const bvc_model *m, const Workspace &w, void *ws_base, int B, int kind,
                   const std::vector<StepNode> &plan, bvc_model::StepGraph **out) {
    void *probe = g_kprobe.enabled ? (void *)g_kprobe.dev : nullptr;
    for (auto it = m->graphs.begin(); it != m->graphs.end(); ++it)
        if (it->kind == kind && it->B == B && it->ws == ws_base && it->probe == probe) {
            m->graphs.splice(m->graphs.begin(), m->graphs, it);          // most recently used first
            *out = &m->graphs.front();
            return BVC_OK;
        }
    if (!m->cap_stream) BVC_HIP_TRY(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    if (!m->side_stream) BVC_HIP_TRY(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
    while ((int)m->cap_events.size() < EV_COUNT) {
        hipEvent_t e;
        BVC_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->cap_events.push_back(e);
    }
    bvc_model::StepGraph sg{kind, B, ws_base, probe, nullptr, nullptr, nullptr};
    for (int which = 0; which < 2; ++which) {
        hipGraph_t graph = nullptr;
        g_capturing = true;
        hipError_t e = hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal);
        int rc = BVC_OK;
        if (e == hipSuccess) rc = launch_steps(m, plan, w, which ? GRAPH_STEPS : 1, m->cap_stream, m->side_branch ? m->side_stream : nullptr);
        // always close the capture, also when a launch inside it failed, so the stream stays usable
        hipError_t e2 = (e == hipSuccess) ? hipStreamEndCapture(m->cap_stream, &graph) : e;
        g_capturing = false;
        if (rc) { if (graph) (void)hipGraphDestroy(graph); if (sg.exec1) (void)hipGraphExecDestroy(sg.exec1); return rc; }
        if (e2 != hipSuccess || !graph) {
            if (sg.exec1) (void)hipGraphExecDestroy(sg.exec1);
            set_error("hipGraph capture failed: %s", hipGetErrorString(e2));
            return BVC_EHIP;
        }
        hipGraphExec_t ex = nullptr;
        BVC_HIP_TRY(hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0));
        BVC_HIP_TRY(hipGraphDestroy(graph));
        (which ? sg.execN : sg.exec1) = ex;
    }
    BVC_HIP_TRY(hipEventCreateWithFlags(&sg.idle, hipEventDisableTiming));
    while (m->graphs.size() >= GRAPH_CACHE_ENTRIES) {       // bound the cache: the least recently used entry goes, once it is idle
        bvc_model::StepGraph &old = m->graphs.back();
        (void)hipEventSynchronize(old.idle);                  // (never recorded: returns at once)
        (void)hipGraphExecDestroy(old.exec1); (void)hipGraphExecDestroy(old.execN); (void)hipEventDestroy(old.idle);
        m->graphs.pop_back();
    }
    m->graphs.push_front(sg);
    *out = &m->graphs.front();
    return BVC_OK;
}

// Does the launch-per-layer schedule of this call fold the hop?  Whenever the model does (`encode_fold` / `decode_fold`), streaming hops
// included: every schedule runs the same layer list, so that their results are the same bits.
int step_fold(const bvc_model *m, bool encode, int64_t T) {
    if (!m->px0_dec3.wp || m->side_branch) return 0;
    (void)T;
    return (encode ? m->encode_fold : m->decode_fold) ? STEP_FOLD : 0;
}

int run_recurrence(const bvc_model *m, const Workspace &w, void *ws_base, int B, int64_t T, int kind, hipStream_t s) {
    const std::vector<StepNode> plan = build_step(m, w, B, kind);
    // (begin_call was given count_kernels(build_step(...)) kernels per step)
    int rc;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    // a caller's capture takes the kernels directly (no graph of our own is built, replayed or marked idle inside it)
    if (!m->use_graph || g_stream_tick || cs != hipStreamCaptureStatusNone) {
        int rc2;
        for (int64_t t = 0; t < T; ++t)
            if ((rc2 = launch_steps(m, plan, w, 1, s, nullptr))) return rc2;
        return BVC_OK;
    }
    std::lock_guard<std::mutex> lk(m->graph_mu);          // cache look-up, replay and the idle mark are one critical section
    bvc_model::StepGraph *g = nullptr;
    if ((rc = get_step_graph(m, w, ws_base, B, kind, plan, &g))) {
        if (rc != BVC_EHIP) return rc;
        // stream capture unavailable (e.g. the caller is itself capturing): same kernels, launched eagerly
        (void)hipGetLastError();
        for (int64_t t = 0; t < T; ++t)
            if ((rc = launch_steps(m, plan, w, 1, s, nullptr))) return rc;
        return BVC_OK;
    }
    int64_t t = 0;
    for (; t + GRAPH_STEPS <= T; t += GRAPH_STEPS) BVC_HIP_TRY(hipGraphLaunch(g->execN, s));
    for (; t < T; ++t) BVC_HIP_TRY(hipGraphLaunch(g->exec1, s));
    BVC_HIP_TRY(hipEventRecord(g->idle, s));
    return BVC_OK;
}

int begin_call(const bvc_model *m, const Workspace &w, const CallDesc &v, int steps_nodes, hipStream_t s) {
    CallDesc d = v;
    d.t = 0;
    d.nodes_per_step = steps_nodes;
    if (g_kprobe.enabled) {
        const size_t need = (size_t)2 * d.T * steps_nodes;
        if (need > g_kprobe.capacity) { set_error("kprobe buffer too small for T=%lld", (long long)d.T); return BVC_EINVAL; }
        {
            // first half: start stamps (atomicMin, so all ones); second half: end stamps (atomicMax, so zero)
            BVC_HIP_TRY(hipMemsetAsync(g_kprobe.dev, 0xFF, need / 2 * sizeof(unsigned long long), s));
            BVC_HIP_TRY(hipMemsetAsync(g_kprobe.dev + need / 2, 0, need / 2 * sizeof(unsigned long long), s));
            g_kprobe.T = d.T; g_kprobe.nodes = steps_nodes;
        }
    }
    return launch_set_desc(w.desc, d, s);
}

// GRU state lives fragment-packed in hbuf[parity]; h0 goes to parity 0
int init_state(const Workspace &w, const float *d_h0, int B, int H, hipStream_t s) {
    if (d_h0) return launch_repack_rows(d_h0, w.hbuf, H, B, H, 0, s);
    return launch_fill(w.hbuf, 0.0f, (long long)((B + 15) / 16) * 16 * H, s);
}

int read_state(const Workspace &w, int B, int H, int64_t T, float *d_hT, hipStream_t s) {
    const float *src = w.hbuf + (T & 1) * (long long)((B + 15) / 16) * 16 * H;
    return launch_repack_rows(src, d_hT, H, B, H, 1, s);
}

// ---- persistent recurrence (k_flow.hip): hop tables and launch -----------------------------------------
// Is the model laid out for the persistent kernel?  (h_dim a multiple of 128 up to 1024 or below 128; narrow z / mel layers)
int flow_census(const bvc_model *m);

int build_flow(bvc_model *m) {
    const int H = m->cfg.h_dim, Z = m->cfg.z_dim, X = m->cfg.num_mels;
    m->flow_perh = flow_perh(H);
    if (Z > 128 || X > 128) m->flow_perh = 0;
    if (!m->flow_perh) return BVC_OK;
    void *st = nullptr, *dst = nullptr;
    BVC_HIP_TRY(hipHostMalloc(&st, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(st, 0, 64);
    m->h_status = static_cast<volatile unsigned *>(st);
    BVC_HIP_TRY(hipHostGetDevicePointer(&dst, st, 0));
    m->d_status = static_cast<unsigned *>(dst);
    int dev = 0;
    BVC_HIP_TRY(hipGetDevice(&dev));
    BVC_HIP_TRY(hipDeviceGetAttribute(&m->cu_count, hipDeviceAttributeMultiprocessorCount, dev));
    int rc = flow_kernels_init();
    if (rc) return rc;
    return flow_census(m);
}

// Reads the sticky status word (no synchronisation): the first call after a persistent kernel gave up reports it.
int sticky_status(const bvc_model *m) {
    if (!m || !m->h_status) return BVC_OK;
    const unsigned v = *m->h_status;
    if (!v) return BVC_OK;
    *m->h_status = 0u;
    m->census_due = true;
    set_error("a persistent recurrence kernel of an earlier call gave up waiting (frame %u, layer %u): the results of that call "
              "are invalid.  All its workgroups must be resident together - is another process using this GPU?",
              (v & 0x7FFFFFFFu) >> 4, (v & 15u));
    return BVC_ETIMEOUT;
}

inline FlowLin flin(const Linear &l, size_t kb_offset = 0, bool with_bias = true) {
    FlowLin f;
    f.w = l.wp + kb_offset * 256; f.bias = with_bias ? l.b : nullptr; f.wnb = l.in / 16; f.pad_ = 0;
    return f;
}

// The layers of one frame (encode: bvrnn.py:187-206, decode: bvrnn.py:222-227).  Halves of a concatenated input that do
// not depend on the frame's own chain - phi_x(y_t) in enc.0, phi_z(z_t) in dec.0 and in the GRU's input gates when the
// codes are known - are batched over all frames beforehand and enter as addends (part0 / part_gru).
void flow_layers(const bvc_model *m, bool encode, FlowArgs *a) {
    const int hb = m->cfg.h_dim / 16;
    a->enc0h = flin(m->enc[0], hb, false);            // enc.0[:, H:] h  (+ part0 = enc.0[:, :H] phi_x + b)
    a->enc1 = flin(m->enc[1]);
    a->enc2 = flin(m->enc[2]);
    a->pz0 = flin(m->phi_z[0]);
    a->pz1 = flin(m->phi_z[1]);
    a->pz2 = flin(m->phi_z[2]);
    a->dec0h = flin(m->dec[0], hb, encode);           // dec.0[:, H:] h; decode: + part0 = dec.0[:, :H] phi_z + b
    a->dec0z = flin(m->dec[0], 0, false);             // dec.0[:, :H] phi_z (encode)
    a->dec1 = flin(m->dec[1]);
    a->dec2 = flin(m->dec[2]);
    a->dec3 = flin(m->dec[3]);
    a->px0 = flin(m->phi_x[0]);
    a->px1 = flin(m->phi_x[1]);
    a->px2 = flin(m->phi_x[2]);
    if ((encode ? m->encode_fold : m->decode_fold) && m->px0_dec3.wp) a->pxc = flin(m->px0_dec3);
    a->w_hh = m->w_hh_il;
    a->w_ihx = m->w_ih_il;
    a->w_ihz = m->w_ih_il + (size_t)hb * 3 * 256;
    a->b_ih = m->b_ih; a->b_hh = m->b_hh;
    a->hb = hb; a->zb = m->cfg.z_dim / 16; a->xb = m->cfg.num_mels / 16;
}

// The persistent kernel needs every one of its workgroups resident (they wait for each other) and a workgroup takes a whole
// compute unit (8 waves x 256 VGPRs): utterance groups x feature tiles must not exceed the device's CU count (256 on MI355X:
// up to 64 utterances at h_dim 1024).  Anything else takes the launch-per-layer schedule.
// Larger batches interleave MG utterance groups ("chains") per workgroup (h_dim 1024 only; k_flow.hip, MULTI).
constexpr int FLOW_MAX_CHAINS = 8;
enum { RS_PERSISTENT = 0, RS_LAYERS = 1, RS_AUTO = 2 };
inline int flow_grid_tiles(const bvc_model *m) {          // feature tiles covered by a persistent grid (rounded up to 8: one per XCD)
    const int H = m->cfg.h_dim, X = m->cfg.num_mels, Z = m->cfg.z_dim;
    return ((H > X ? (H > Z ? H : Z) : (X > Z ? X : Z)) / 16 + 7) / 8 * 8;
}
inline int flow_chains_static(const bvc_model *m, int B) {      // 0: not usable; else utterance groups per workgroup
    if (m->recurrence == RS_LAYERS || m->side_branch || !m->flow_resident || (g_stream_tick && !g_tick_flow) || m->flow_perh <= 0) return 0;
    const int ntg = flow_grid_tiles(m);
    const int mt = (B + 15) / 16;
    const int slots = m->cu_count / ntg;                   // workgroups per feature tile that fit on the device
    if (slots <= 0) return 0;
    const int mg = (mt + slots - 1) / slots;
    if (mg <= 1) return 1;
    static const bool no_multi = getenv("BVC_FLOW_NO_CHAINS") != nullptr;
    if (no_multi || m->flow_perh != 8 || mg > FLOW_MAX_CHAINS) return 0;
    return mg;
}

// The persistent kernel needs all its workgroups resident at once (they wait for each other), so launches from
// different streams are serialised through one event: at most one is in flight per process and device.
std::mutex g_flow_mu;
constexpr int FLOW_MAX_TICKETS = 2;
hipEvent_t g_flow_ev[16][FLOW_MAX_TICKETS] = {};
unsigned long long g_flow_n[16] = {};
// RS_AUTO: the end of the last recurrence-bearing call on this device (any model of this process) and the stream it ran on.
// A call that starts while the previous one - issued on ANOTHER stream - is still running has company: batches are in flight
// on several streams, where the launch-per-layer chains of the streams interleave on the chip while persistent launches (each takes
// every compute unit) would run one after the other.  The switch is sticky for AUTO_HOLD calls so that all streams change together.
struct LastCall { hipEvent_t ev = nullptr; hipStream_t s = nullptr; bool any = false; int hold = 0; };
LastCall g_last_call[16];
constexpr int AUTO_HOLD = 2;

// Which schedule does THIS call take?  0: launch per layer; else utterance groups per workgroup of the persistent kernel.
// Called once per recurrence-bearing call (run_encode / run_decode); mark_call_end() follows at its end.
int flow_chains(const bvc_model *m, int B, hipStream_t s) {
    if (m->census_due && (!g_stream_tick || g_tick_flow)) {               // a time-out was reported: is a full grid still co-resident?  (synchronises: error path only)
        hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs0) == hipSuccess && cs0 == hipStreamCaptureStatusNone) {
            m->census_due = false;
            (void)flow_census(m);
        } else {
            (void)hipGetLastError();
        }
    }
    const int chains = flow_chains_static(m, B);
    if (!chains) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
    // a caller's capture: the persistent launch cannot be captured (its one-at-a-time ticket is a host-side wait on an event
    // recorded outside the capture, and replays would skip it): captured calls take the launch-per-layer kernels
    if (cs != hipStreamCaptureStatusNone) return 0;
    if (m->recurrence != RS_AUTO) return chains;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return chains; }
    std::lock_guard<std::mutex> lk(g_flow_mu);
    LastCall &lc = g_last_call[dev & 15];
    const bool company = lc.any && lc.s != s && hipEventQuery(lc.ev) == hipErrorNotReady;
    (void)hipGetLastError();
    if (company) lc.hold = AUTO_HOLD;
    else if (lc.hold > 0) --lc.hold;
    return (company || lc.hold > 0) ? 0 : chains;
}

// Fences (bvc_flow_fence): work a caller issued on some stream - an RCCL collective that holds compute units while it waits for
// its peers, say - that must have finished before the next persistent launch starts.  A small ring; a persistent launch
// waits for every pending entry (later launches wait for that launch through the ticket).
constexpr int FLOW_FENCES = 8;
struct FlowFence { hipEvent_t ev = nullptr; bool pending = false; };
FlowFence g_fence[16][FLOW_FENCES];
unsigned g_fence_n[16] = {};

int mark_call_end(hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return BVC_OK; }
    int dev = 0;
    BVC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_flow_mu);
    LastCall &lc = g_last_call[dev & 15];
    if (!lc.ev) BVC_HIP_TRY(hipEventCreateWithFlags(&lc.ev, hipEventDisableTiming));
    BVC_HIP_TRY(hipEventRecord(lc.ev, s));
    lc.s = s; lc.any = true;
    return BVC_OK;
}

inline float *flow_buf(const Workspace &w, int id, int parity) { return w.flow + (size_t)(id * 2 + parity) * w.flow_slot; }

// One-off at model creation: can a full persistent grid (one workgroup per compute unit the device reports) be resident at
// once?  flow_census_kernel has the recurrence kernels' footprint - 512 threads, every VGPR, the filler kernels' LDS -: every
// workgroup adds itself to a counter and waits (bounded) until all have.  A CU mask, a partition mode or another tenant of the
// device that keeps workgroups from becoming co-resident shows up here; the model then stays on the launch-per-layer schedule
// (flow_resident = false).
int flow_census(const bvc_model *m) {
    m->flow_resident = false;
    const int ntg = flow_grid_tiles(m);
    int slots = m->cu_count / ntg;
    if (slots <= 0) return BVC_OK;
    std::lock_guard<std::mutex> lk(g_flow_mu);           // (no persistent launch of this process starts beside the census)
    if (!m->census_ctr) BVC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->census_ctr), 64));
    unsigned *ctr = m->census_ctr;
    // a stream of the library's own, synchronised on its own: neither the null stream nor hipDeviceSynchronize() is legal while
    // another thread captures a graph
    if (!m->census_stream) BVC_HIP_TRY(hipStreamCreateWithFlags(&m->census_stream, hipStreamNonBlocking));
    if (getenv("BVC_FLOW_CENSUS_OVERSUBSCRIBE")) slots += 1;             // tests: a grid the device cannot hold
    const int grid = ntg * slots;
    unsigned h[2] = {0u, 0u};
    for (int attempt = 0; attempt < 2; ++attempt) {
        BVC_HIP_TRY(hipMemsetAsync(ctr, 0, 64, m->census_stream));
        // ~50 ms: a workgroup that has to queue behind a resident one shows up as a time-out
        const int rc = launch_flow_census(ctr, grid, 200000u, m->census_stream);
        hipError_t e = hipStreamSynchronize(m->census_stream);
        if (e == hipSuccess) e = hipMemcpy(h, ctr, sizeof(h), hipMemcpyDeviceToHost);
        if (rc) return rc;
        BVC_HIP_TRY(e);
        m->flow_resident = h[0] == (unsigned)grid && h[1] == 0u;
        if (m->flow_resident || attempt == 1) break;
        // "device busy" is not "grid does not fit": work of this process on other streams (another model serving, say) holds
        // compute units for a while - let it drain and count once more before giving the persistent schedule up for good
        if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); break; }       // (illegal under a capture: keep the first answer)
    }
    if (!m->flow_resident && !getenv("BVC_QUIET"))
        fprintf(stderr, "bvcodec: residency census: %u of %d recurrence workgroups became co-resident (%u gave up) - this model stays on the "
                        "launch-per-layer schedule (get_option \"flow_resident\" = 0).  Is another process using this GPU?\n", h[0] - h[1], grid, h[1]);
    return BVC_OK;
}

int decode_epilogue(const bvc_model *m, const float *keep, int B, int64_t T, float *d_mel, hipStream_t s);

// All T frames of BVRNN.encode (encode = true) or BVRNN.decode in one launch.  w.part_dec0 (and w.part_gru for decode)
// must hold the pre-computed halves; h0 may be null (zero state).
int run_flow(const bvc_model *m, const Workspace &w, bool encode, int chains, const float *d_h0, int B, int64_t T, const float *d_bits,
             float *d_codes, float *d_prob, float *d_all_h, float *d_mel, float *d_hT, hipStream_t s) {
    const int H = m->cfg.h_dim, Z = m->cfg.z_dim, X = m->cfg.num_mels;
    const int mt16 = ((B + 15) / 16) * 16;
    int rc;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        BVC_HIP_TRY(hipStreamIsCapturing(s, &cs));
        if (cs != hipStreamCaptureStatusNone) { set_error("the persistent recurrence cannot be captured into a graph"); return BVC_EINVAL; }
    }
    float *h0p = flow_buf(w, FB_H, 0);
    FlowArgs a;
    memset(&a, 0, sizeof(a));
    flow_layers(m, encode, &a);
    a.flow = w.flow;
    a.slot_bytes = (unsigned)(w.flow_slot * sizeof(float));
    a.B = B; a.MT = mt16 / 16; a.T = T;
    a.MG = chains;
    a.NTG = (H > X ? (H > Z ? H : Z) : (X > Z ? X : Z)) / 16;
    a.part0 = w.part_dec0;
    a.part_gru = encode ? nullptr : w.part_gru;
    a.codes = d_codes; a.prob = d_prob; a.bits = d_bits; a.all_h = d_all_h; a.mel = d_mel;
    a.keep = (encode && !d_mel) ? nullptr : w.pxB;   // folded hop: ELU(dec.4) of all frames (pxB is idle once the batched phi_x / phi_z layers are through);
                                                     // encode keeps it only when the caller wants the decoder's output too (bvc_forward)
    a.mean = m->mean_mel; a.stdv = m->std_mel;
    a.var_bit = m->cfg.var_bit;
    a.status = m->d_status;
    { static const bool hot = getenv("BVC_FLOW_HOTW") != nullptr; a.dbg_hot_w = hot ? 1 : 0; }
    a.spin_limit = m->flow_spin_limit;
    a.dbg_withhold = m->flow_debug_withhold;
    if (g_kprobe.enabled) {                    // bench instrumentation: per-layer entry / exit stamps of workgroup 0
        const int nodes = encode ? 14 : 8;
        const size_t need = (size_t)FLOW_STAMPS * T * nodes;
        if (need > g_kprobe.capacity) { set_error("kprobe buffer too small for T=%lld", (long long)T); return BVC_EINVAL; }
        BVC_HIP_TRY(hipMemsetAsync(g_kprobe.dev, 0, need * sizeof(unsigned long long), s));
        g_kprobe.T = T; g_kprobe.nodes = nodes;
        a.probe = g_kprobe.dev; a.probe_nodes = nodes; a.probe_first = encode ? 1 : 7;
        a.probe_wg = getenv("BVC_PROBE_WG") ? atoi(getenv("BVC_PROBE_WG")) : 0;
        a.probe_wave = getenv("BVC_PROBE_WAVE") ? atoi(getenv("BVC_PROBE_WAVE")) & 7 : 0;
    }
    // the flow region filled with the sentinel, h(-1) in its first buffer (FB_H, parity 0, at the start of the region) and the device copy
    // of the arguments: one kernel (a misaligned initial state takes the three separate ones)
    const long long n_flow = (long long)FB_COUNT * 2 * (long long)w.flow_slot;
    const bool fused_prepare = !d_h0 || (reinterpret_cast<uintptr_t>(d_h0) & 15) == 0;
    if (fused_prepare) {
        if ((rc = launch_flow_prepare(a, w.flow_args, reinterpret_cast<unsigned *>(w.flow), n_flow, (long long)mt16 * H, d_h0, B, H, s))) return rc;
    } else {
        if ((rc = launch_fill_u32(reinterpret_cast<unsigned *>(w.flow), FLOW_POISON, n_flow, s))) return rc;
        if ((rc = launch_fill(h0p, 0.0f, (long long)mt16 * H, s))) return rc;
        if ((rc = launch_repack_rows(d_h0, h0p, H, B, H, 0, s))) return rc;
    }
    if (d_all_h && (rc = launch_repack_rows(h0p, d_all_h, (long long)T * H, B, H, 1, s))) return rc;     // all_h[:, 0] = h0
    {
        std::lock_guard<std::mutex> lk(g_flow_mu);
        int dev = 0;
        BVC_HIP_TRY(hipGetDevice(&dev));
        dev &= 15;
        static const int tickets = (getenv("BVC_FLOW_TICKETS") && atoi(getenv("BVC_FLOW_TICKETS")) == 2) ? 2 : 1;
        hipEvent_t &ev = g_flow_ev[dev][g_flow_n[dev] % tickets];        // the launch `tickets` launches ago must have finished
        if (!ev) BVC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (g_flow_n[dev] >= (unsigned long long)tickets) BVC_HIP_TRY(hipStreamWaitEvent(s, ev, 0));
        for (auto &f : g_fence[dev])
            if (f.pending) { BVC_HIP_TRY(hipStreamWaitEvent(s, f.ev, 0)); f.pending = false; }
        ProbeScope probe(PK_LINEAR, s);
        static const bool fill = !(getenv("BVC_FLOW_FILL") && getenv("BVC_FLOW_FILL")[0] == '0');
        if ((rc = launch_flow(a, w.flow_args, m->flow_perh, encode, fill && !m->flow_debug_nofill && a.MG == 1, s, fused_prepare))) return rc;
        BVC_HIP_TRY(hipEventRecord(ev, s));
        ++g_flow_n[dev];
    }
    if (d_hT && (rc = launch_repack_rows(flow_buf(w, FB_H, (int)(T & 1)), d_hT, H, B, H, 1, s))) return rc;
    // folded decode: the decoder's output dec.6(u_t) for all frames at once (bvrnn.py:224-225)
    if (a.pxc.w && d_mel && (rc = decode_epilogue(m, w.pxB, B, T, d_mel, s))) return rc;
    return BVC_OK;
}

// Three-layer ELU MLP over ALL frames (phi_x at bvrnn.py:178, phi_z at bvrnn.py:223): in (B*T rows, utterance-major)
// -> pxA, one fragment-packed [mt16][H] matrix per frame.  (Re-ordering the rows frame-major in the first layer, so
// that the last one writes whole 1 KiB blocks - GO_FRAME_MAJOR_ROWS / GO_PACKED_FRAMES - measured 0.3 ms per step
// SLOWER: the first layer's row scatter costs more than the last layer's 16-byte granules.)
int batched_mlp3(const bvc_model *m, const Workspace &w, const Linear (&l)[3], const float *in, int K0, int B, int64_t T,
                 hipStream_t s) {
    const int H = m->cfg.h_dim;
    const int mt16 = ((B + 15) / 16) * 16;
    const int BT = (int)((long long)B * T);
    int rc;
    if (T <= SMALL_T_FRAMES) {
        // a streaming hop (1-2 frames): B*T rows fill a handful of the batched kernel's 128x128 tiles (102 us per 1024^2
        // layer at 256 streams); the recurrent-layer kernel takes the rows of ONE frame (row stride T*K0) in 10 us and
        // writes the fragment-packed frame matrix directly
        // (all T frames in one launch per layer: the frame is the grid's second dimension)
        const long long FS = (long long)mt16 * H;
        auto layer = [&](const Linear &ln, DynPtr x, DynPtr y) {
            GemmParams p = lin_params(ln, x, B, y);
            p.frames = (int)T;
            return launch_gemm_skinny(p, EPI_ELU, s, m->mtw);
        };
        if ((rc = layer(l[0], dp_static_frames(in, T * K0, K0), dp_static_frames(w.pxC, H, FS, 1)))) return rc;
        if ((rc = layer(l[1], dp_static_frames(w.pxC, H, FS, 1), dp_static_frames(w.pxB, H, FS, 1)))) return rc;
        return layer(l[2], dp_static_frames(w.pxB, H, FS, 1), dp_static_frames(w.pxA, H, FS, 1));
    }
    if ((rc = launch_gemm_batched(in, K0, l[0].w, K0, l[0].b, BT, H, K0, 1, w.pxC, H, s))) return rc;
    if ((rc = launch_gemm_batched(w.pxC, H, l[1].w, H, l[1].b, BT, H, H, 1, w.pxB, H, s))) return rc;
    return launch_gemm_batched(w.pxB, H, l[2].w, H, l[2].b, BT, H, H, 1, w.pxA, H, s, GO_PACKED_FROM_UTT, T, mt16);
}

// Everything of BVRNN.encode that does not depend on the recurrence, for all frames: phi_x(yn) (bvrnn.py:178) and the phi_x half of
// enc.0 with its bias (bvrnn.py:189) -> w.part_dec0, natural (B,T,H).  The batched GEMMs for a whole utterance; on a streaming hop
// (1-2 frames: B*T rows fill a handful of their 128x128 tiles, 102 us per 1024^2 layer at 256 streams) the recurrent-layer kernel
// frame by frame (10 us).  Same bits either way (one order of summation: k_gemm.hip).
int encode_prologue(const bvc_model *m, const Workspace &w, int B, int64_t T, hipStream_t s) {
    const int H = m->cfg.h_dim, X = m->cfg.num_mels;
    const int mt16 = ((B + 15) / 16) * 16;
    int rc;
    if (T <= SMALL_T_FRAMES) {                             // all T frames in one launch per layer: the frame is the grid's second dimension
        const long long FS = (long long)mt16 * H;
        auto layer = [&](GemmParams p, int epi) { p.frames = (int)T; return launch_gemm_skinny(p, epi, s, m->mtw); };
        if ((rc = layer(lin_params(m->phi_x[0], dp_static_frames(w.yn, T * X, X), B, dp_static_frames(w.pxC, H, FS, 1)), EPI_ELU))) return rc;
        if ((rc = layer(lin_params(m->phi_x[1], dp_static_frames(w.pxC, H, FS, 1), B, dp_static_frames(w.pxB, H, FS, 1)), EPI_ELU))) return rc;
        if ((rc = layer(lin_params(m->phi_x[2], dp_static_frames(w.pxB, H, FS, 1), B, dp_static_frames(w.pxA, H, FS, 1)), EPI_ELU))) return rc;
        GemmParams p = lin_params(m->enc[0], dp_static_frames(w.pxA, H, FS, 1), B, dp_static_frames(w.part_dec0, T * H, H));
        p.seg[0] = mkseg(dp_static_frames(w.pxA, H, FS, 1), m->enc[0].wp, 2 * H / 16, H, 0);
        finish(p);
        return layer(p, EPI_LINEAR);
    }
    const int BT = (int)((long long)B * T);
    if ((rc = launch_gemm_batched(w.yn, X, m->phi_x[0].w, X, m->phi_x[0].b, BT, H, X, 1, w.pxC, H, s))) return rc;
    if ((rc = launch_gemm_batched(w.pxC, H, m->phi_x[1].w, H, m->phi_x[1].b, BT, H, H, 1, w.pxB, H, s))) return rc;
    if ((rc = launch_gemm_batched(w.pxB, H, m->phi_x[2].w, H, m->phi_x[2].b, BT, H, H, 1, w.pxC, H, s))) return rc;
    return launch_gemm_batched(w.pxC, H, m->enc[0].w, 2 * H, m->enc[0].b, BT, H, H, 0, w.part_dec0, H, s);
}

// ... and of BVRNN.decode (the codes are known): phi_z(z) (bvrnn.py:223), the phi_z half of dec.0 with its bias (bvrnn.py:224) ->
// w.part_dec0 (B,T,H), and the phi_z half of the GRU's input gates with b_ih (bvrnn.py:227) -> w.part_gru (B,T,3H).
int decode_prologue(const bvc_model *m, const Workspace &w, const float *d_codes, int B, int64_t T, hipStream_t s) {
    const int H = m->cfg.h_dim, Z = m->cfg.z_dim;
    const int mt16 = ((B + 15) / 16) * 16;
    int rc;
    if (T <= SMALL_T_FRAMES) {                             // all T frames in one launch per layer: the frame is the grid's second dimension
        const long long FS = (long long)mt16 * H;
        auto layer = [&](GemmParams p, int epi) { p.frames = (int)T; return launch_gemm_skinny(p, epi, s, m->mtw); };
        if ((rc = layer(lin_params(m->phi_z[0], dp_static_frames(d_codes, T * Z, Z), B, dp_static_frames(w.pxC, H, FS, 1)), EPI_ELU))) return rc;
        if ((rc = layer(lin_params(m->phi_z[1], dp_static_frames(w.pxC, H, FS, 1), B, dp_static_frames(w.pxB, H, FS, 1)), EPI_ELU))) return rc;
        if ((rc = layer(lin_params(m->phi_z[2], dp_static_frames(w.pxB, H, FS, 1), B, dp_static_frames(w.pxA, H, FS, 1)), EPI_ELU))) return rc;
        GemmParams p = lin_params(m->dec[0], dp_static_frames(w.pxA, H, FS, 1), B, dp_static_frames(w.part_dec0, T * H, H));
        p.seg[0] = mkseg(dp_static_frames(w.pxA, H, FS, 1), m->dec[0].wp, 2 * H / 16, H, 0);
        finish(p);
        if ((rc = layer(p, EPI_LINEAR))) return rc;
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.nseg = 1;
        q.seg[0] = mkseg(dp_static_frames(w.pxA, H, FS, 1), m->w_ih + (size_t)(H / 16) * 256, 2 * H / 16, H, 0);
        q.M = B; q.N = 3 * H; q.bias0 = m->b_ih;
        q.y = dp_static_frames(w.part_gru, T * 3 * H, 3 * H);
        finish(q);
        return layer(q, EPI_LINEAR);
    }
    const int BT = (int)((long long)B * T);
    if ((rc = launch_gemm_batched(d_codes, Z, m->phi_z[0].w, Z, m->phi_z[0].b, BT, H, Z, 1, w.pxC, H, s))) return rc;
    if ((rc = launch_gemm_batched(w.pxC, H, m->phi_z[1].w, H, m->phi_z[1].b, BT, H, H, 1, w.pxB, H, s))) return rc;
    if ((rc = launch_gemm_batched(w.pxB, H, m->phi_z[2].w, H, m->phi_z[2].b, BT, H, H, 1, w.pxC, H, s))) return rc;
    if ((rc = launch_gemm_batched(w.pxC, H, m->dec[0].w, 2 * H, m->dec[0].b, BT, H, H, 0, w.part_dec0, H, s))) return rc;
    return launch_gemm_batched(w.pxC, H, m->w_ih_nat + H, 2 * H, m->b_ih, BT, 3 * H, H, 0, w.part_gru, 3 * H, s);
}

// dec.6 over the kept u = ELU(dec.4) of all frames (the folded hop): the decoder's output mel^ (bvrnn.py:224-225)
int decode_epilogue(const bvc_model *m, const float *keep, int B, int64_t T, float *d_mel, hipStream_t s) {
    const int H = m->cfg.h_dim, X = m->cfg.num_mels;
    const int BT = (int)((long long)B * T);
    if (T <= SMALL_T_FRAMES)
        return launch_gemm_skinny(lin_params(m->dec[3], dp_static(keep, H), BT, dp_static(d_mel, X)), EPI_LINEAR, s, m->mtw);
    return launch_gemm_batched(keep, H, m->dec[3].w, H, m->dec[3].b, BT, X, H, 0, d_mel, X, s);
}

// d_melhat (optional): the decoder's output dec(phi_z(z_t), h_t) of every frame (B,T,num_mels) - what BVRNN.decode(codes) would compute over again
// from the same trajectory (bvrnn.py:202 vs :224-225; the fused forward, bvc_forward)
int run_encode_body(const bvc_model *m, const Workspace &w, void *ws_base, const float *d_mel, const float *d_bits,
                    const float *d_h0, int B, int64_t T, float *d_codes, float *d_all_h, float *d_hT, float *d_prob,
                    float *d_melhat, hipStream_t s) {
    const int H = m->cfg.h_dim, X = m->cfg.num_mels;
    const long long BT = (long long)B * T;
    int rc;
    if (m->cfg.var_bit && !d_bits) { set_error("bits per frame required when var_bit=1"); return BVC_EINVAL; }
    // y = (y - mean) / std ; phi_x over all frames (bvrnn.py:173-178)
    if ((rc = launch_normalize_rows(d_mel, m->mean_mel, m->std_mel, BT, X, w.yn, s))) return rc;
    const int chains = flow_chains(m, B, s);
    if ((rc = encode_prologue(m, w, B, T, s))) return rc;
    if (chains) return run_flow(m, w, true, chains, d_h0, B, T, d_bits, d_codes, d_prob, d_all_h, d_melhat, d_hT, s);
    if ((rc = init_state(w, d_h0, B, H, s))) return rc;
    if (d_all_h && (rc = launch_repack_rows(w.hbuf, d_all_h, (long long)T * H, B, H, 1, s))) return rc;
    CallDesc d;
    memset(&d, 0, sizeof(d));
    d.p[DS_PARTD] = w.part_dec0; d.p[DS_CODES] = d_codes; d.p[DS_BITS] = const_cast<float *>(d_bits);
    d.p[DS_PROB] = d_prob; d.p[DS_ALLH] = d_all_h;
    d.T = T;
    const int kind_e = STEP_ENCODE | step_fold(m, true, T);
    if (d_melhat) {                                   // folded: keep ELU(dec.4) of every frame, dec.6 behind the recurrence; else dec.6's own output
        if (kind_e & STEP_FOLD) d.p[DS_KEEP_ENC] = w.pxB;
        else d.p[DS_MEL] = d_melhat;
    }
    if ((rc = begin_call(m, w, d, count_kernels(build_step(m, w, B, kind_e)), s))) return rc;
    if ((rc = run_recurrence(m, w, ws_base, B, T, kind_e, s))) return rc;
    if (d_hT && (rc = read_state(w, B, H, T, d_hT, s))) return rc;
    if (d_melhat && (kind_e & STEP_FOLD) && (rc = decode_epilogue(m, w.pxB, B, T, d_melhat, s))) return rc;
    return BVC_OK;
}

int run_encode(const bvc_model *m, const Workspace &w, void *ws_base, const float *d_mel, const float *d_bits,
               const float *d_h0, int B, int64_t T, float *d_codes, float *d_all_h, float *d_hT, float *d_prob,
               hipStream_t s, float *d_melhat = nullptr) {
    const int rc = run_encode_body(m, w, ws_base, d_mel, d_bits, d_h0, B, T, d_codes, d_all_h, d_hT, d_prob, d_melhat, s);
    const int rc2 = mark_call_end(s);
    return rc ? rc : rc2;
}

int run_decode_body(const bvc_model *m, const Workspace &w, void *ws_base, const float *d_codes, const float *d_h0, int B,
                    int64_t T, float *d_mel, float *d_hT, hipStream_t s) {
    const int H = m->cfg.h_dim, Z = m->cfg.z_dim;
    int rc;
    // phi_z depends on the codes only: all frames at once, outside the recurrence (bvrnn.py:223)
    CallDesc d;
    memset(&d, 0, sizeof(d));
    const int chains = flow_chains(m, B, s);
    const bool flow = chains > 0;
    const bool pre = flow || m->precomp_pz;            // (BVC_NO_PRECOMP=1: the round-1 step with both halves inside, another order of summation)
    const int kind = pre ? STEP_DECODE_PRE : STEP_DECODE;
    if (pre) {
        // ... and so do the phi_z halves of dec.0 (bvrnn.py:224) and of the GRU's input product (bvrnn.py:227)
        if ((rc = decode_prologue(m, w, d_codes, B, T, s))) return rc;
        d.p[DS_PARTD] = w.part_dec0; d.p[DS_PARTG] = w.part_gru;
        if (flow) return run_flow(m, w, false, chains, d_h0, B, T, nullptr, nullptr, nullptr, nullptr, d_mel, d_hT, s);
    } else {
        if ((rc = batched_mlp3(m, w, m->phi_z, d_codes, Z, B, T, s))) return rc;
        d.p[DS_PZ] = w.pxA;
    }
    if ((rc = init_state(w, d_h0, B, H, s))) return rc;
    d.p[DS_MEL] = d_mel;
    d.T = T;
    const int kind_d = kind | step_fold(m, false, T);
    if (kind_d & STEP_FOLD) d.p[DS_KEEP] = w.pxB;       // (idle once the batched phi_z layers are through)
    if ((rc = begin_call(m, w, d, count_kernels(build_step(m, w, B, kind_d)), s))) return rc;
    if ((rc = run_recurrence(m, w, ws_base, B, T, kind_d, s))) return rc;
    if (d_hT && (rc = read_state(w, B, H, T, d_hT, s))) return rc;
    if ((kind_d & STEP_FOLD) && d_mel && (rc = decode_epilogue(m, w.pxB, B, T, d_mel, s))) return rc;     // the decoder's output, all frames at once
    return BVC_OK;
}

int run_decode(const bvc_model *m, const Workspace &w, void *ws_base, const float *d_codes, const float *d_h0, int B,
               int64_t T, float *d_mel, float *d_hT, hipStream_t s) {
    const int rc = run_decode_body(m, w, ws_base, d_codes, d_h0, B, T, d_mel, d_hT, s);
    const int rc2 = mark_call_end(s);
    return rc ? rc : rc2;
}

// ---- BVRNN.forward (bvrnn.py:86-160): the training-time pass, forward values only --------------------
// One frame conditioned on state `sel` (0: h, the teacher-forced state; 1: h2, the state fed with generated
// features).  h2 lives in part_i (the side-branch buffer, unused here), both states ping-pong by frame parity.
std::vector<StepNode> build_forward_step(const bvc_model *m, const Workspace &w, int B, int sel, bool greedy,
                                         bool update_h, bool update_h2) {
    const int H = m->cfg.h_dim, Z = m->cfg.z_dim, X = m->cfg.num_mels;
    std::vector<StepNode> plan;
    const long long MH = (long long)((B + 15) / 16) * 16 * H;
    float *hb[2] = {w.hbuf, w.part_i};
    auto cur = [&](int which) { return dp_parity(hb[which], H, MH, 0, 1); };
    auto nxt = [&](int which) { return dp_parity(hb[which] + MH, H, -MH, 0, 1); };
    float *e1 = w.step[0], *e2 = w.step[1];
    float *pz1 = w.step[2], *pz2 = w.step[3], *pz3 = w.step[4];
    float *d1 = w.step[5], *d2 = w.step[6], *d3 = w.step[7], *dn = w.step[8];
    float *g1 = w.step[9], *g2 = w.step[10], *g3 = w.step[11];
    float *q1 = w.step[12], *q2 = w.step[13];
    auto S = [&](float *p, int ld) { return dp_static(p, ld, 1); };
    int node = 0;
    auto K = [&](GemmParams p, int epi) {
        p.desc = w.desc; p.node = node++; p.probe = nullptr;
        finish(p);
        plan.push_back(StepNode{OP_KERNEL, BR_MAIN, -1, p, epi});
    };
    const DynPtr hs = cur(sel);
    // enc_t and the sample (bvrnn.py:115-129)
    K(lin2_params(m->enc[0], dp_frame(DS_PX, H, 0, 1), H, hs, H, B, S(e1, H)), EPI_ELU);
    K(lin_params(m->enc[1], S(e1, H), B, S(e2, H)), EPI_ELU);
    {
        GemmParams p = lin_params(m->enc[2], S(e2, H), B, dp_frame(DS_CODES, Z));
        p.var_bit = m->cfg.var_bit;
        p.sample = greedy ? CS_GREEDY : CS_SAMPLE;
        p.aux = dp_frame(DS_BITS, 1);
        p.y2 = greedy ? dp_null() : dp_frame(DS_NOISE, Z);
        p.y3 = dp_frame(DS_PROB, Z);
        K(p, EPI_CODE);
    }
    // prior_t (bvrnn.py:116,119)
    K(lin_params(m->prior[0], hs, B, S(q1, H)), EPI_ELU);
    K(lin_params(m->prior[1], S(q1, H), B, S(q2, H)), EPI_ELU);
    K(lin_params(m->prior[2], S(q2, H), B, dp_frame(DS_PRIOR, Z)), EPI_SIGMOID);
    // phi_z, dec (bvrnn.py:131-137)
    K(lin_params(m->phi_z[0], dp_frame(DS_CODES, Z), B, S(pz1, H)), EPI_ELU);
    K(lin_params(m->phi_z[1], S(pz1, H), B, S(pz2, H)), EPI_ELU);
    K(lin_params(m->phi_z[2], S(pz2, H), B, S(pz3, H)), EPI_ELU);
    K(lin2_params(m->dec[0], S(pz3, H), H, hs, H, B, S(d1, H)), EPI_ELU);
    K(lin_params(m->dec[1], S(d1, H), B, S(d2, H)), EPI_ELU);
    K(lin_params(m->dec[2], S(d2, H), B, S(d3, H)), EPI_ELU);
    {
        GemmParams p = lin_params(m->dec[3], S(d3, H), B, dp_frame(DS_MEL, X));
        p.y2 = S(dn, X); p.mean = m->mean_mel; p.stdv = m->std_mel;
        K(p, EPI_MEL);
    }
    auto gru = [&](DynPtr xin, int which) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.M = B; p.N = H; p.gate_rows = H;
        p.y = nxt(which);
        p.y2 = dp_null();
        p.aux = cur(which);
        p.nseg = 3;
        p.gate_il = 1;
        p.seg[0] = mkseg(xin, m->w_ih_il, 2 * H / 16, H, 0);
        p.seg[1] = mkseg(S(pz3, H), m->w_ih_il + (size_t)(H / 16) * 3 * 256, 2 * H / 16, H, 0);
        p.seg[2] = mkseg(cur(which), m->w_hh_il, H / 16, H, 1);
        p.bias0 = m->b_ih; p.bias1 = m->b_hh;
        K(p, EPI_GRU);
    };
    if (update_h) gru(dp_frame(DS_PX, H, 0, 1), 0);                 // h  <- GRU([phi_x_t, phi_z_t], h)      bvrnn.py:142-143
    if (update_h2) {                                                 // h2 <- GRU([phi_x_t_gen, phi_z_t], h2) bvrnn.py:139,144-145
        K(lin_params(m->phi_x[0], S(dn, X), B, S(g1, H)), EPI_ELU);
        K(lin_params(m->phi_x[1], S(g1, H), B, S(g2, H)), EPI_ELU);
        K(lin_params(m->phi_x[2], S(g2, H), B, S(g3, H)), EPI_ELU);
        gru(S(g3, H), 1);
    }
    return plan;
}

int run_forward(const bvc_model *m, const Workspace &w, const float *d_mel, const float *d_bits,
                const uint8_t *h_use_gen, bool update_h, bool update_h2, const float *d_noise, int B, int64_t T,
                float *d_dec, float *d_kld, float *d_z, float *d_prob, float *d_prior, hipStream_t s) {
    const int H = m->cfg.h_dim, X = m->cfg.num_mels, Z = m->cfg.z_dim;
    const long long BT = (long long)B * T;
    int rc;
    if (!m->has_prior) { set_error("bvc_bvrnn_forward: the model was created without the prior.* tensors"); return BVC_EMISSING; }
    if (m->cfg.var_bit && !d_bits) { set_error("bits per frame required when var_bit=1"); return BVC_EINVAL; }
    if (Z > H) { set_error("bvc_bvrnn_forward: z_dim > h_dim is not supported"); return BVC_EINVAL; }
    // y = (y - mean) / std ; phi_x over all frames (bvrnn.py:96-101)
    if ((rc = launch_normalize_rows(d_mel, m->mean_mel, m->std_mel, BT, X, w.yn, s))) return rc;
    if ((rc = batched_mlp3(m, w, m->phi_x, w.yn, X, B, T, s))) return rc;
    // h = h2 = 0 (bvrnn.py:103-104); both parities so that a state that is never updated stays zero
    const long long MH = (long long)((B + 15) / 16) * 16 * H;
    if ((rc = launch_fill(w.hbuf, 0.0f, 2 * MH, s))) return rc;
    if ((rc = launch_fill(w.part_i, 0.0f, 2 * MH, s))) return rc;
    // optional outputs fall back to workspace buffers that are idle during the recurrence (Z <= H, Z <= num_mels or not:
    // pxB / pxC hold B*T*H floats each, mel B*T*num_mels)
    float *prob = d_prob ? d_prob : w.pxB;
    float *prior = d_prior ? d_prior : w.pxC;
    float *z = d_z ? d_z : (Z <= X ? w.mel : w.pxB + BT * Z);
    if (!d_z && Z > X && 2 * Z > H) { set_error("bvc_bvrnn_forward: pass d_z for this z_dim"); return BVC_EINVAL; }
    CallDesc d;
    memset(&d, 0, sizeof(d));
    d.p[DS_PX] = w.pxA; d.p[DS_CODES] = z; d.p[DS_BITS] = const_cast<float *>(d_bits);
    d.p[DS_PROB] = prob; d.p[DS_PRIOR] = prior; d.p[DS_MEL] = d_dec; d.p[DS_NOISE] = const_cast<float *>(d_noise);
    d.T = T;
    const bool greedy = d_noise == nullptr;
    const std::vector<StepNode> plan0 = build_forward_step(m, w, B, 0, greedy, update_h, update_h2);
    const std::vector<StepNode> plan1 = build_forward_step(m, w, B, 1, greedy, update_h, update_h2);
    const bool kp = g_kprobe.enabled;                 // the in-kernel probes index by a fixed kernel count per step
    g_kprobe.enabled = false;
    rc = begin_call(m, w, d, count_kernels(plan0), s);
    for (int64_t t = 0; !rc && t < T; ++t) rc = launch_steps(m, h_use_gen[t] ? plan1 : plan0, w, 1, s, nullptr);
    g_kprobe.enabled = kp;
    if (rc) return rc;
    return launch_kld_frames(prob, prior, m->cfg.var_bit ? d_bits : nullptr, B, T, Z, d_kld, s);
}

// Runs the generator; stop_after: -1 = everything, otherwise the tap index of bvc_test_vocoder_tap.
int run_vocoder(const bvc_model *m, const Workspace &w, const float *d_mel, int B, int64_t T, int64_t length,
                float div, float *d_wav, int stop_after, const float **tap, int64_t *tap_len, int *tap_ch,
                hipStream_t s) {
    const bvc_config &c = m->cfg;
    int rc;
    // pad[6,0] + conv_pre (models.py:212-213); input is already time-major (B,T,80)
    if ((rc = launch_conv_mfma(m->conv_pre, d_mel, T, w.y0, T, B, CE_STORE, nullptr, nullptr, 1.0f, s))) return rc;
    if (stop_after == 0) { *tap = w.y0; *tap_len = T; *tap_ch = c.upsample_initial_channel; return BVC_OK; }
    const float *cur_in = w.y0;
    int64_t Lin = T;
    for (int i = 0; i < c.n_up; ++i) {
        const int C = m->stage_ch[i];
        const int64_t L = (Lin + 1) * c.up_rates[i];
        // ConvTranspose1d as a 2-tap conv with u*C columns over Lin+1 rows (models.py:216-217)
        if ((rc = launch_conv_mfma(m->ups[i], cur_in, Lin, w.X, Lin + 1, B, CE_STORE, nullptr, nullptr, 1.0f, s))) return rc;
        if (stop_after == 1 + 2 * i) { *tap = w.X; *tap_len = L; *tap_ch = C; return BVC_OK; }
        for (int j = 0; j < c.n_resk; ++j) {                            // three parallel AMP blocks
            const float *cur = w.X;
            for (int d = 0; d < 3; ++d) {
                const AmpPair &ap = m->amp[i][j][d];
                if (!m->fused_amp && (rc = launch_conv_mfma(ap.c1, cur, L, w.U, L, B, CE_STORE, nullptr, nullptr, 1.0f, s))) return rc;
                float *dst;
                int epi = CE_RES;
                if (d < 2) dst = (d == 0) ? w.P : w.Q;
                else {
                    dst = w.XS;
                    epi = (j == 0) ? CE_RES : (j + 1 < c.n_resk ? CE_RES_ACC : CE_RES_ACC_DIV);
                    if (c.n_resk == 1) epi = CE_RES;
                }
                if (m->fused_amp) {
                    if ((rc = launch_amp_pair(ap.c1, ap.c2, cur, L, dst, B, epi, w.XS, (float)c.n_resk, s, nullptr, m->amp_kernels))) return rc;
                } else if ((rc = launch_conv_mfma(ap.c2, w.U, L, dst, L, B, epi, cur, w.XS, (float)c.n_resk, s))) return rc;
                cur = dst;
            }
        }
        if (stop_after == 2 + 2 * i) { *tap = w.XS; *tap_len = L; *tap_ch = C; return BVC_OK; }
        cur_in = w.XS;
        Lin = L;
    }
    const int64_t n_out = length < Lin ? length : Lin;
    return launch_conv_post(cur_in, Lin, m->post_c, m->post_ks, m->post_w, m->post_b, m->post_a, m->post_ib, div,
                            d_wav, n_out, B, s);
}


// ---- incremental (history-buffer) vocoder for streaming --------------------------------------------
// Every activation tensor of the generator is kept as a (B, H + kmax*rate, C) buffer whose first H rows
// are the last H rows of the previous hop; a hop computes only the rows of the new frames and then
// rotates the last H rows to the front of the twin buffer (ping-pong: source and destination overlap
// when fewer than H rows are new).
struct StreamTensor { float *buf[2]; int C, H, rate; long long rows; };
struct RotEntry { float *buf[2]; long long bs; int C, H, rate, pad_; };

// (one workgroup per (tensor, stream) with four 16-byte pieces in flight per thread measured slower: 67 against 56 us at 256 streams)
__global__ __launch_bounds__(256) void stream_rotate_kernel(const RotEntry *__restrict__ tab, int k, int parity) {
    const RotEntry e = tab[blockIdx.z];
    const long long n4 = (long long)e.H * e.C / 4;
    const float4 *src = reinterpret_cast<const float4 *>(e.buf[parity] + (long long)blockIdx.y * e.bs +
                                                         (long long)k * e.rate * e.C);
    float4 *dst = reinterpret_cast<float4 *>(e.buf[parity ^ 1] + (long long)blockIdx.y * e.bs);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void stream_rows_in_kernel(const float *__restrict__ src, long long src_bs,
                                                             float *__restrict__ dst, long long dst_bs, long long n) {
    const float *s = src + (long long)blockIdx.y * src_bs;
    float *d = dst + (long long)blockIdx.y * dst_bs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = s[i];
}

}  // namespace

struct bvc_vocoder_stream {
    const bvc_model *m = nullptr;
    int B = 0, kmax = 0, parity = 0;
    // slide: ONE buffer per tensor with room for cap_frames frames; the window of a hop starts `cursor` frames into it and the history is
    // moved back to the front only when the room is used up (every cap_frames / kmax - 1 hops at least) instead of after every hop.  The
    // addresses of a hop then change from hop to hop: not for a hop that is replayed from a graph (bvc_stream_codec's eager ticks only).
    bool slide = false;
    int cursor = 0, cap_frames = 0;
    int64_t frames = 0;
    float *pool = nullptr;
    size_t pool_floats = 0;
    RotEntry *d_tab = nullptr;
    int n_ten = 0, max_hc4 = 0;
    StreamTensor mel, y0;
    std::vector<StreamTensor> X, XS;                  // per stage
    std::vector<StreamTensor> P, Q;                   // per (stage, AMP block): each block's intermediates keep their own history
    ~bvc_vocoder_stream() {
        if (pool) (void)hipFree(pool);
        if (d_tab) (void)hipFree(d_tab);
    }
};

namespace {

const int64_t STREAM_WARM_FRAMES = 32;   // rate * 32 - 64 >= 60 = the longest receptive field of an AMP pair, for every stage rate >= 8
const int STREAM_H = 64;     // history rows per stage: >= (ks-1)*dil + (ks-1) of every AMP pair (max 60) and a multiple of every rate

int stream_push(bvc_vocoder_stream *st, const float *d_mel, int k, float div, float *d_wav, hipStream_t s) {
    const bvc_model *m = st->m;
    const bvc_config &c = m->cfg;
    const int B = st->B, p = st->parity;
    int rc;
    auto bs = [](const StreamTensor &t) { return t.rows * t.C; };
    // first row of this hop's window of a tensor (its history; the new rows follow)
    auto at = [&](const StreamTensor &t) { return t.buf[p] + (long long)st->cursor * t.rate * t.C; };
    // new mel rows behind the history
    stream_rows_in_kernel<<<dim3((unsigned)((k * st->mel.C + 255) / 256), B), 256, 0, s>>>(
        d_mel, (long long)k * st->mel.C, at(st->mel) + (long long)st->mel.H * st->mel.C, bs(st->mel), (long long)k * st->mel.C);
    BVC_HIP_TRY(hipGetLastError());
    // conv_pre: mel rows [Hm, Hm+k) -> y0 rows [Hy, Hy+k)
    {
        ConvWindow w{bs(st->mel), bs(st->y0), st->mel.H, 0};
        float *out = at(st->y0) + (long long)(st->y0.H - st->mel.H) * st->y0.C;
        if ((rc = launch_conv_mfma(m->conv_pre, at(st->mel), st->mel.H + k, out, st->mel.H + k, B, CE_STORE, nullptr,
                                   nullptr, 1.0f, s, &w))) return rc;
    }
    const StreamTensor *prev = &st->y0;
    long long rate_prev = 1;
    for (int i = 0; i < c.n_up; ++i) {
        const int u = c.up_rates[i];
        const StreamTensor &X = st->X[i], &XS = st->XS[i];
        // transposed conv as a 2-tap conv over the view (rows/u, u*C): view row q <-> X rows [u*q, u*q+u)
        {
            const long long hq = X.H / u;                              // history rows of the view
            const long long nq = rate_prev * k;                        // new view rows
            ConvWindow w{bs(*prev), bs(X), hq, 0};
            const float *in = at(*prev) + (long long)(prev->H - hq) * prev->C;
            if ((rc = launch_conv_mfma(m->ups[i], in, hq + nq, at(X), hq + nq, B, CE_STORE, nullptr, nullptr, 1.0f, s, &w))) return rc;
        }
        const long long L = X.H + (long long)X.rate * k;
        // t_origin only decides which rows lie before the start of the signal; from STREAM_WARM_FRAMES frames on none
        // does, so the value is frozen there (a hop captured into a hipGraph then replays with identical arguments)
        const long long fr = st->frames < STREAM_WARM_FRAMES ? st->frames : STREAM_WARM_FRAMES;
        ConvWindow w{bs(X), bs(X), X.H, (long long)X.rate * fr - X.H};
        for (int j = 0; j < c.n_resk; ++j) {
            const StreamTensor &P = st->P[i * c.n_resk + j], &Q = st->Q[i * c.n_resk + j];
            const float *cur = at(X);
            for (int d = 0; d < 3; ++d) {
                const AmpPair &ap = m->amp[i][j][d];
                float *dst;
                int epi = CE_RES;
                if (d < 2) dst = (d == 0) ? at(P) : at(Q);
                else {
                    dst = at(XS);
                    epi = (j == 0) ? CE_RES : (j + 1 < c.n_resk ? CE_RES_ACC : CE_RES_ACC_DIV);
                    if (c.n_resk == 1) epi = CE_RES;
                }
                if ((rc = launch_amp_pair(ap.c1, ap.c2, cur, L, dst, B, epi, at(XS), (float)c.n_resk, s, &w, m->amp_kernels))) return rc;
                cur = dst;
            }
        }
        prev = &XS;
        rate_prev = X.rate;
    }
    {
        ConvWindow w{bs(*prev), 0, prev->H, 0};
        if ((rc = launch_conv_post(at(*prev), prev->H + rate_prev * k, m->post_c, m->post_ks, m->post_w, m->post_b,
                                   m->post_a, m->post_ib, div, d_wav, rate_prev * k, B, s, &w))) return rc;
    }
    if (!st->slide) {
        stream_rotate_kernel<<<dim3((unsigned)((st->max_hc4 + 255) / 256), B, st->n_ten), 256, 0, s>>>(st->d_tab, k, p);
        st->parity ^= 1;
    } else {
        st->cursor += k;                                     // the next hop's window starts behind this hop's rows
        if (st->cursor + st->kmax > st->cap_frames) {        // no room for another hop: history back to the front (the twin buffer IS the buffer)
            stream_rotate_kernel<<<dim3((unsigned)((st->max_hc4 + 255) / 256), B, st->n_ten), 256, 0, s>>>(st->d_tab, st->cursor, 0);
            st->cursor = 0;
        }
    }
    BVC_HIP_TRY(hipGetLastError());
    st->frames += k;
    return BVC_OK;
}

__global__ void tap_copy_kernel(const float *src, float *dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

}  // namespace

// ---- whole-hop streaming codec (BASELINE configs[4]) -----------------------------------------------------
// B parallel streams, a fixed hop of new samples per tick; one tick = front-end of the frames the hop completes ->
// BVRNN.encode (state carried) -> BVRNN.decode (state carried) -> incremental vocoder.  Everything a tick touches
// lives at fixed device addresses and everything that changes from tick to tick (how many samples are buffered)
// is DEVICE state, so a tick with k new frames and vocoder parity p is the same launch sequence every time: it is
// captured once per (k, p) into a hipGraph and replayed.
namespace {

struct StreamDev { int fill; int pad_[3]; };             // samples buffered: sbuf[:, 0] is sample 256*F - 256, F = frames emitted

__global__ __launch_bounds__(256) void sc_append_kernel(const StreamDev *__restrict__ st, const float *__restrict__ xin, int hop,
                                                        float *__restrict__ sbuf, int cap) {
    const int fill = st->fill;
    float *d = sbuf + (long long)blockIdx.y * cap + fill;
    const float *x = xin + (long long)blockIdx.y * hop;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hop; i += gridDim.x * 256) d[i] = x[i];
}
// first tick only: samples -256..-1 of the reflect padding (meldataset.py:72-81): x[-i] = x[i]
__global__ __launch_bounds__(256) void sc_reflect_left_kernel(float *__restrict__ sbuf, int cap, int pad) {
    float *d = sbuf + (long long)blockIdx.x * cap;
    for (int i = 1 + threadIdx.x; i <= pad; i += 256) d[pad - i] = d[pad + i];
}
__global__ __launch_bounds__(256) void sc_shift_kernel(const float *__restrict__ src, long long sstride, int soff,
                                                       float *__restrict__ dst, long long dstride, int n) {
    const float *a = src + (long long)blockIdx.y * sstride + soff;
    float *d = dst + (long long)blockIdx.y * dstride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = a[i];
}
__global__ void sc_advance_kernel(StreamDev *st, int delta) { if (threadIdx.x == 0) st->fill += delta; }

}  // namespace

struct bvc_stream_codec {
    const bvc_model *m = nullptr;
    int B = 0, hop = 0, kmax = 0, cap = 0;
    float bits = 0.0f, scale = 1.0f, out_div = 1.0f;
    int fill = 0;                       // host mirror of StreamDev::fill (same arithmetic)
    int64_t frames = 0, ticks = 0;
    bool first = true;
    // device memory (one allocation)
    char *pool = nullptr;
    StreamDev *d_state = nullptr;
    float *d_in = nullptr, *sbuf = nullptr, *stmp = nullptr, *mel = nullptr, *bitsbuf = nullptr, *codes = nullptr, *melhat = nullptr,
          *wav = nullptr, *h_enc = nullptr, *h_dec = nullptr;
    void *ws = nullptr; size_t ws_bytes = 0;
    bvc_vocoder_stream *voc = nullptr;
    hipGraphExec_t graph[8][2] = {};    // [k][vocoder parity]
    bool use_graph = true;
    bool tick_flow = true;      // the ticks' recurrences on the persistent kernel where it is available (BVC_STREAM_FLOW=0: never)
    ~bvc_stream_codec() {
        for (auto &gk : graph) for (auto g : gk) if (g) (void)hipGraphExecDestroy(g);
        if (voc) bvc_vocoder_stream_destroy(voc);
        if (pool) (void)hipFree(pool);
    }
};

namespace {

// the launches of one tick with k new frames (k > 0), in stream order
int stream_tick_body(bvc_stream_codec *st, int k, hipStream_t s) {
    const bvc_model *m = st->m;
    const int B = st->B;
    int rc;
    Workspace w;
    if ((rc = check_ws(m, B, k, st->ws, st->ws_bytes, &w))) return rc;
    // front-end on the sample buffer: frame j of this tick reads sbuf[:, 256 j : 256 j + 1024)
    if ((rc = launch_stft_logmel(m->fe, st->sbuf, B, st->cap, k, 0, st->scale, st->mel, s))) return rc;
    // drop the 256 k samples no later frame reads (through a scratch copy: the ranges overlap)
    const int keep = st->cap - 256 * k;
    sc_shift_kernel<<<dim3((unsigned)((keep + 255) / 256), B), 256, 0, s>>>(st->sbuf, st->cap, 256 * k, st->stmp, st->cap, keep);
    sc_shift_kernel<<<dim3((unsigned)((keep + 255) / 256), B), 256, 0, s>>>(st->stmp, st->cap, 0, st->sbuf, st->cap, keep);
    BVC_HIP_TRY(hipGetLastError());
    if ((rc = run_encode(m, w, st->ws, st->mel, m->cfg.var_bit ? st->bitsbuf : nullptr, st->h_enc, B, k, st->codes, nullptr, st->h_enc,
                         nullptr, s))) return rc;
    if ((rc = run_decode(m, w, st->ws, st->codes, st->h_dec, B, k, st->melhat, st->h_dec, s))) return rc;
    return bvc_vocoder_stream_push(st->voc, st->melhat, k, st->out_div, st->wav, s);
}

}  // namespace

// =================================================================================================
extern "C" {

int bvc_abi_version(void) { return BVC_ABI_VERSION; }
const char *bvc_last_error(void) { return g_err; }

int bvc_model_create(const bvc_config *cfg, const bvc_tensor *tensors, int32_t n_tensors, bvc_model **out) {
    if (!out) { set_error("null out pointer"); return BVC_EINVAL; }
    *out = nullptr;
    int rc = check_config(cfg);
    if (rc) return rc;
    if (!tensors || n_tensors <= 0) { set_error("no tensors given"); return BVC_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: the gfx950 kernels cannot run (there is no CPU fallback)");
        return BVC_ENODEVICE;
    }
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i)
        if (tensors[i].name) tm[tensors[i].name] = &tensors[i];
    std::unique_ptr<bvc_model> m(new bvc_model());
    m->cfg = *cfg;
    if ((rc = conv_kernels_init())) return rc;
    if ((rc = skinny_kernels_init())) return rc;
    {
        const char *ng = getenv("BVC_NO_GRAPH");
        m->use_graph = !(ng && ng[0] == '1');
        const char *sb = getenv("BVC_SIDE_BRANCH");
        m->side_branch = (sb && sb[0] == '1');
        const char *mw = getenv("BVC_MTW");
        if (mw && (mw[0] == '2' || mw[0] == '4')) m->mtw = mw[0] - '0';
        const char *ua = getenv("BVC_UNFUSED_AMP");
        m->fused_amp = !(ua && ua[0] == '1');
        m->amp_kernels = amp_kernels_default();
        const char *np = getenv("BVC_NO_PRECOMP");
        m->precomp_pz = !(np && np[0] == '1') && !m->side_branch;
    }
    if ((rc = build_frontend(m.get(), tm))) return rc;
    if ((rc = build_bvrnn(m.get(), tm))) return rc;
    {
        const char *rr = getenv("BVC_RECURRENCE");
        m->recurrence = (rr && strcmp(rr, "layers") == 0) ? RS_LAYERS : (rr && strcmp(rr, "persistent") == 0) ? RS_PERSISTENT : RS_AUTO;
    }
    if ((rc = build_flow(m.get()))) return rc;
    if ((rc = build_vocoder(m.get(), tm))) return rc;
    BVC_HIP_TRY(hipDeviceSynchronize());
    *out = m.release();
    return BVC_OK;
}

void bvc_model_destroy(bvc_model *m) { delete m; }

int64_t bvc_num_frames(const bvc_model *m, int64_t L) {
    if (!m) return BVC_EINVAL;
    const bvc_config &c = m->cfg;
    const int64_t pr = c.n_fft - c.pad_left - c.hop;
    if (L <= c.pad_left || L <= pr) return BVC_EINVAL;       // reflect padding needs pad < L
    return (L + c.pad_left + pr - c.n_fft) / c.hop + 1;
}

int64_t bvc_vocoder_length(const bvc_model *m, int64_t T) {
    if (!m || T <= 0) return BVC_EINVAL;
    return stage_len(m, T, m->cfg.n_up - 1);
}

size_t bvc_workspace_bytes(const bvc_model *m, int32_t B, int64_t T) {
    if (!m || B <= 0 || T <= 0) return 0;
    Workspace w;
    carve(m, B, T, nullptr, &w);
    return w.total;
}

int bvc_stft_logmel(const bvc_model *m, const float *d_wav, int32_t B, int64_t L, float scale, float *d_mel,
                    void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    if (!m || !d_wav || !d_mel) { set_error("null argument"); return BVC_EINVAL; }
    const int64_t T = bvc_num_frames(m, L);
    if (B <= 0 || T <= 0) { set_error("input too short for reflect padding (L=%lld)", (long long)L); return BVC_EINVAL; }
    return launch_stft_logmel(m->fe, d_wav, B, L, T, m->cfg.pad_left, scale, d_mel, (hipStream_t)stream);
}

int bvc_bvrnn_encode(const bvc_model *m, const float *d_mel, const float *d_bits, const float *d_h0, int32_t B,
                     int64_t T, float *d_codes, float *d_all_h, float *d_hT, float *d_prob, void *d_ws,
                     size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_mel || !d_codes) { set_error("null argument"); return BVC_EINVAL; }
    return run_encode(m, w, d_ws, d_mel, d_bits, d_h0, B, T, d_codes, d_all_h, d_hT, d_prob, (hipStream_t)stream);
}

int bvc_bvrnn_decode(const bvc_model *m, const float *d_codes, const float *d_h0, int32_t B, int64_t T, float *d_mel,
                     float *d_hT, void *d_ws, size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_codes || !d_mel) { set_error("null argument"); return BVC_EINVAL; }
    return run_decode(m, w, d_ws, d_codes, d_h0, B, T, d_mel, d_hT, (hipStream_t)stream);
}

int bvc_bvrnn_forward(const bvc_model *m, const float *d_mel, const float *d_bits, const uint8_t *h_use_gen,
                      int32_t update_h, int32_t update_h2, const float *d_noise, int32_t B, int64_t T, float *d_dec,
                      float *d_kld, float *d_z, float *d_prob, float *d_prior, void *d_ws, size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_mel || !h_use_gen || !d_dec || !d_kld) { set_error("null argument"); return BVC_EINVAL; }
    if (!update_h && !update_h2) { set_error("bvc_bvrnn_forward: at least one state must be updated"); return BVC_EINVAL; }
    for (int64_t t = 0; t < T; ++t)
        if ((h_use_gen[t] && !update_h2) || (!h_use_gen[t] && !update_h)) {
            set_error("bvc_bvrnn_forward: frame %lld is conditioned on a state that is never updated", (long long)t);
            return BVC_EINVAL;
        }
    return run_forward(m, w, d_mel, d_bits, h_use_gen, update_h != 0, update_h2 != 0, d_noise, B, T, d_dec, d_kld, d_z,
                       d_prob, d_prior, (hipStream_t)stream);
}

int bvc_bigvgan(const bvc_model *m, const float *d_mel, int32_t B, int64_t T, int64_t length, float out_scale_div,
                float *d_wav, void *d_ws, size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_mel || !d_wav || length <= 0) { set_error("null argument or non-positive length"); return BVC_EINVAL; }
    return run_vocoder(m, w, d_mel, B, T, length, out_scale_div, d_wav, -1, nullptr, nullptr, nullptr,
                       (hipStream_t)stream);
}

int bvc_encode(const bvc_model *m, const float *d_wav, int32_t B, int64_t L, float scale, float bits_per_frame,
               float *d_codes, void *d_ws, size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    if (!m) { set_error("null model"); return BVC_EINVAL; }
    const int64_t T = bvc_num_frames(m, L);
    if (T <= 0) { set_error("input too short for reflect padding (L=%lld)", (long long)L); return BVC_EINVAL; }
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_wav || !d_codes) { set_error("null argument"); return BVC_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_stft_logmel(m->fe, d_wav, B, L, T, m->cfg.pad_left, scale, w.mel, s))) return rc;
    if ((rc = launch_fill(w.bits, bits_per_frame, (long long)B * T, s))) return rc;
    return run_encode(m, w, d_ws, w.mel, w.bits, nullptr, B, T, d_codes, nullptr, nullptr, nullptr, s);
}

int bvc_forward(const bvc_model *m, const float *d_wav, int32_t B, int64_t L, float scale, float bits_per_frame, int64_t length,
                float out_scale_div, float *d_codes, float *d_wav_out, void *d_ws, size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    if (!m) { set_error("null model"); return BVC_EINVAL; }
    const int64_t T = bvc_num_frames(m, L);
    if (T <= 0) { set_error("input too short for reflect padding (L=%lld)", (long long)L); return BVC_EINVAL; }
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_wav || !d_wav_out || length <= 0) { set_error("null argument or non-positive length"); return BVC_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_stft_logmel(m->fe, d_wav, B, L, T, m->cfg.pad_left, scale, w.mel, s))) return rc;
    if ((rc = launch_fill(w.bits, bits_per_frame, (long long)B * T, s))) return rc;
    // the encoder's recurrence runs the decoder of every frame anyway (bvrnn.py:198-204): its outputs ARE what BVRNN.decode(codes) would
    // compute over again from the same states.  w.mel has been consumed (normalised into another buffer) before the first of them is
    // written; codes nobody asked for go to a workspace tensor that encode does not use.
    float *codes = d_codes ? d_codes : w.part_gru;
    if ((rc = run_encode(m, w, d_ws, w.mel, w.bits, nullptr, B, T, codes, nullptr, nullptr, nullptr, s, w.mel))) return rc;
    return run_vocoder(m, w, w.mel, B, T, length, out_scale_div, d_wav_out, -1, nullptr, nullptr, nullptr, s);
}

int bvc_decode(const bvc_model *m, const float *d_codes, int32_t B, int64_t T, int64_t length, float out_scale_div,
               float *d_wav, void *d_ws, size_t ws_bytes, void *stream) {
    if (int st_ = sticky_status(m)) return st_;
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (!d_codes || !d_wav || length <= 0) { set_error("null argument or non-positive length"); return BVC_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    if ((rc = run_decode(m, w, d_ws, d_codes, nullptr, B, T, w.mel, nullptr, s))) return rc;
    return run_vocoder(m, w, w.mel, B, T, length, out_scale_div, d_wav, -1, nullptr, nullptr, nullptr, s);
}

static int vocoder_stream_create(const bvc_model *m, int32_t B, int32_t max_frames_per_push, bool slide, bvc_vocoder_stream **out);

int bvc_vocoder_stream_create(const bvc_model *m, int32_t B, int32_t max_frames_per_push, bvc_vocoder_stream **out) {
    return vocoder_stream_create(m, B, max_frames_per_push, false, out);
}

static int vocoder_stream_create(const bvc_model *m, int32_t B, int32_t max_frames_per_push, bool slide, bvc_vocoder_stream **out) {
    if (!m || !out || B <= 0 || max_frames_per_push <= 0) { set_error("bvc_vocoder_stream_create: bad arguments"); return BVC_EINVAL; }
    const bvc_config &c = m->cfg;
    // the history must cover every receptive field and stay aligned with the transposed-conv views
    long long rate = 1;
    for (int i = 0; i < c.n_up; ++i) {
        const int u = c.up_rates[i];
        if (STREAM_H % u) { set_error("streaming vocoder: upsample rate %d does not divide the history (%d)", u, STREAM_H); return BVC_EINVAL; }
        for (int j = 0; j < c.n_resk; ++j)
            for (int d = 0; d < 3; ++d) {
                const AmpPair &ap = m->amp[i][j][d];
                if ((ap.c1.ks - 1) * ap.c1.dil + (ap.c2.ks - 1) * ap.c2.dil > STREAM_H) {
                    set_error("streaming vocoder: AMP receptive field exceeds the history"); return BVC_EINVAL;
                }
            }
        rate *= u;
    }
    if (m->post_ks - 1 > STREAM_H) { set_error("streaming vocoder: conv_post kernel exceeds the history"); return BVC_EINVAL; }
    std::unique_ptr<bvc_vocoder_stream> st(new bvc_vocoder_stream());
    st->m = m; st->B = B; st->kmax = max_frames_per_push;
    st->slide = slide;
    // frames of room behind the history: 32 (16 hops of one or two frames between two moves of the histories; 2.2 GB of buffers at 256 streams),
    // more only where longer hops need it
    st->cap_frames = slide ? std::max(2 * max_frames_per_push + 8, 32) : max_frames_per_push;
    const long long room = st->cap_frames;
    auto mk = [&](int C, int H, int r) { StreamTensor t; t.buf[0] = t.buf[1] = nullptr; t.C = C; t.H = H; t.rate = r; t.rows = H + (long long)r * room; return t; };
    st->mel = mk(c.num_mels, (m->conv_pre.ks - 1) * m->conv_pre.dil, 1);
    st->y0 = mk(c.upsample_initial_channel, STREAM_H / c.up_rates[0], 1);
    if (st->y0.H < st->mel.H) st->y0.H = st->mel.H, st->y0.rows = st->y0.H + room;
    rate = 1;
    for (int i = 0; i < c.n_up; ++i) {
        rate *= c.up_rates[i];
        for (auto *v : {&st->X, &st->XS}) v->push_back(mk(m->stage_ch[i], STREAM_H, (int)rate));
        for (int j = 0; j < c.n_resk; ++j)
            for (auto *v : {&st->P, &st->Q}) v->push_back(mk(m->stage_ch[i], STREAM_H, (int)rate));
    }
    std::vector<StreamTensor *> all = {&st->mel, &st->y0};
    for (auto *v : {&st->X, &st->XS, &st->P, &st->Q})
        for (auto &t : *v) all.push_back(&t);
    size_t total = 0;
    const int copies = slide ? 1 : 2;
    for (auto *t : all) {
        total += copies * (size_t)B * t->rows * t->C;
        // moving the history back to the front must not overlap itself: it happens with more than cap_frames - 2 kmax frames behind it
        if (slide && (long long)(st->cap_frames - 2 * st->kmax + 1) * t->rate < t->H) { set_error("streaming vocoder: sliding window too short for its history"); return BVC_EINVAL; }
    }
    if (hipMalloc(reinterpret_cast<void **>(&st->pool), total * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("streaming vocoder: cannot allocate %zu bytes of history buffers", total * sizeof(float));
        return BVC_ENOMEM;
    }
    st->pool_floats = total;
    size_t off = 0;
    std::vector<RotEntry> tab;
    for (auto *t : all) {
        for (int q = 0; q < copies; ++q) { t->buf[q] = st->pool + off; off += (size_t)B * t->rows * t->C; }
        if (slide) t->buf[1] = t->buf[0];
        RotEntry e; e.buf[0] = t->buf[0]; e.buf[1] = t->buf[1]; e.bs = t->rows * t->C; e.C = t->C; e.H = t->H; e.rate = t->rate; e.pad_ = 0;
        tab.push_back(e);
        if ((t->H * t->C) % 4) { set_error("streaming vocoder: history of a tensor is not a multiple of 4 floats"); return BVC_EINVAL; }
        st->max_hc4 = std::max(st->max_hc4, t->H * t->C / 4);
    }
    st->n_ten = (int)tab.size();
    BVC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&st->d_tab), tab.size() * sizeof(RotEntry)));
    BVC_HIP_TRY(hipMemcpy(st->d_tab, tab.data(), tab.size() * sizeof(RotEntry), hipMemcpyHostToDevice));
    BVC_HIP_TRY(hipMemset(st->pool, 0, total * sizeof(float)));
    *out = st.release();
    return BVC_OK;
}

void bvc_vocoder_stream_destroy(bvc_vocoder_stream *st) { delete st; }

int bvc_vocoder_stream_reset(bvc_vocoder_stream *st, void *stream) {
    if (!st) { set_error("null stream state"); return BVC_EINVAL; }
    BVC_HIP_TRY(hipMemsetAsync(st->pool, 0, st->pool_floats * sizeof(float), (hipStream_t)stream));
    st->parity = 0; st->frames = 0; st->cursor = 0;
    return BVC_OK;
}

int bvc_vocoder_stream_push(bvc_vocoder_stream *st, const float *d_mel, int32_t k, float out_scale_div, float *d_wav,
                            void *stream) {
    if (!st || !d_mel || !d_wav) { set_error("null argument"); return BVC_EINVAL; }
    if (int st_ = sticky_status(st->m)) return st_;
    if (k <= 0 || k > st->kmax) { set_error("bvc_vocoder_stream_push: k=%d outside 1..%d", (int)k, st->kmax); return BVC_EINVAL; }
    return stream_push(st, d_mel, k, out_scale_div, d_wav, (hipStream_t)stream);
}

#ifdef BVC_PHASE_PROBE
int bvc_phase_probe_read(unsigned long long *out, int reset) { return bvc::phase_probe_read(out, reset); }
#endif

int bvc_stream_codec_create(const bvc_model *m, int32_t B, int32_t hop_samples, float bits_per_frame, float scale,
                            float out_scale_div, bvc_stream_codec **out) {
    if (!m || !out || B <= 0 || hop_samples <= 0) { set_error("bvc_stream_codec_create: bad arguments"); return BVC_EINVAL; }
    const bvc_config &c = m->cfg;
    if (hop_samples <= c.pad_left) { set_error("bvc_stream_codec_create: the hop must exceed the left reflect padding (%d samples)", c.pad_left); return BVC_EINVAL; }
    std::unique_ptr<bvc_stream_codec> st(new bvc_stream_codec());
    st->m = m; st->B = B; st->hop = hop_samples; st->bits = bits_per_frame; st->scale = scale; st->out_div = out_scale_div;
    st->kmax = (hop_samples + c.hop - 1) / c.hop + 1;
    if (st->kmax > 7) { set_error("bvc_stream_codec_create: hop too long (%d frames per tick)", st->kmax); return BVC_EINVAL; }
    st->cap = c.n_fft + c.hop * st->kmax + hop_samples;
    st->ws_bytes = bvc_workspace_bytes(m, B, st->kmax);
    int spf = 1;                                             // samples per frame
    for (int i = 0; i < c.n_up; ++i) spf *= c.up_rates[i];
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_state = take(sizeof(StreamDev)), o_in = take((size_t)B * hop_samples * 4), o_sbuf = take((size_t)B * st->cap * 4),
                 o_stmp = take((size_t)B * st->cap * 4), o_mel = take((size_t)B * st->kmax * c.num_mels * 4),
                 o_bits = take((size_t)B * st->kmax * 4), o_codes = take((size_t)B * st->kmax * c.z_dim * 4),
                 o_melhat = take((size_t)B * st->kmax * c.num_mels * 4), o_wav = take((size_t)B * st->kmax * spf * 4),
                 o_he = take((size_t)B * c.h_dim * 4), o_hd = take((size_t)B * c.h_dim * 4), o_ws = take(st->ws_bytes);
    if (hipMalloc(reinterpret_cast<void **>(&st->pool), off) != hipSuccess) {
        (void)hipGetLastError();
        set_error("bvc_stream_codec_create: cannot allocate %zu bytes", off);
        return BVC_ENOMEM;
    }
    BVC_HIP_TRY(hipMemset(st->pool, 0, off));
    char *p = st->pool;
    st->d_state = reinterpret_cast<StreamDev *>(p + o_state);
    st->d_in = reinterpret_cast<float *>(p + o_in); st->sbuf = reinterpret_cast<float *>(p + o_sbuf); st->stmp = reinterpret_cast<float *>(p + o_stmp);
    st->mel = reinterpret_cast<float *>(p + o_mel); st->bitsbuf = reinterpret_cast<float *>(p + o_bits); st->codes = reinterpret_cast<float *>(p + o_codes);
    st->melhat = reinterpret_cast<float *>(p + o_melhat); st->wav = reinterpret_cast<float *>(p + o_wav);
    st->h_enc = reinterpret_cast<float *>(p + o_he); st->h_dec = reinterpret_cast<float *>(p + o_hd); st->ws = p + o_ws;
    int rc;
    if ((rc = launch_fill(st->bitsbuf, bits_per_frame, (long long)B * st->kmax, nullptr))) return rc;
    st->fill = c.pad_left;                                   // room for the left reflect padding of frame 0
    StreamDev init{st->fill, {0, 0, 0}};
    BVC_HIP_TRY(hipMemcpy(st->d_state, &init, sizeof(init), hipMemcpyHostToDevice));
    { const char *ng = getenv("BVC_STREAM_NO_GRAPH"); st->use_graph = !(ng && ng[0] == '1'); }
    { const char *tf = getenv("BVC_STREAM_FLOW"); st->tick_flow = !(tf && tf[0] == '0'); }
    // ticks that are launched eagerly (persistent recurrence) let the generator's windows slide through their buffers instead of moving
    // every history back after every hop (56 us of a 1.5 ms tick at 256 streams); BVC_STREAM_SLIDE=0: never
    const char *sl = getenv("BVC_STREAM_SLIDE");
    const bool slide = st->tick_flow && flow_chains_static(m, B) != 0 && !(sl && sl[0] == '0');
    if ((rc = vocoder_stream_create(m, B, st->kmax, slide, &st->voc))) return rc;
    BVC_HIP_TRY(hipDeviceSynchronize());
    *out = st.release();
    return BVC_OK;
}

void bvc_stream_codec_destroy(bvc_stream_codec *st) { delete st; }

int bvc_stream_codec_buffers(bvc_stream_codec *st, float **d_in, float **d_codes, float **d_wav, int32_t *max_frames_per_tick) {
    if (!st) { set_error("null stream codec"); return BVC_EINVAL; }
    if (d_in) *d_in = st->d_in;
    if (d_codes) *d_codes = st->codes;
    if (d_wav) *d_wav = st->wav;
    if (max_frames_per_tick) *max_frames_per_tick = st->kmax;
    return BVC_OK;
}

int bvc_stream_codec_tick(bvc_stream_codec *st, int32_t *n_frames, void *stream) {
    if (!st) { set_error("null stream codec"); return BVC_EINVAL; }
    if (int st_ = sticky_status(st->m)) return st_;
    const bvc_config &c = st->m->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int B = st->B;
    // the hop joins the sample buffer (device-side fill level), frame 0's left reflect padding once the first samples are there
    sc_append_kernel<<<dim3((unsigned)((st->hop + 255) / 256), B), 256, 0, s>>>(st->d_state, st->d_in, st->hop, st->sbuf, st->cap);
    if (st->first) {
        sc_reflect_left_kernel<<<dim3(B), 256, 0, s>>>(st->sbuf, st->cap, c.pad_left);
        st->first = false;
    }
    BVC_HIP_TRY(hipGetLastError());
    const int fill = st->fill + st->hop;
    const int k = fill >= c.n_fft ? (fill - c.n_fft) / c.hop + 1 : 0;
    if (k > st->kmax) { set_error("bvc_stream_codec_tick: internal frame count %d", k); return BVC_EINVAL; }
    int rc = BVC_OK;
    if (k > 0) {
        const int parity = st->voc->parity;
        // Which schedule?  Where the persistent recurrence kernel is available (flow_chains_static: the model's option, the
        // residency census, the batch) the tick is launched eagerly and its two recurrences are one persistent launch each - at 256
        // streams 1.53 ms per tick against 1.69 ms for the launch-per-layer recurrence, which gains nothing from a graph on the GPU
        // side (1.68 eager / 1.70 replayed; the replay only saves host time).  Otherwise (recurrence = layers, no resident grid,
        // BVC_STREAM_FLOW=0) the warm tick is one hipGraph of launch-per-layer kernels as before.  Same bits either way.
        const bool tick_flow = st->tick_flow && flow_chains_static(st->m, B) != 0;
        const bool warm = !tick_flow && !st->voc->slide && st->use_graph && s != nullptr && st->frames >= STREAM_WARM_FRAMES;     // (the default stream cannot be captured)
        g_stream_tick = true; g_tick_flow = tick_flow;
        if (!warm) {
            rc = stream_tick_body(st, k, s);
        } else {
            hipGraphExec_t &ge = st->graph[k][parity];
            if (!ge) {                                       // first warm tick of this shape: capture it (the capture does not execute)
                hipGraph_t graph = nullptr;
                const int vp = st->voc->parity; const int64_t vf = st->voc->frames;
                g_capturing = true;
                hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) rc = stream_tick_body(st, k, s);
                hipError_t e2 = (e == hipSuccess) ? hipStreamEndCapture(s, &graph) : e;
                g_capturing = false;
                st->voc->parity = vp; st->voc->frames = vf;  // the captured push advanced the host-side bookkeeping: undo, the replay redoes it
                if (!rc && (e2 != hipSuccess || !graph)) { set_error("bvc_stream_codec_tick: hipGraph capture failed: %s", hipGetErrorString(e2)); rc = BVC_EHIP; }
                if (!rc && hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0) != hipSuccess) { set_error("bvc_stream_codec_tick: hipGraphInstantiate failed"); rc = BVC_EHIP; }
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!rc) {
                if (hipGraphLaunch(ge, s) != hipSuccess) { set_error("bvc_stream_codec_tick: hipGraphLaunch failed"); rc = BVC_EHIP; }
                st->voc->parity ^= 1; st->voc->frames += k;  // what stream_push() does on the host side
            }
        }
        g_stream_tick = false; g_tick_flow = false;
        if (rc) return rc;
    }
    sc_advance_kernel<<<1, 64, 0, s>>>(st->d_state, st->hop - c.hop * k);
    BVC_HIP_TRY(hipGetLastError());
    st->fill = fill - c.hop * k;
    st->frames += k;
    st->ticks += 1;
    if (n_frames) *n_frames = k;
    return BVC_OK;
}

int bvc_model_set_option(bvc_model *m, const char *name, int32_t value) {
    if (!m || !name) { set_error("bvc_model_set_option: null argument"); return BVC_EINVAL; }
    if (strcmp(name, "recurrence") == 0) {
        // 0: persistent kernel for every call, 1: one launch per layer for every call, 2 (default): automatic - persistent while
        // calls come one at a time, launch per layer while calls of several streams overlap (see flow_chains)
        if (value < 0 || value > 2) { set_error("bvc_model_set_option: recurrence must be 0 (persistent), 1 (layers) or 2 (auto)"); return BVC_EINVAL; }
        m->recurrence = value;
        return BVC_OK;
    }
    if (strcmp(name, "flow_spin_limit") == 0) {            // polls before a wait inside the persistent kernel gives up (tests)
        if (value < 1) { set_error("bvc_model_set_option: flow_spin_limit must be positive"); return BVC_EINVAL; }
        m->flow_spin_limit = (unsigned)value;
        return BVC_OK;
    }
    if (strcmp(name, "flow_debug_withhold") == 0) {        // tests only: workgroup 0 of every persistent launch does nothing
        m->flow_debug_withhold = value != 0;
        return BVC_OK;
    }
    if (strcmp(name, "decode_fold") == 0) {                // 1 (default): dec.6 -> norm -> phi_x.0 as one layer in the persistent decode kernel
        m->decode_fold = value != 0;
        return BVC_OK;
    }
    if (strcmp(name, "encode_fold") == 0) {                // 1 (default): the same in the persistent encode kernel
        m->encode_fold = value != 0;
        return BVC_OK;
    }
    if (strcmp(name, "flow_debug_nofill") == 0) {          // tests only: the layer program without filler quanta
        m->flow_debug_nofill = value != 0;
        return BVC_OK;
    }
    if (strcmp(name, "vocoder_c16_kernel") == 0) {         // 1 (default): C = 16 AMP pairs on the persistent kernel; 0: generic kernel.  Same bits
        m->amp_kernels = value ? (m->amp_kernels | AMPK_C16) : (m->amp_kernels & ~AMPK_C16);
        return BVC_OK;
    }
    if (strcmp(name, "vocoder_full_tiles") == 0) {         // 1 (default): C = 8 AMP pairs on the two-rows-per-tile kernel; 0: generic kernel.  Same bits
        m->amp_kernels = value ? (m->amp_kernels | AMPK_C8) : (m->amp_kernels & ~AMPK_C8);
        return BVC_OK;
    }
    set_error("bvc_model_set_option: unknown option '%s'", name);
    return BVC_EINVAL;
}

int bvc_flow_fence(void *stream) {
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    BVC_HIP_TRY(hipStreamIsCapturing(s, &cs));
    if (cs != hipStreamCaptureStatusNone) { set_error("bvc_flow_fence: not while the stream is being captured"); return BVC_EINVAL; }
    int dev = 0;
    BVC_HIP_TRY(hipGetDevice(&dev));
    dev &= 15;
    std::lock_guard<std::mutex> lk(g_flow_mu);
    FlowFence &f = g_fence[dev][g_fence_n[dev]++ % FLOW_FENCES];
    if (!f.ev) BVC_HIP_TRY(hipEventCreateWithFlags(&f.ev, hipEventDisableTiming));
    // a slot that is still pending holds a fence nobody has waited for yet (more than FLOW_FENCES fences without a persistent launch in
    // between): order this stream behind it first, so that the new record implies the old one
    if (f.pending) BVC_HIP_TRY(hipStreamWaitEvent(s, f.ev, 0));
    BVC_HIP_TRY(hipEventRecord(f.ev, s));
    f.pending = true;
    return BVC_OK;
}

int bvc_model_get_option(const bvc_model *m, const char *name, int32_t *value) {
    if (!m || !name || !value) { set_error("bvc_model_get_option: null argument"); return BVC_EINVAL; }
    if (strcmp(name, "recurrence") == 0) { *value = m->recurrence; return BVC_OK; }
    if (strcmp(name, "decode_fold") == 0) { *value = m->decode_fold; return BVC_OK; }
    if (strcmp(name, "encode_fold") == 0) { *value = m->encode_fold; return BVC_OK; }
    if (strcmp(name, "flow_resident") == 0) { *value = m->flow_resident ? 1 : 0; return BVC_OK; }       // result of the residency census
    if (strcmp(name, "flow_supported") == 0) { *value = m->flow_perh > 0 ? 1 : 0; return BVC_OK; }    // h_dim laid out for the persistent kernel
    if (strcmp(name, "compute_units") == 0) { *value = m->cu_count; return BVC_OK; }
    set_error("bvc_model_get_option: unknown option '%s'", name);
    return BVC_EINVAL;
}

int bvc_model_status(const bvc_model *m, uint32_t *code) {
    if (!m) { set_error("null model"); return BVC_EINVAL; }
    unsigned v = 0;
    if (m->h_status) {
        BVC_HIP_TRY(hipDeviceSynchronize());
        v = *m->h_status;
        if (v) { *m->h_status = 0u; m->census_due = true; }
    }
    if (code) *code = v;
    if (v) {
        set_error("a persistent recurrence kernel gave up waiting (frame %u, layer %u): its results are invalid",
                  (v & 0x7FFFFFFFu) >> 4, (v & 15u));
        return BVC_ETIMEOUT;
    }
    return BVC_OK;
}

int bvc_model_poll_status(const bvc_model *m, uint32_t *code) {
    if (!m) { set_error("null model"); return BVC_EINVAL; }
    unsigned v = 0;
    if (m->h_status) {
        v = *m->h_status;
        if (v) { *m->h_status = 0u; m->census_due = true; }
    }
    if (code) *code = v;
    if (v) {
        set_error("a persistent recurrence kernel gave up waiting (frame %u, layer %u): the results of the call that has just been "
                  "synchronised are invalid.  All its workgroups must be resident together - is another process using this GPU?",
                  (v & 0x7FFFFFFFu) >> 4, (v & 15u));
        return BVC_ETIMEOUT;
    }
    return BVC_OK;
}

int bvc_probe_begin(int32_t kind, int32_t sample_every, int32_t max_samples) {
    if (kind <= PK_NONE || kind > PK_POST || sample_every < 1 || max_samples < 1) {
        set_error("bvc_probe_begin: bad arguments");
        return BVC_EINVAL;
    }
    while ((int)g_probe.ev.size() < 2 * max_samples) {
        hipEvent_t e;
        BVC_HIP_TRY(hipEventCreate(&e));
        g_probe.ev.push_back(e);
    }
    g_probe.kind = kind; g_probe.every = sample_every; g_probe.counter = 0; g_probe.used = 0;
    return BVC_OK;
}

int bvc_probe_end(double *mean_us, double *min_us, int32_t *n_samples) {
    g_probe.kind = PK_NONE;
    BVC_HIP_TRY(hipDeviceSynchronize());
    double sum = 0.0, mn = 1e30;
    for (int i = 0; i < g_probe.used; ++i) {
        float ms = 0.0f;
        BVC_HIP_TRY(hipEventElapsedTime(&ms, g_probe.ev[2 * i], g_probe.ev[2 * i + 1]));
        sum += ms * 1e3;
        if (ms * 1e3 < mn) mn = ms * 1e3;
    }
    if (mean_us) *mean_us = g_probe.used ? sum / g_probe.used : 0.0;
    if (min_us) *min_us = g_probe.used ? mn : 0.0;
    if (n_samples) *n_samples = g_probe.used;
    return BVC_OK;
}

int bvc_resample_poly(const float *d_x, int32_t B, int64_t L_in, const double *d_h, int32_t ntaps, int32_t up, int32_t down,
                      int64_t n_pre_remove, float *d_y, int64_t n_out, void *stream) {
    if (!d_x || !d_h || !d_y || B <= 0 || L_in <= 0 || ntaps <= 0 || up < 1 || down < 1 || n_pre_remove < 0 || n_out <= 0) {
        set_error("bvc_resample_poly: bad arguments");
        return BVC_EINVAL;
    }
    return launch_resample_poly(d_x, B, L_in, d_h, ntaps, up, down, n_pre_remove, d_y, n_out, (hipStream_t)stream);
}

int bvc_peak_normalize(float *d_x, int32_t B, int64_t L, void *stream) {
    if (!d_x || B <= 0 || L <= 0) { set_error("bvc_peak_normalize: bad arguments"); return BVC_EINVAL; }
    return launch_peak_normalize(d_x, B, L, (hipStream_t)stream);
}

int bvc_pack_codes(const float *d_codes, int32_t B, int64_t T, int32_t z_dim, int32_t nbits, uint8_t *d_bytes, void *stream) {
    if (!d_codes || !d_bytes || B <= 0 || T <= 0 || z_dim <= 0 || nbits < 0 || nbits > z_dim) {
        set_error("bvc_pack_codes: bad arguments");
        return BVC_EINVAL;
    }
    return launch_pack_codes(d_codes, (long long)B * T, z_dim, nbits, d_bytes, (hipStream_t)stream);
}

int bvc_unpack_codes(const uint8_t *d_bytes, int32_t B, int64_t T, int32_t z_dim, int32_t nbits, float *d_codes, void *stream) {
    if (!d_codes || (!d_bytes && nbits > 0) || B <= 0 || T <= 0 || z_dim <= 0 || nbits < 0 || nbits > z_dim) {
        set_error("bvc_unpack_codes: bad arguments");
        return BVC_EINVAL;
    }
    return launch_unpack_codes(d_bytes, (long long)B * T, z_dim, nbits, d_codes, (hipStream_t)stream);
}

int bvc_kprobe_enable(int32_t on) {
    if (on && !g_kprobe.dev) {
        const size_t cap = (size_t)FLOW_STAMPS * 16 * 4096;     // up to 4096 frames x 16 nodes (x stamp kinds of the persistent kernel)
        BVC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_kprobe.dev), cap * sizeof(unsigned long long)));
        g_kprobe.capacity = cap;
    }
    g_kprobe.enabled = on != 0;
    return BVC_OK;
}

int bvc_kprobe_read(int32_t node_lo, int32_t node_hi, double *mean_us, double *min_us, int32_t *n_samples) {
    BVC_HIP_TRY(hipDeviceSynchronize());
    double sum = 0.0, mn = 1e30;
    int n = 0;
    if (g_kprobe.dev && g_kprobe.T > 0) {
        const size_t cnt = (size_t)2 * g_kprobe.T * g_kprobe.nodes;
        std::vector<unsigned long long> h(cnt);
        BVC_HIP_TRY(hipMemcpy(h.data(), g_kprobe.dev, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (long long t = 0; t < g_kprobe.T; ++t)
            for (int k = node_lo; k < node_hi && k < g_kprobe.nodes; ++k) {
                const unsigned long long a = h[t * g_kprobe.nodes + k], b = h[cnt / 2 + t * g_kprobe.nodes + k];
                if (b <= a) continue;
                const double us = (double)(b - a) * 0.01;        // 100 MHz ticks
                sum += us; if (us < mn) mn = us; ++n;
            }
    }
    if (mean_us) *mean_us = n ? sum / n : 0.0;
    if (min_us) *min_us = n ? mn : 0.0;
    if (n_samples) *n_samples = n;
    return BVC_OK;
}

// Raw stamps of the persistent recurrence's probing wave: mean / min of (stamp kind `to` - stamp kind `from`) over the frames,
// for the layers [node_lo, node_hi).  from = -1: `to` of layer k against stamp 1 (published) of layer k - 1, i.e. since the
// previous layer's output left.
int bvc_kprobe_read_span(int32_t from, int32_t to, int32_t node_lo, int32_t node_hi, double *mean_us, double *min_us, int32_t *n_samples) {
    BVC_HIP_TRY(hipDeviceSynchronize());
    double sum = 0.0, mn = 1e30;
    int n = 0;
    if (from < -1 || from >= FLOW_STAMPS || to < 0 || to >= FLOW_STAMPS) { set_error("bvc_kprobe_read_span: stamp kinds are 0..%d", FLOW_STAMPS - 1); return BVC_EINVAL; }
    if (g_kprobe.dev && g_kprobe.T > 0 && (size_t)FLOW_STAMPS * g_kprobe.T * g_kprobe.nodes <= g_kprobe.capacity) {
        const size_t per = (size_t)g_kprobe.T * g_kprobe.nodes, cnt = (size_t)FLOW_STAMPS * per;
        std::vector<unsigned long long> h(cnt);
        BVC_HIP_TRY(hipMemcpy(h.data(), g_kprobe.dev, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (long long t = 0; t < g_kprobe.T; ++t)
            for (int k = node_lo; k < node_hi && k < g_kprobe.nodes; ++k) {
                unsigned long long a = 0;
                if (from >= 0) a = h[(size_t)from * per + t * g_kprobe.nodes + k];
                else {
                    // from the previous layer's "published" stamp: the nearest earlier node that was stamped at all (with the folded hop the
                    // program has no dec.6 node: its slot stays empty), wrapping into the previous frame's last layer
                    long long idx = t * g_kprobe.nodes + k - 1;
                    for (int back = 0; back < g_kprobe.nodes && idx >= 0 && !a; ++back, --idx) a = h[per + idx];
                    if (!a) continue;
                }
                const unsigned long long b = h[(size_t)to * per + t * g_kprobe.nodes + k];
                if (!a || b <= a) continue;
                const double us = (double)(b - a) * 0.01;        // 100 MHz ticks
                sum += us; if (us < mn) mn = us; ++n;
            }
    }
    if (mean_us) *mean_us = n ? sum / n : 0.0;
    if (min_us) *min_us = n ? mn : 0.0;
    if (n_samples) *n_samples = n;
    return BVC_OK;
}

int bvc_test_linear(const float *d_x, const float *d_w, const float *d_bias, int32_t M, int32_t N, int32_t K,
                    int32_t act, float *d_y, void *stream) {
    // test helper (allocates + synchronises): packs the natural weight on the device first
    if (K % 16 || N % 16) { set_error("bvc_test_linear: N and K must be multiples of 16"); return BVC_EINVAL; }
    float *wp = nullptr;
    BVC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&wp), (size_t)N * K * sizeof(float)));
    hipStream_t s = (hipStream_t)stream;
    // W[N][K] natural == "matrix with N rows of length K": its B-operand packing equals the A-operand packing
    int rc = launch_repack_rows(d_w, wp, K, N, K, 0, s);
    Linear l; l.w = d_w; l.wp = wp; l.b = d_bias; l.in = K; l.out = N;
    if (!rc) rc = launch_gemm_skinny(lin_params(l, dp_static(d_x, K), M, dp_static(d_y, N)), act ? EPI_ELU : EPI_LINEAR, s);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(wp);
    if (!rc && e != hipSuccess) { set_error("bvc_test_linear: %s", hipGetErrorString(e)); rc = BVC_EHIP; }
    return rc;
}

int bvc_test_linear_batched(const float *d_x, const float *d_w, const float *d_bias, int32_t M, int32_t N, int32_t K,
                            int32_t act, float *d_y, void *stream) {
    return launch_gemm_batched(d_x, K, d_w, K, d_bias, M, N, K, act, d_y, N, (hipStream_t)stream);
}

int bvc_test_snakebeta(const float *d_x, int64_t n, float alpha, float beta, float *d_y, void *stream) {
    if (!d_x || !d_y || n <= 0) { set_error("bvc_test_snakebeta: bad arguments"); return BVC_EINVAL; }
    const float a = (float)std::exp((double)alpha);                                  // as make_conv() derives them
    const float ib = 1.0f / ((float)std::exp((double)beta) + 0.000000001f);
    return launch_snakebeta_test(d_x, n, a, ib, d_y, (hipStream_t)stream);
}

int bvc_test_vocoder_tap(const bvc_model *m, const float *d_mel, int32_t B, int64_t T, int32_t which, float *d_out,
                         int64_t *out_numel_per_batch, void *d_ws, size_t ws_bytes, void *stream) {
    Workspace w;
    int rc = check_ws(m, B, T, d_ws, ws_bytes, &w);
    if (rc) return rc;
    if (which < 0 || which > 2 * m->cfg.n_up) { set_error("tap index out of range"); return BVC_EINVAL; }
    const float *tap = nullptr;
    int64_t len = 0;
    int ch = 0;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = run_vocoder(m, w, d_mel, B, T, 1, 1.0f, nullptr, which, &tap, &len, &ch, s))) return rc;
    const long long n = (long long)B * len * ch;
    if (out_numel_per_batch) *out_numel_per_batch = len * ch;
    if (d_out) {
        hipLaunchKernelGGL(tap_copy_kernel, dim3(1024), dim3(256), 0, s, tap, d_out, n);
        BVC_HIP_TRY(hipGetLastError());
    }
    return BVC_OK;
}

}  // extern "C"
