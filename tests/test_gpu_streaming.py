"""Chunked / streaming encode+decode reproduces the offline path (BASELINE configs[4] shape of
processing: 20 ms = 441-sample hops).  Needs the MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model():
    from gpu_common import make_model
    return make_model(True, 1024)[0]


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("hop", [441, 1000, 4096])
def test_streaming_equals_offline(model, hop, incremental):
    from bvcodec import synth
    from bvcodec.streaming import StreamingDecoder, StreamingEncoder
    B, L = 3, 256 * 60 + 123
    x = synth.synthetic_speech(B, L, seed=17, kind="speech").to(DEV)
    codes_off = model.encode(x, 3000)
    wav_off = model.decode(codes_off, L)

    enc = StreamingEncoder(model, B, 3000)
    dec = StreamingDecoder(model, B, incremental=incremental)
    codes, wavs = [], []
    for s in range(0, L, hop):
        c = enc.push(x[:, s:s + hop])
        codes.append(c)
        wavs.append(dec.push(c))
    c = enc.flush()
    codes.append(c)
    wavs.append(dec.push(c))
    codes = torch.cat(codes, 1)
    T = codes.shape[1]
    wavs.append(dec.flush(L - 256 * T))
    wav = torch.cat(wavs, 1)
    assert codes.shape == codes_off.shape and torch.equal(codes, codes_off)          # bit-exact codes
    assert wav.shape == wav_off.shape
    err = (wav - wav_off).abs().max().item()
    assert err <= 1e-6, err                                                             # same arithmetic per sample


def test_incremental_vocoder_equals_offline_for_any_chunking(model):
    """bvc_vocoder_stream_push over ragged chunks == the matching slice of bvc_bigvgan over everything;
    reset() starts a new utterance on the same state."""
    from bvcodec.streaming import VocoderStream
    B, T = 2, 57
    g = torch.Generator().manual_seed(5)
    mel = (torch.randn(B, T, 80, generator=g) * 1.5 - 4.0).to(DEV)
    ref = model.vocoder(mel, 10 ** 12, _time_major=True)[:, 0]
    vs = VocoderStream(model.engine(mel), B, 5)
    for trial in range(2):
        chunks, t = [], 0
        sizes = [1, 5, 2, 1, 1, 4, 3, 5, 5, 5, 5, 5, 5, 5, 5] if trial == 0 else [13, 1, 20, 23]   # >kmax: split inside
        for k in sizes:
            k = min(k, T - t)
            if k == 0:
                break
            chunks.append(vs.push(mel[:, t:t + k].contiguous()))
            t += k
        assert t == T
        wav = torch.cat(chunks, 1)
        assert wav.shape == (B, 256 * T)
        err = (wav - ref[:, :256 * T]).abs().max().item()
        assert err <= 1e-6, (trial, err)
        vs.reset()
    from bvcodec import _abi
    with pytest.raises(RuntimeError):                      # more frames than the state was sized for
        _abi.check(vs.eng.lib.bvc_vocoder_stream_push(vs.handle, _abi.ptr(mel), 6, 1.0, _abi.ptr(ref), vs.eng.stream()))


def test_streaming_hop_latency_is_real_time(model):
    """256 concurrent streams, 20 ms hops: a hop must take well under 20 ms (p50 reported)."""
    import time
    from bvcodec import synth
    from bvcodec.streaming import StreamingDecoder, StreamingEncoder
    B, hop = 256, 441
    x = synth.synthetic_speech(B, hop * 60, seed=3, kind="noise").to(DEV)
    enc, dec = StreamingEncoder(model, B, 3000), StreamingDecoder(model, B)
    lat = []
    for i in range(60):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w = dec.push(enc.push(x[:, i * hop:(i + 1) * hop]))
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    p50 = float(np.median(lat[20:])) * 1e3
    print(f"p50 per-hop latency for 256 streams: {p50:.2f} ms")
    assert torch.isfinite(w).all()
    assert p50 < 20.0


@pytest.mark.parametrize("bitrate,nbytes", [(3000, 5), (1500, 3), (6000, 8), (700, 1)])
def test_wire_format_roundtrip(model, bitrate, nbytes):
    from bvcodec import synth
    x = synth.synthetic_speech(2, 256 * 20 + 5, seed=4, kind="speech").to(DEV)
    codes = model.encode(x, bitrate)
    packed = model.pack(codes, bitrate)
    assert packed.dtype == torch.uint8 and packed.shape == (2, 20, nbytes)
    assert torch.equal(model.unpack(packed, bitrate), codes)
    # reference bit order: LSB first within a byte
    n = model.active_bits(bitrate)
    bits = (codes[:, :, :n].cpu().numpy() > 0.75).astype(np.uint8)
    pad = np.zeros((2, 20, nbytes * 8 - n), np.uint8)
    ref = np.packbits(np.concatenate([bits, pad], 2), axis=2, bitorder="little")
    assert np.array_equal(packed.cpu().numpy(), ref)
    # payload rate: 86.13 frames/s * nbytes * 8 bit
    assert abs(nbytes * 8 * 22050 / 256 - bitrate) < 8 * 86.2 or bitrate > 5512


@pytest.mark.parametrize("fs_in,L", [(24000, 24000 + 17), (16000, 9000), (48000, 20001), (22050, 5000)])
def test_preprocessing_matches_scipy(fs_in, L):
    """example.py:15-17 on the GPU: resample_poly to 22.05 kHz + peak normalisation, vs scipy (float64)."""
    import scipy.signal as ss
    from bvcodec import preprocess
    rng = np.random.default_rng(fs_in)
    x = (0.3 * rng.standard_normal((3, L))).astype(np.float32)
    x[1] = np.sin(2 * np.pi * 440.0 * np.arange(L) / fs_in).astype(np.float32)
    ref = ss.resample_poly(x.astype(np.float64), 22050, fs_in, axis=1)
    got = preprocess.resample_poly(torch.from_numpy(x).to(DEV), 22050, fs_in).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-6
    refn = ref / np.max(np.abs(ref), axis=1, keepdims=True)
    gotn = preprocess.prepare_speech(torch.from_numpy(x).to(DEV), fs_in).cpu().numpy()
    assert np.abs(gotn - refn).max() < 2e-6 and abs(np.abs(gotn).max() - 1.0) < 1e-6


def test_concurrent_stream_picker():
    """bvcodec.dist.concurrent_stream_sets returns up to three sets of three distinct, usable streams."""
    from bvcodec import dist as bdist
    dev = torch.device(DEV)
    sets = bdist.concurrent_stream_sets(3, dev)
    assert 1 <= len(sets) <= 3
    for ss in sets:
        assert len(ss) == 3 and len({s.cuda_stream for s in ss}) == 3
        bufs = [torch.zeros(256, device=dev) for _ in ss]
        for _ in range(20):
            for s_, b in zip(ss, bufs):
                with torch.cuda.stream(s_):
                    b.add_(1.0)
        torch.cuda.synchronize()
        assert all(float(b[0]) == 20.0 for b in bufs)
    assert len(bdist.concurrent_streams(2, dev)) == 2


@pytest.mark.parametrize("B,schedule", [(3, "flow"), (40, "flow"), (256, "flow"),               # 256 streams: BASELINE configs[4]'s own size
                                        (3, "graph"), (40, "graph"), (256, "graph"), (3, "eager")])
def test_whole_hop_codec_equals_offline_and_oracle(model, B, schedule, monkeypatch):
    """bvc_stream_codec_tick (BASELINE configs[4]: 441-sample hops) against the offline path AND, directly, against the CPU
    oracle, on each of its schedules: "flow" (the default: one persistent launch per recurrence), "graph" (launch-per-layer
    kernels, the tick replayed from a hipGraph once warm) and "eager" (the same launches without the graph)."""
    from gpu_common import make_model
    from bvcodec import synth
    from bvcodec.streaming import StreamingCodec
    from oracle import codec as ocodec
    if schedule != "flow":
        monkeypatch.setenv("BVC_STREAM_FLOW", "0")
    if schedule == "eager":
        monkeypatch.setenv("BVC_STREAM_NO_GRAPH", "1")
    _, conf, vr, ge = make_model(True, 1024)
    hop, hops = 441, 120                                            # 2.4 s: ~206 frames, > 170 of them replayed from graphs
    L = hop * hops
    x = synth.synthetic_speech(B, L, seed=23, kind="speech").to(DEV)
    sc = StreamingCodec(model, B, 3000, hop=hop)
    codes, wavs, ks = [], [], []
    for i in range(hops):
        c, w = sc.push(x[:, i * hop:(i + 1) * hop])
        ks.append(c.shape[1])
        codes.append(c.clone())
        wavs.append(w.clone())
    torch.cuda.synchronize()
    codes, wav = torch.cat(codes, 1), torch.cat(wavs, 1)
    F = codes.shape[1]
    assert F == (L - 768) // 256 + 1 and set(ks) <= {0, 1, 2} and wav.shape[1] == 256 * F
    # offline references: the launch-per-layer schedule (the ticks' own recurrence kernels) and the library default (the persistent
    # kernel; phi_x / phi_z on the all-frames GEMMs where a hop runs them frame by frame): one order of summation per output
    # (k_gemm.hip), so the streamed codes are bit for bit the offline ones.  (The hops never emit a frame that needs samples beyond L.)
    ref_model = make_model(True, 1024, env={"BVC_RECURRENCE": "layers"})[0]
    codes_off = ref_model.encode(x, 3000)
    assert torch.equal(codes, codes_off[:, :F])
    assert torch.equal(model.encode(x, 3000)[:, :F], codes)
    wav_off = ref_model.decode(codes_off, L)
    assert (wav - wav_off[:, :256 * F]).abs().max().item() <= 2e-6
    # the oracle directly (2 utterances): free-running codes; waveform of the oracle decoding the streamed codes
    torch.set_num_threads(16)
    oc = ocodec.OracleCodec(conf, vr, ge)
    r = oc.encode(x[:2].cpu(), 3000, full=True)
    mism = codes[:2].cpu() != r["codes"][:, :F]
    margin = (r["prob"][:, :F] - 0.5).abs()
    assert not bool((mism & (margin > 1e-5)).any()), int(mism.sum())
    if not bool(mism.any()):
        ref_wav = oc.decode(r["codes"], L)[:, :256 * F]
        assert float((wav[:2].cpu() - ref_wav).pow(2).mean().sqrt()) < 1e-4
    model.check_status()


@pytest.mark.parametrize("B,hop", [(3, 700), (40, 1100), (5, 300), (17, 1500)])
def test_whole_hop_codec_other_hops_equal_offline(model, B, hop):
    """Hops other than configs[4]'s 441 samples: 1-5 and more frames per tick (three and four frames go through the recurrent-layer
    kernel with the frame in the grid's second dimension, five and more through the batched GEMMs), other window capacities of the
    sliding generator buffers, ticks that complete no frame.  Codes bit for bit the offline ones, waveform within 2e-6."""
    from bvcodec import synth
    from bvcodec.streaming import StreamingCodec
    hops = max(24, 22050 * 2 // hop)
    L = hop * hops
    x = synth.synthetic_speech(B, L, seed=31, kind="speech").to(DEV)
    sc = StreamingCodec(model, B, 3000, hop=hop)
    codes, wavs, ks = [], [], set()
    for i in range(hops):
        c, w = sc.push(x[:, i * hop:(i + 1) * hop])
        ks.add(c.shape[1])
        codes.append(c.clone())
        wavs.append(w.clone())
    torch.cuda.synchronize()
    codes, wav = torch.cat(codes, 1), torch.cat(wavs, 1)
    F = codes.shape[1]
    assert F == (L - 768) // 256 + 1 and wav.shape[1] == 256 * F and len(ks) >= 2
    codes_off = model.encode(x, 3000)
    assert torch.equal(codes, codes_off[:, :F])
    wav_off = model.decode(codes_off, L)
    assert (wav - wav_off[:, :256 * F]).abs().max().item() <= 2e-6
    model.check_status()
