"""The schedule bench.py reports as `multi_stream` - independent batches issued round-robin on several HIP
streams through ONE model (shared weights, per-stream workspaces, one persistent recurrence at a time) - must
give bit for bit what the same calls give one after the other on one stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _run(model, xs, L, streams, rounds):
    outs = []
    for k in range(rounds * len(xs)):
        st = streams[k % len(streams)]
        with torch.cuda.stream(st):
            x = xs[k % len(xs)]
            codes = model.encode(x, 3000)
            outs.append((codes, model.decode(codes, L)))
    for st in streams:
        torch.cuda.current_stream(DEV).wait_stream(st)
    torch.cuda.synchronize(DEV)
    return outs


@pytest.mark.parametrize("recurrence", ["persistent", "layers", "auto"])
def test_four_stream_schedule_equals_serial(recurrence):
    """Every schedule - the persistent kernel, one launch per layer, and 'auto' (the default: persistent for serial calls, launch per
    layer once calls of several streams overlap) - sums every output in ONE order (k_gemm.hip), so codes and waveforms of overlapping
    calls are bit for bit those of the serial run, AND those of the persistent kernel, whichever kernels a call happened to get."""
    from gpu_common import make_model
    from bvcodec import dist as bdist, synth
    env = {"BVC_RECURRENCE": "layers"} if recurrence == "layers" else None
    model, conf, _, _ = make_model(True, 1024, env=env)
    canon = make_model(True, 1024)[0]
    if recurrence != "layers":
        model.set_recurrence(recurrence)
    B, L = 64, int(22050 * 1.2)
    xs = [synth.synthetic_speech(B, L, seed=100 + i, kind="noise" if i % 2 else "speech").to(DEV) for i in range(4)]
    try:
        serial = _run(model, xs, L, [torch.cuda.current_stream(DEV)], 1)
        streams = bdist.concurrent_stream_sets(4, DEV)[0]           # what bench.py uses
        assert len({s.cuda_stream for s in streams}) == 4
        conc = _run(model, xs, L, streams, 3)
        for k, (codes, wav) in enumerate(conc):
            ref_codes, ref_wav = serial[k % 4]
            assert torch.equal(codes, ref_codes), f"codes of call {k} differ under the {len(streams)}-stream schedule"
            assert torch.equal(wav, ref_wav), f"waveform of call {k} differs under the {len(streams)}-stream schedule"
        torch.cuda.synchronize(DEV)
        model.check_status()
        if recurrence == "auto":                               # ... and one at a time it is the persistent kernel again
            again = _run(model, xs, L, [torch.cuda.current_stream(DEV)], 2)
            for k in range(4, 8):
                assert torch.equal(again[k][0], serial[k % 4][0]) and torch.equal(again[k][1], serial[k % 4][1])
        # the persistent kernel's results, whatever this run's schedule was
        canon.set_recurrence("persistent")
        ref = _run(canon, xs, L, [torch.cuda.current_stream(DEV)], 1)
        for k in range(4):
            assert torch.equal(ref[k][0], serial[k][0]) and torch.equal(ref[k][1], serial[k][1]), k
    finally:
        canon.set_recurrence("auto")
        if recurrence != "layers":
            model.set_recurrence("auto")


def test_two_models_on_two_host_threads():
    """include/bvcodec.h: models are independent.  Two models (different weights), each driven by its
    own host thread on its own stream (ctypes releases the GIL inside a call, so the library calls really overlap on the host):
    the persistent launches of the two go through the process-wide ticket one after the other, everything else overlaps; both
    threads must get bit for bit what the same calls give from one thread."""
    import threading
    from gpu_common import make_model
    from bvcodec import synth
    models = [make_model(True, 1024, seed=1234)[0], make_model(True, 1024, seed=4321)[0]]
    B, L, rounds = 48, int(22050 * 0.9), 4
    xs = [synth.synthetic_speech(B, L, seed=200 + i, kind="speech").to(DEV) for i in range(2)]
    try:
        for m in models:
            m.set_recurrence("persistent")
        ref = []
        for m, x in zip(models, xs):
            codes = m.encode(x, 3000)
            ref.append((codes, m.decode(codes, L)))
        torch.cuda.synchronize(DEV)
        streams = [torch.cuda.Stream(DEV) for _ in models]
        results, errors = [[] for _ in models], []

        def worker(i):
            try:
                with torch.cuda.stream(streams[i]):
                    for _ in range(rounds):
                        codes = models[i].encode(xs[i], 3000)
                        results[i].append((codes, models[i].decode(codes, L)))
                streams[i].synchronize()
            except Exception as e:                               # noqa: BLE001 - reported by the assert below
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        torch.cuda.synchronize(DEV)
        assert not errors, errors
        for i in range(2):
            assert len(results[i]) == rounds
            for codes, wav in results[i]:
                assert torch.equal(codes, ref[i][0]) and torch.equal(wav, ref[i][1])
            models[i].check_status()
    finally:
        for m in models:
            m.set_recurrence("auto")


def test_many_workspaces_on_one_model():
    """More streams (hence workspaces) than the launch-per-layer schedule's old 16-entry graph cache held."""
    from gpu_common import make_model
    from bvcodec import synth
    model, conf, _, _ = make_model(True, 1024, env={"BVC_RECURRENCE": "layers"})
    B, L = 8, 22050 // 2
    x = synth.synthetic_speech(B, L, seed=7, kind="speech").to(DEV)
    ref = _run(model, [x], L, [torch.cuda.current_stream(DEV)], 1)[0]
    streams = [torch.cuda.Stream(DEV) for _ in range(12)]
    for codes, wav in _run(model, [x], L, streams, 24):
        assert torch.equal(codes, ref[0]) and torch.equal(wav, ref[1])


def test_recurrence_schedule_can_be_switched_at_run_time():
    """model.set_recurrence('layers' / 'persistent') (bvc_model_set_option): both schedules on one model, same weights, the same
    bits - codes, decoder output, waveform (one order of summation per output: k_gemm.hip)."""
    from gpu_common import make_model
    from bvcodec import synth
    model, conf, _, _ = make_model(True, 1024)
    B, L = 32, 22050
    x = synth.synthetic_speech(B, L, seed=77, kind="speech").to(DEV)
    h0 = torch.zeros(1, B, 1024, device=DEV)
    try:
        model.set_recurrence("persistent")
        codes_p = model.encode(x, 3000)
        wav_p = model.decode(codes_p, L)
        mel_p, hT_p = model.bvrnn.decode(codes_p, h0)
        model.set_recurrence("layers")
        codes_l = model.encode(x, 3000)
        wav_l = model.decode(codes_p, L)
        mel_l, hT_l = model.bvrnn.decode(codes_p, h0)
    finally:
        model.set_recurrence("auto")
    assert torch.equal(codes_l, codes_p)
    assert torch.equal(mel_l, mel_p) and torch.equal(hT_l, hT_p)
    assert torch.equal(wav_l, wav_p)
    assert torch.equal(model.encode(x, 3000), codes_p)       # and back
    model.check_status()


@pytest.mark.parametrize("B,frames", [(80, 9), (130, 5), (256, 3), (512, 2)])
def test_large_batch_on_interleaved_chains_equals_layer_schedule(B, frames):
    """More than 64 utterances exceed one workgroup per (utterance group, feature tile): the persistent kernel then
    works through several utterance groups ("chains") per workgroup (k_flow.hip, MULTI).  Same arithmetic per output as
    the launch-per-layer schedule: identical codes and mel, including a last, partly filled utterance group and a
    workgroup with fewer chains than the others; and against the oracle on a few utterances."""
    from gpu_common import make_model
    from bvcodec import synth
    from oracle import codec as ocodec
    model, conf, sd1, sd2 = make_model(True, 1024)
    L = 256 * frames + 17
    x = synth.synthetic_speech(B, L, seed=B, kind="speech").to(DEV)
    try:
        model.set_recurrence("persistent")
        codes = model.encode(x, 3000)
        h0 = torch.zeros(1, B, 1024, device=DEV)
        mel, hT = model.bvrnn.decode(codes, h0)
        torch.cuda.synchronize(DEV)
        model.check_status()
        model.set_recurrence("layers")
        codes_l = model.encode(x, 3000)
        mel_l, hT_l = model.bvrnn.decode(codes, h0)
    finally:
        model.set_recurrence("auto")
    assert torch.equal(codes, codes_l)
    assert torch.equal(mel, mel_l) and torch.equal(hT, hT_l)
    oc = ocodec.OracleCodec(conf, sd1, sd2)
    pick = [0, 63, 64, B - 1]                              # first / last utterance of groups that sit in different chains
    r = oc.encode(x[pick].cpu(), 3000, full=True)
    mism = codes[pick].cpu() != r["codes"]
    assert not bool((mism & ((r["prob"] - 0.5).abs() > 1e-5)).any())


@pytest.mark.parametrize("recurrence", ["auto", "persistent", "layers"])
def test_one_model_on_two_host_threads(recurrence):
    """include/bvcodec.h: several host threads may issue calls on ONE model at once, each with its own workspace and stream
    (the facade keeps one workspace per stream).  Two threads, different inputs, every schedule: each thread must get bit for
    bit what the same calls give from one thread - also under `auto`, where the two threads' calls see each other as company
    and change schedule on the way (one order of summation: same bits)."""
    import threading
    from gpu_common import make_model
    from bvcodec import synth
    model = make_model(True, 1024, seed=1234)[0]
    B, L, rounds = 40, int(22050 * 0.8), 5
    xs = [synth.synthetic_speech(B, L, seed=300 + i, kind="speech").to(DEV) for i in range(2)]
    try:
        model.set_recurrence("persistent")
        ref = []
        for x in xs:
            codes = model.encode(x, 3000)
            ref.append((codes, model.decode(codes, L)))
        torch.cuda.synchronize(DEV)
        model.set_recurrence(recurrence)
        streams = [torch.cuda.Stream(DEV) for _ in xs]
        results, errors = [[] for _ in xs], []

        def worker(i):
            try:
                with torch.cuda.stream(streams[i]):
                    for _ in range(rounds):
                        codes = model.encode(xs[i], 3000)
                        results[i].append((codes, model.decode(codes, L)))
                streams[i].synchronize()
            except Exception as e:                               # noqa: BLE001 - reported by the assert below
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        torch.cuda.synchronize(DEV)
        assert not errors, errors
        for i in range(2):
            assert len(results[i]) == rounds
            for codes, wav in results[i]:
                assert torch.equal(codes, ref[i][0]) and torch.equal(wav, ref[i][1])
        model.check_status()
    finally:
        model.set_recurrence("auto")
