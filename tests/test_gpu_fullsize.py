"""Code parity at the sizes that are benchmarked (BASELINE configs[1]: 64 x 5 s; configs[3] shard: 64 x 10 s),
against the CPU oracle, through the C ABI.  Goldens pin B=2, T<=43 with inputs chosen away from ties; here the
input is whatever the synthetic generator gives, ~1e6 (2e6) active bits, so some bits DO sit within rounding
noise of a tie: the test states how many utterances diverge, where, and at what margin."""
import numpy as np
import pytest
import torch

from parity_stats import divergence_stats, teacher_forced_mismatches

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seconds,kind,seed", [(5.0, "noise", 11), (10.0, "speech", 12)])
def test_code_divergence_vs_oracle_at_benchmark_size(seconds, kind, seed):
    from gpu_common import make_model
    from bvcodec import synth
    from oracle import bvrnn as obv
    model, conf, vr, _ = make_model(True, 1024)
    torch.set_num_threads(16)
    B, L, bitrate = 64, int(22050 * seconds), 3000
    x = synth.synthetic_speech(B, L, seed=seed, kind=kind)
    nb = int(model.bits_per_frame(bitrate))
    # the recurrence alone: both sides start from the SAME log-mel (the front-end has its own tests)
    mel = model.mel_spectrogram(x.to(DEV))
    T = mel.shape[1]
    bits = torch.full((B, T), float(nb))
    h0 = torch.zeros(1, B, conf["h_dim"])
    codes, all_h, prob = model.bvrnn.encode(mel, bits.to(DEV), h0.to(DEV), return_prob=True)
    codes, all_h = codes.cpu(), all_h.cpu()
    ref = obv.encode(vr, mel.cpu(), bits, h0[0], var_bit=True)
    st = divergence_stats(codes, ref["codes"], ref["prob"], nb)
    print(f"\n{B} x {seconds:g} s ({T} frames, {nb} active bits/frame, {B * T * nb} bits): "
          f"{st['diverged_utterances']} utterances diverge from the oracle; first divergent frames "
          f"{st['first_divergent_frames']}; oracle |p-0.5| there {['%.1e' % m for m in st['first_divergence_margins']]}; "
          f"bits within 1e-6 / 1e-5 of a tie: {st['bits_within_1e-6_of_a_tie']} / {st['bits_within_1e-5_of_a_tie']}")
    assert st["max_first_divergence_margin"] < 1e-5, st
    # the facade (own front-end) gives the same codes as the operator fed with its mel
    assert torch.equal(model.encode(x.to(DEV), bitrate).cpu(), codes)
    # every frame, not just the prefix: restart the oracle from the HIP state at each frame
    forced = obv.encode(vr, mel.cpu(), bits, h0[0], var_bit=True, forced_h=all_h)
    n, worst = teacher_forced_mismatches(codes, forced["codes"], forced["prob"], nb)
    print(f"teacher-forced: {n} differing bits of {B * T * nb}, largest oracle |p-0.5| among them {worst:.1e}; "
          f"max |prob - oracle prob| {float((prob.cpu() - forced['prob']).abs().max()):.2e}")
    assert worst < 1e-5
    assert float((prob.cpu() - forced["prob"]).abs().max()) < 5e-6
    # decode of the HIP codes: mel and final state against the oracle decoding the same codes
    mel_hat, hT = model.bvrnn.decode(codes.to(DEV), h0.to(DEV))
    dref = obv.decode(vr, codes, h0[0])
    assert float((mel_hat.cpu() - dref["mel"]).abs().max()) < 2e-4
    assert float((hT.cpu()[0] - dref["h_last"]).abs().max()) < 2e-5
    model.check_status()
