"""How far do the codes of the HIP path run with the CPU oracle's?  (shared by the GPU tests and bench.py)

A code bit is round(sigmoid(logit)); the two implementations sum in different orders (~1e-7 relative), so a
bit may legitimately differ only where the oracle's own probability is within rounding noise of 0.5 - and from
such a frame on the utterance is desynchronised (different codes -> different state).  So per utterance: the
FIRST frame with a differing bit, and the oracle's |p - 0.5| at the differing bits of that frame."""
import numpy as np


def divergence_stats(codes, ref_codes, ref_prob, active_bits=None):
    codes, ref_codes, ref_prob = (np.asarray(a) for a in (codes, ref_codes, ref_prob))
    B, T, Z = codes.shape
    nb = Z if active_bits is None else int(active_bits)
    mism = codes[:, :, :nb] != ref_codes[:, :, :nb]
    assert np.array_equal(codes[:, :, nb:], ref_codes[:, :, nb:]), "masked positions must be 0.5 in both"
    first_frames, margins = [], []
    for b in range(B):
        fr = np.nonzero(mism[b].any(1))[0]
        if fr.size == 0:
            continue
        t0 = int(fr[0])
        bits = np.nonzero(mism[b, t0])[0]
        first_frames.append(t0)
        margins.append(float(np.abs(ref_prob[b, t0, bits].astype(np.float64) - 0.5).max()))
    margin_all = np.abs(ref_prob[:, :, :nb].astype(np.float64) - 0.5)
    return {
        "utterances": B, "frames": T, "active_bits_per_frame": nb,
        "diverged_utterances": len(margins),
        "first_divergent_frames": first_frames,
        "first_divergence_margins": margins,
        "max_first_divergence_margin": max(margins) if margins else 0.0,
        "bits_within_1e-6_of_a_tie": int((margin_all < 1e-6).sum()),
        "bits_within_1e-5_of_a_tie": int((margin_all < 1e-5).sum()),
        "mismatching_bits_total": int(mism.sum()),
    }


def teacher_forced_mismatches(codes, ref_codes_forced, ref_prob_forced, active_bits=None):
    """With the oracle restarted from the HIP path's own state at every frame, every frame is comparable:
    returns (#differing bits, largest oracle |p - 0.5| among them)."""
    codes, ref, prob = (np.asarray(a) for a in (codes, ref_codes_forced, ref_prob_forced))
    nb = codes.shape[2] if active_bits is None else int(active_bits)
    mism = codes[:, :, :nb] != ref[:, :, :nb]
    if not mism.any():
        return 0, 0.0
    return int(mism.sum()), float(np.abs(prob[:, :, :nb][mism].astype(np.float64) - 0.5).max())
