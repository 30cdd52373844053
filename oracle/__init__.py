"""CPU oracle for the BVRNNCodecModel encode/decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it, and there only as the checker / the timed CPU baseline.  The product
package (``bernoulli-var-speech-codec_amd`` a.k.a. ``bvcodec``) never imports it and
fails loudly when its HIP library is missing.

What it is: an op-for-op restatement, on PyTorch-CPU / numpy, of the reference's
algorithm for the path (citations are relative to the reference checkout):

* ``melbank.py``   - ``librosa.filters.mel`` (Slaney scale + Slaney norm) as called at
                     ``third_party/BigVGAN/meldataset.py:68`` and ``torch.hann_window``
                     (``meldataset.py:70``).  librosa is a third-party dependency that is
                     NOT vendored in the reference (pinned ``librosa==0.8.1`` in
                     ``third_party/BigVGAN/requirements.txt:3``) and not installed in the
                     image: its published algorithm is restated; parity at this one
                     boundary is pinned against the independent implementation
                     ``transformers.audio_utils.mel_filter_bank`` (captured by
                     ``tests/golden/make_melbank_pin.py``, checked in ``tests/test_oracle_golden.py``), i.e. "parity unpinned" w.r.t.
                     librosa itself.
* ``frontend.py``  - ``mel_spectrogram``  (``meldataset.py:60-95``, ``:38-39``).
* ``bvrnn.py``     - ``BVRNN.encode`` / ``BVRNN.decode`` (``bvrnn.py:163-229``, nets
                     ``bvrnn.py:44-83``), GRU cell equations of ``torch.nn.GRU``; ``BVRNN.forward``
                     (``bvrnn.py:86-160``: sampler, prior, KLD) with the random numbers as inputs.
* ``bigvgan.py``   - ``BigVGAN.forward`` (``third_party/BigVGAN/models.py:207-238``),
                     ``AMPBlock1.forward`` (``models.py:103-121``), ``SnakeBeta``
                     (``third_party/BigVGAN/activations.py:107-120``), weight-norm fold.
* ``codec.py``     - ``BVRNNCodecModel.encode/decode/forward``
                     (``bvrnn_codec_model.py:44-76``).

How it is pinned: ``tests/golden/*.npz`` were produced by importing the reference itself
in the build container (``tests/golden/make_golden.py`` and ``make_golden_forward.py``, committed) on seeded synthetic
checkpoints; ``tests/test_oracle_golden.py`` checks every oracle stage against them.
The reference has no tests, golden vectors or known-answer fixtures of its own
(SURVEY.md section 4), and its trained checkpoints are Git-LFS pointers.

Every function takes a ``dtype`` (float32 mirrors the reference; float64 gives a
high-precision "truth" used to report which side of a near-tie a code bit lies on).
"""
