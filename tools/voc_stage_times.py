#!/usr/bin/env python3
"""Per-stage time of the vocoder at the bench shape (B=64, T=430): cumulative time of bvc_test_vocoder_tap up to
the end of each AMP stage, differenced (used for the tile-shape sweep of amp_pair_kernel: see launch_amp_pair)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_common import make_model
from bvcodec import _abi
model = make_model()[0]
B, T = 64, 430
rng = np.random.default_rng(0)
mel = torch.from_numpy((-4 + 1.6 * rng.standard_normal((B, T, 80))).astype(np.float32)).to("cuda:0")
eng = model.engine(mel)
lib = eng.lib
ws, nws = eng.workspace(B, T)
n = ctypes.c_int64()
def run(which, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _abi.check(lib.bvc_test_vocoder_tap(eng.handle, _abi.ptr(mel), B, T, which, None, ctypes.byref(n), ws, nws, eng.stream()))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
prev = run(1)
out = []
for stage in range(4):
    up = run(1 + 2 * stage) if stage else prev
    amp = run(2 + 2 * stage)
    out.append(amp - up)
    prev = amp
w = model.vocoder(mel.permute(0, 2, 1), 110250)
torch.cuda.synchronize(); t0 = time.perf_counter(); w = model.vocoder(mel.permute(0, 2, 1), 110250); torch.cuda.synchronize()
print("amp stage ms", [round(x, 2) for x in out], "total", round((time.perf_counter() - t0) * 1e3, 2), flush=True)
