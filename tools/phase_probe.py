#!/usr/bin/env python3
"""Per-phase cycle counts of amp_pair_kernel per vocoder stage (needs a library built with -DBVC_PHASE_PROBE;
run with BVC_LIB=.../libbvcodec_probe.so).  Phases: 0 load+S1, 1 conv1 mma, 2 S2->LDS, 3 residual prefetch issue,
4 conv2 mma, 5 epilogue.  Cycles are summed over workgroups (thread 0 of each)."""
import ctypes, os, sys
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
lib.bvc_phase_probe_read.restype = ctypes.c_int
ws, nws = eng.workspace(B, T)
n = ctypes.c_int64()
buf = (ctypes.c_ulonglong * 16)()
prev = np.zeros(16)
for stage in range(4):
    which = 2 + 2 * stage
    for rep in range(2):
        lib.bvc_phase_probe_read(buf, 1)
        _abi.check(lib.bvc_test_vocoder_tap(eng.handle, _abi.ptr(mel), B, T, which, None, ctypes.byref(n), ws, nws, eng.stream()))
        torch.cuda.synchronize()
        lib.bvc_phase_probe_read(buf, 0)
    cur = np.array(list(buf), dtype=np.float64)
    d = cur - prev
    prev = cur
    tot = d[:6].sum()
    print(f"stage {stage}: total {tot/1e9:.3f} Gcyc  " + "  ".join(f"p{i} {100*d[i]/tot:.1f}%" for i in range(6)), flush=True)
