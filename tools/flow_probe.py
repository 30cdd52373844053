"""Per-layer time of the persistent recurrence as workgroup 0 sees it (entry of the layer -> its output published).
    python tools/flow_probe.py [seconds [batch]]"""
import ctypes
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bvcodec import BVRNNCodecModel, _abi, config, synth   # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
conf = config.load_config(config.DEFAULT_CONFIG)
d = tempfile.mkdtemp()
p1, p2 = synth.write_checkpoints(conf, d, seed=1234)
model = BVRNNCodecModel(config.DEFAULT_CONFIG, p1, p2).to("cuda:0")
L = int(22050 * secs)
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = synth.synthetic_speech(BATCH, L, seed=0, kind="noise").cuda()
lib = _abi.load()
for _ in range(2):
    codes = model.encode(x, 3000)
    model.decode(codes, L)
torch.cuda.synchronize()


def kread(lo, hi):
    mean, mn, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
    _abi.check(lib.bvc_kprobe_read(lo, hi, ctypes.byref(mean), ctypes.byref(mn), ctypes.byref(n)))
    return mean.value, mn.value, n.value


_abi.check(lib.bvc_kprobe_enable(1))
codes = model.encode(x, 3000)
names = ["enc0", "enc1", "enc2", "pz0", "pz1", "pz2", "dec0", "dec1", "dec2", "dec3", "px0", "px1", "px2", "gru"]
tot = 0.0
for i, nm in enumerate(names):
    m, mn, n = kread(i, i + 1)
    tot += m
    print(f"encode {nm:5s} mean {m:6.2f} us  min {mn:6.2f}  (n={n})")
print(f"encode sum of layers {tot:.1f} us per frame")
model.decode(codes, L)
tot = 0.0
for i, nm in enumerate(names[6:]):
    m, mn, n = kread(i, i + 1)
    tot += m
    print(f"decode {nm:5s} mean {m:6.2f} us  min {mn:6.2f}  (n={n})")
print(f"decode sum of layers {tot:.1f} us per frame")
_abi.check(lib.bvc_kprobe_enable(0))
model.check_status()
