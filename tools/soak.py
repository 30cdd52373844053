"""Soak run on the GPU box: 2000 streaming ticks of 64 streams against the offline call over the whole 40 s (codes bit for bit, waveform),
then 300 headline steps back to back (same bits as the first, no time-out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_common import make_model
from bvcodec import synth
from bvcodec.streaming import StreamingCodec
DEV = torch.device("cuda:0")
model = make_model(True, 1024)[0]
B, hop, hops = 64, 441, 2000
L = hop * hops
x = synth.synthetic_speech(B, L, seed=77, kind="speech").to(DEV)
sc = StreamingCodec(model, B, 3000, hop=hop)
codes, wavs = [], []
t0 = time.time()
for i in range(hops):
    c, w = sc.push(x[:, i * hop:(i + 1) * hop])
    codes.append(c.clone()); wavs.append(w.clone())
torch.cuda.synchronize()
print("ticks", hops, "in", round(time.time() - t0, 2), "s")
codes, wav = torch.cat(codes, 1), torch.cat(wavs, 1)
F = codes.shape[1]
off = model.encode(x, 3000)
print("frames", F, "codes equal", bool(torch.equal(codes, off[:, :F])))
wav_off = model.decode(off, L)
print("max |dwav|", float((wav - wav_off[:, :256 * F]).abs().max()))
model.check_status()
# 300 headline steps back to back
y = synth.synthetic_speech(64, 110250, seed=0, kind="noise").to(DEV)
ref = model.encode(y, 3000); refw = model.decode(ref, 110250)
t0 = time.time()
for i in range(300):
    c = model.encode(y, 3000); w = model.decode(c, 110250)
torch.cuda.synchronize()
print("300 steps", round((time.time() - t0) / 300 * 1e3, 2), "ms/step; last equal first:", bool(torch.equal(c, ref) and torch.equal(w, refw)))
model.check_status()
print("soak ok")
