#!/usr/bin/env python3
"""Experiment: how much of the vocoder hides under the recurrent chains if it runs on its own stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_common import make_model
from bvcodec import synth
from bvcodec.model import SCALING
model = make_model()[0]
dev = torch.device("cuda:0")
x = synth.synthetic_speech(64, 110250, seed=0, kind="noise").to(dev)
h0 = torch.zeros(1, 64, 1024, device=dev)

def run(nchain, nvoc, steps=12, prio=False):
    chains = [torch.cuda.Stream(dev, priority=(-1 if prio else 0)) for _ in range(nchain)]
    vocs = [torch.cuda.Stream(dev) for _ in range(nvoc)]
    def go(n):
        for k in range(n):
            s = chains[k % nchain]
            with torch.cuda.stream(s):
                codes = model.encode(x, 3000)
                mel, _ = model.bvrnn.decode(codes, h0)
                ev = torch.cuda.Event(); ev.record(s)
            v = vocs[k % nvoc] if nvoc else s
            with torch.cuda.stream(v):
                v.wait_event(ev)
                wav = model.vocoder(mel, 110250, _scale_div=SCALING, _time_major=True)
        torch.cuda.synchronize()
    go(4)
    t0 = time.perf_counter(); go(steps); dt = time.perf_counter() - t0
    print(f"chain streams {nchain}, vocoder streams {nvoc}, prio {prio}: {1e3*dt/steps:.1f} ms/step -> {64*5*steps/dt:.0f} x RT", flush=True)

run(1, 0); run(2, 0); run(1, 1); run(2, 1); run(2, 2); run(3, 1); run(2, 1, prio=True)
