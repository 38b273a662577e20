"""Diagnostic (manual, GPU box): is there throughput to gain from running the two Siamese halves out of phase?
Two independent half-batch engines on two HIP streams vs one full-batch engine on one stream."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW
from eyegaze_multimodal_amd.data import randn_windows

dev = torch.device("cuda")
kw = dict(in_channels=8, num_classes=3, max_len=256, use_spectrogram=False, use_ibs=False, use_cross_attention=False)


def make(B):
    torch.manual_seed(42)
    m = DualEEGTransformer(**kw).to(dev).train()
    eng = m.engine(B, 1024, dev)
    opt = HipAdamW(m, lr=1e-4, weight_decay=0.01)
    x1, x2, y = randn_windows(B, 8, 1024, seed=1, device=dev)
    return m, eng, opt, (x1, x2, y)


def step(eng, opt, data, i, one):
    opt.begin_step(eng, seed=100 + i)
    eng.forward(*data, train=True)
    eng.backward(gloss=one)
    opt.step(eng)


one = torch.ones(1, device=dev)
full = make(256)
for i in range(5):
    step(full[1], full[2], full[3], i, one)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    step(full[1], full[2], full[3], i, one)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / 30
print(f"one engine  B=256          : {t_full*1e3:.3f} ms/step  {256/t_full:.0f} samples/s")

ha, hb = make(128), make(128)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for i in range(5):
    for h, s in ((ha, sa), (hb, sb)):
        with torch.cuda.stream(s):
            step(h[1], h[2], h[3], i, one)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    for h, s in ((ha, sa), (hb, sb)):
        with torch.cuda.stream(s):
            step(h[1], h[2], h[3], i, one)
torch.cuda.synchronize()
t_two = (time.perf_counter() - t0) / 30
print(f"two engines B=128 x 2 streams: {t_two*1e3:.3f} ms/step  {256/t_two:.0f} samples/s")
t0 = time.perf_counter()
for i in range(30):
    step(ha[1], ha[2], ha[3], i, one)
torch.cuda.synchronize()
t_half = (time.perf_counter() - t0) / 30
print(f"one engine  B=128          : {t_half*1e3:.3f} ms/step  {128/t_half:.0f} samples/s")
