"""Diagnostic (manual, GPU box): does the structure of the per-step dropout seed matter?  HIP only, many runs.
mode seq: seed = seed0 + step (what the trainer passes); mode mix: seed = splitmix64(seed0 + step)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW
from eyegaze_multimodal_amd.data import synth_windows

steps, B, DROP, lr = 300, 32, float(sys.argv[2]) if len(sys.argv) > 2 else 0.1, 3e-4
kw = dict(in_channels=8, num_classes=3, max_len=256, dropout=DROP, use_spectrogram=False, use_ibs=False, use_cross_attention=True)
x1, x2, y = synth_windows(steps * B, 8, 1024, 3, seed=11)
x1, x2, y = x1.cuda(), x2.cuda(), y.cuda()
torch.manual_seed(42)
sd0 = {k: v.detach().clone() for k, v in DualEEGTransformer(**kw).state_dict().items()}
M64 = (1 << 64) - 1


def splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def settle(losses):
    m = np.convolve(np.array(losses), np.ones(8) / 8, mode="valid")
    bad = np.nonzero(m >= 0.1)[0]
    return int(bad[-1]) + 8 if len(bad) else 0


def run(seed0, mix, dtype="bf16"):
    model = DualEEGTransformer(**kw, compute_dtype=dtype)
    model.load_state_dict(sd0)
    model = model.cuda().train()
    opt = HipAdamW(model, lr=lr)
    eng = model.engine(B, 1024, torch.device("cuda"))
    one = torch.ones(1, device="cuda")
    out = []
    for i in range(steps):
        j = slice(i * B, (i + 1) * B)
        s = seed0 + i
        opt.begin_step(eng, seed=(splitmix64(s) & 0x7FFFFFFFFFFFFFFF) if mix else s)
        eng.forward(x1[j], x2[j], y[j], train=True)
        eng.backward(gloss=one)
        opt.step(eng)
        out.append(eng.a["loss"].clone())
    return [float(v) for v in torch.stack(out).cpu()]


n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for mix in (False, True):
    res = [settle(run(1000003 * k + 17, mix)) for k in range(n)]
    print("mix" if mix else "seq", "settle steps:", sorted(res), " stuck(>=292):", sum(r >= 292 for r in res), flush=True)
