"""Diagnostic (manual, GPU box): run-to-run spread of the train-mode (dropout 0.1) loss curve — HIP bf16 / f32 with different
dropout seeds, and the CPU oracle with different torch seeds — on the task of tests/parity_training_run.py.
Prints, per run, the first step after which the 8-step mean loss stays below 0.1."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW
from eyegaze_multimodal_amd.data import synth_windows
from oracle import dual_eeg_oracle as O

steps, B, DROP, lr = 300, 32, 0.1, 3e-4
kw = dict(in_channels=8, num_classes=3, max_len=256, dropout=DROP, use_spectrogram=False, use_ibs=False, use_cross_attention=True)
cfg = O.ModelCfg(**kw)
x1, x2, y = synth_windows(steps * B, 8, 1024, 3, seed=11)
torch.manual_seed(42)
sd0 = {k: v.detach().clone() for k, v in DualEEGTransformer(**kw).state_dict().items()}


def settle(losses):
    m = np.convolve(np.array(losses), np.ones(8) / 8, mode="valid")
    bad = np.nonzero(m >= 0.1)[0]
    return int(bad[-1]) + 8 if len(bad) else 0


def hip_run(dtype, seed0):
    model = DualEEGTransformer(**kw, compute_dtype=dtype)
    model.load_state_dict(sd0)
    model = model.cuda().train()
    opt = HipAdamW(model, lr=lr)
    eng = model.engine(B, 1024, torch.device("cuda"))
    one = torch.ones(1, device="cuda")
    out = []
    for i in range(steps):
        j = slice(i * B, (i + 1) * B)
        opt.begin_step(eng, seed=seed0 + i)
        eng.forward(x1[j].cuda(), x2[j].cuda(), y[j].cuda(), train=True)
        eng.backward(gloss=one)
        opt.step(eng)
        out.append(float(eng.a["loss"]))
    return out


def cpu_run(seed):
    torch.manual_seed(seed)
    params = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    state, out = {}, []
    for i in range(steps):
        j = slice(i * B, (i + 1) * B)
        for p in params.values():
            p.grad = None
        o = O.forward(x1[j], x2[j], params, cfg, y[j], train=True)
        o["loss_ce"].backward()
        out.append(float(o["loss_ce"].detach()))
        with torch.no_grad():
            O.clip_and_adamw({k: p.data for k, p in params.items()}, {k: p.grad for k, p in params.items()}, state, step=i + 1, lr=lr)
    return out


torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
for dtype in ("bf16", "f32"):
    for seed0 in (0, 100000, 200000, 300000, 400000, 500000):
        l = hip_run(dtype, seed0)
        print(f"hip {dtype} seed0={seed0:6d}: settles at step {settle(l):3d}  mean(100:200)={np.mean(l[100:200]):.3f}", flush=True)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.time()
    l = cpu_run(seed)
    print(f"cpu f32 torch seed={seed}: settles at step {settle(l):3d}  mean(100:200)={np.mean(l[100:200]):.3f}  ({time.time()-t0:.0f} s)", flush=True)
