"""Manual experiment (GPU box): val-accuracy / loss-curve parity of the HIP path against the CPU oracle trained on
the SAME data, SAME initial weights, SAME batch order, dropout 0 (so both are deterministic).
    python tests/parity_training_run.py [steps] [batch] [dropout]  -> gpurun_out/parity_train.json
With dropout > 0 both sides run in train mode (HIP: counter-hash masks, oracle: torch masks): the masks differ, so the
comparison is statistical (loss level, validation accuracy), not step by step."""
import json
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
DROP = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
TRAIN = DROP > 0
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
kw = dict(in_channels=8, num_classes=3, max_len=256, dropout=DROP, use_spectrogram=False, use_ibs=False, use_cross_attention=True)
cfg = O.ModelCfg(**kw)
torch.manual_seed(42)
model = DualEEGTransformer(**kw, compute_dtype=os.environ.get("EYEGAZE_DTYPE", "bf16"))
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.cuda().train()
ntrain, nval = steps * B, 192
x1, x2, y = synth_windows(ntrain + nval, 8, 1024, 3, seed=11)
tr = slice(0, ntrain)
va = slice(ntrain, ntrain + nval)
lr = 3e-4
# ---- HIP ----
opt = HipAdamW(model, lr=lr)
eng = model.engine(B, 1024, torch.device("cuda"))
hip_loss = []
one = torch.ones(1, device="cuda")
t0 = time.perf_counter()
for i in range(steps):
    j = slice(i * B, (i + 1) * B)
    opt.begin_step(eng, seed=i)
    # hard-coded 0.1 dropout sites (D:161) stay off too: the engine is driven in eval mode with gradients
    eng.forward(x1[j].cuda(), x2[j].cuda(), y[j].cuda(), train=TRAIN)
    eng.backward(gloss=one)
    opt.step(eng)
    hip_loss.append(float(eng.a["loss"]))
t_hip = time.perf_counter() - t0
model.eval()
with torch.no_grad():
    pred_h = torch.cat([model(x1[va][k:k + 64].cuda(), x2[va][k:k + 64].cuda())["logits"].argmax(-1).cpu() for k in range(0, nval, 64)])
acc_h = float((pred_h == y[va]).float().mean())
# ---- CPU oracle ----
params = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
state = {}
cpu_loss = []
t0 = time.perf_counter()
for i in range(steps):
    j = slice(i * B, (i + 1) * B)
    for p in params.values():
        p.grad = None
    out = O.forward(x1[j], x2[j], params, cfg, y[j], train=TRAIN)
    out["loss_ce"].backward()
    cpu_loss.append(float(out["loss_ce"]))
    with torch.no_grad():
        O.clip_and_adamw({k: p.data for k, p in params.items()}, {k: p.grad for k, p in params.items()}, state, step=i + 1, lr=lr)
t_cpu = time.perf_counter() - t0
with torch.no_grad():
    pred_c = O.forward(x1[va], x2[va], {k: v.detach() for k, v in params.items()}, cfg)["logits"].argmax(-1)
acc_c = float((pred_c == y[va]).float().mean())
res = {"steps": steps, "batch": B, "dropout": DROP, "hip_val_acc": acc_h, "cpu_val_acc": acc_c, "agree": float((pred_h == pred_c).float().mean()),
       "hip_loss": hip_loss, "cpu_loss": cpu_loss, "hip_seconds": t_hip, "cpu_seconds": t_cpu,
       "max_abs_loss_gap": float(np.max(np.abs(np.array(hip_loss) - np.array(cpu_loss))))}
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/parity_train%s.json" % ("_dropout" if TRAIN else "")).write_text(json.dumps(res))
print(json.dumps({k: v for k, v in res.items() if "loss" not in k or k == "max_abs_loss_gap"}))
print("hip", [round(v, 3) for v in hip_loss[::4]])
print("cpu", [round(v, 3) for v in cpu_loss[::4]])
