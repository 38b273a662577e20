"""Diagnostic (run by hand on the GPU box): per-parameter relative Frobenius error of the HIP gradients against
the CPU oracle's autograd on the same fixture.  python tests/diag_grads.py cfg3_xattn [bf16|f32]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from eyegaze_multimodal_amd import DualEEGTransformer
from oracle import dual_eeg_oracle as O
from tests.helpers import load_golden, t

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_xattn"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
z, kw, cfg, sd = load_golden(name)
model = DualEEGTransformer(**kw, compute_dtype=dtype)
model.load_state_dict(sd)
model = model.cuda().eval()
x1, x2, labels = t(z["randn/eeg1"]), t(z["randn/eeg2"]), t(z["labels"])
out = model(x1.cuda(), x2.cuda(), labels.cuda())
loss = out["loss_ce"] + (out["loss_ibs_cls"] if "loss_ibs_cls" in out else 0)
loss.backward()
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.endswith("window")}
full = dict(params)
full.update({k: v for k, v in sd.items() if k.endswith("window")})
stages = {}
ref = O.forward(x1, x2, full, cfg, labels, stages=stages)
rl = ref["loss_ce"] + (ref["loss_ibs_cls"] if "loss_ibs_cls" in ref else 0)
rl.backward()
print(f"{name} {dtype}: loss hip={float(loss):.6f} ref={float(rl):.6f}  logits err={float((out['logits'].cpu()-ref['logits']).abs().max()):.3e}")
rows = []
for n, p in model.named_parameters():
    g, r = p.grad.cpu().double(), params[n].grad.double()
    rows.append((float((g - r).norm() / r.norm().clamp_min(1e-30)), float(g.norm()), float(r.norm()), n))
for e, gn, rn, n in rows:
    print(f"{e:9.3e}  hip={gn:10.4e} ref={rn:10.4e}  {n}")
