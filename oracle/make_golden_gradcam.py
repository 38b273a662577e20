"""
TEST INFRASTRUCTURE — golden vectors for the Grad-CAM hook contract.  Runs ONLY in the build container.
Registers, on the REFERENCE model, the hooks the reference's GradCAM class uses (5_Metrics/eeg_metrics.py:742-764: a forward
hook and a full backward hook on `spectrogram_generator.spec_conv[3]`), runs the scoring pass of compute_gradcam_spectrogram
(:800-830: parameters frozen, inputs require grad, score = sum of the predicted-class logits) and stores what the hooks saw.
Data only.

Usage:  python oracle/make_golden_gradcam.py     -> tests/golden/gradcam.npz
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle.dual_eeg_oracle import ModelCfg, synthetic_state_dict  # noqa: E402
from oracle.make_golden import CONFIGS, WEIGHT_SEED, load_reference, make_inputs  # noqa: E402


def main():
    torch.set_num_threads(8)
    model_mod, gen_mod, _ = load_reference()
    kw = CONFIGS["cfg5_a2_spec"]
    cfg = ModelCfg(**kw)
    model = model_mod.DualEEGTransformer(**kw)
    model.load_state_dict(synthetic_state_dict(cfg, WEIGHT_SEED), strict=True)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    x1, x2 = make_inputs(gen_mod, "gen_eeg", cfg.in_channels)
    x1, x2 = x1.requires_grad_(True), x2.requires_grad_(True)
    acts, grads = [], []
    layer = model.spectrogram_generator.spec_conv[3]
    h1 = layer.register_forward_hook(lambda m, i, o: acts.append(o.detach().clone()))
    h2 = layer.register_full_backward_hook(lambda m, gi, go: grads.append(go[0].detach().clone()))
    out = model(x1, x2)
    logits = out["logits"]
    pred = logits.argmax(-1)
    score = logits[torch.arange(logits.shape[0]), pred].sum()
    score.backward()
    h1.remove(); h2.remove()
    assert len(acts) == 2 and len(grads) == 2
    C = cfg.in_channels
    keep = slice(0, C)                          # the images of the first window
    blob = {"pred": pred.numpy(), "logits": logits.detach().numpy(),
            "act": torch.stack(acts)[:, keep].numpy(),                 # forward order: stream 1, stream 2
            "grad_in_hook_order": torch.stack(grads)[:, keep].numpy(),  # backward order: stream 2, stream 1
            "shape": np.array(acts[0].shape)}
    w = torch.stack(grads).mean(dim=(3, 4), keepdim=True)
    blob["cam_stream2_then_1"] = torch.relu((w * torch.stack(acts[::-1])).sum(2))[:, keep].numpy()
    out_path = REPO / "tests" / "golden" / "gradcam.npz"
    np.savez_compressed(out_path, **blob)
    print("wrote", out_path, {k: v.shape for k, v in blob.items()}, out_path.stat().st_size)


if __name__ == "__main__":
    main()
