"""
TEST INFRASTRUCTURE — golden vectors for the analysis-hook contract (SURVEY.md §8b "attribute names other code hooks").
Runs ONLY in the build container (reference mounted at /root/reference).  It registers, on the REFERENCE model, the same
forward hooks the reference's analysis code uses (5_Metrics/eeg_metrics.py:195-205 capture, :335-343 band masking,
:433-452 attention probabilities) and stores what they see / cause.  Data only.

Usage:  python oracle/make_golden_hooks.py     -> tests/golden/hooks.npz
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


def build(model_mod, name):
    kw = CONFIGS[name]
    cfg = ModelCfg(**kw)
    model = model_mod.DualEEGTransformer(**kw)
    model.load_state_dict(synthetic_state_dict(cfg, WEIGHT_SEED), strict=True)
    return model.eval(), cfg


def main():
    torch.set_num_threads(8)
    model_mod, gen_mod, _ = load_reference()
    blob = {}
    # --- cross-attention probabilities through a forward hook on cross_attn.cross_attn.dropout --------------------
    model, cfg = build(model_mod, "cfg3_xattn")
    x1, x2 = make_inputs(gen_mod, "gen_eeg", cfg.in_channels)
    seen = []
    h = model.cross_attn.cross_attn.dropout.register_forward_hook(lambda m, i, o: seen.append(i[0].detach().clone()))
    with torch.no_grad():
        out = model(x1, x2)
    h.remove()
    assert len(seen) == 2 and seen[0].shape == (x1.shape[0], 8, 65, 65)
    blob["xattn_probs"] = torch.stack(seen)[:, :2].numpy()          # [direction, first 2 samples, H, S, S]
    blob["xattn_logits"] = out["logits"].numpy()
    # layer-0 self-attention probabilities (same hook on the encoder's first block; called once per stream)
    seen = []
    h = model.encoder.layers[0].mha.dropout.register_forward_hook(lambda m, i, o: seen.append(i[0].detach().clone()))
    with torch.no_grad():
        model(x1, x2)
    h.remove()
    assert len(seen) == 2
    blob["self0_probs"] = torch.stack(seen)[:, :1].numpy()
    # --- IBS matrices: capture hook and band-masking hook -------------------------------------------------------------
    model, cfg = build(model_mod, "a5_full")
    x1, x2 = make_inputs(gen_mod, "gen_eeg", cfg.in_channels)
    seen = []
    h = model.ibs_matrix_generator.register_forward_hook(lambda m, i, o: seen.append(o.detach().clone()))
    with torch.no_grad():
        base = model(x1, x2)
        direct = model.ibs_matrix_generator(x1, x2)
    h.remove()
    assert len(seen) == 2 and seen[0].shape == (x1.shape[0], 6, 7, 8, 8)
    blob["ibs_conn"] = seen[0].numpy()
    blob["ibs_direct_equal"] = np.array(bool(torch.equal(direct, seen[0])))
    blob["ibs_base_logits"] = base["logits"].numpy()
    masked = []
    for band in range(6):
        def mask(m, i, o, band=band):
            o[:, band] = 0
            return o
        h = model.ibs_matrix_generator.register_forward_hook(mask)
        with torch.no_grad():
            masked.append(model(x1, x2)["logits"].numpy())
        h.remove()
    blob["ibs_masked_logits"] = np.stack(masked)                       # [band, B, ncls]
    # a hook that returns a REPLACEMENT tensor (forward hooks may do that)
    h = model.ibs_matrix_generator.register_forward_hook(lambda m, i, o: o * 0.5)
    with torch.no_grad():
        blob["ibs_halved_logits"] = model(x1, x2)["logits"].numpy()
    h.remove()
    out = REPO / "tests" / "golden" / "hooks.npz"
    np.savez_compressed(out, **blob)
    print("wrote", out, {k: v.shape for k, v in blob.items()}, out.stat().st_size)


if __name__ == "__main__":
    main()
