"""
TEST INFRASTRUCTURE — logits-only fixtures at a sample count that makes "argmax agreement >= 99.5 %" measurable.
Runs ONLY in the build container (the reference is imported by path from /root/reference, as oracle/make_golden.py does).

For each BASELINE.json config (+ the reference's default flags, A5) the reference model (eval mode, CPU fp32, the same
deterministic synthetic weights as every other fixture) classifies N = 512 window pairs.  The inputs are NOT stored: they are
the throughput generator of SURVEY 8d, `randn_windows(N, 8, 1024, seed)` -- torch's CPU generator is reproducible, so the
GPU test regenerates them from the seed.  Stored per config: logits [N, ncls] f32, argmax [N], the top-2 margin [N]; for the
configuration with synchrony matrices also the reference's PLI entries as integer counts [N, 6, C, C] (its sign() decisions).

Usage:  python oracle/make_golden_logits.py
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle.make_golden import CONFIGS, WEIGHT_SEED, load_reference  # noqa: E402
from oracle.dual_eeg_oracle import ModelCfg, synthetic_state_dict  # noqa: E402
from eyegaze_multimodal_amd.data import randn_windows  # noqa: E402  (pure torch CPU helper; nothing of the HIP path runs)

N, SEED, CHUNK = 512, 4242, 64
NAMES = ["cfg1_a1_2class", "cfg2_concat", "cfg3_xattn", "cfg5_a2_spec", "a5_full"]


def main():
    torch.set_num_threads(8)
    model_mod, _, _ = load_reference()
    out = {"n": np.int64(N), "seed": np.int64(SEED)}
    for name in NAMES:
        kw = CONFIGS[name]
        cfg = ModelCfg(**kw)
        model = model_mod.DualEEGTransformer(**kw)
        model.load_state_dict(synthetic_state_dict(cfg, WEIGHT_SEED), strict=True)
        model.eval()
        x1, x2, _ = randn_windows(N, cfg.in_channels, 1024, seed=SEED, num_classes=cfg.num_classes)
        logits = []
        with torch.no_grad():
            for i in range(0, N, CHUNK):
                logits.append(model(x1[i:i + CHUNK], x2[i:i + CHUNK])["logits"])
        lg = torch.cat(logits).numpy().astype(np.float32)
        top = np.sort(lg, axis=-1)
        out[name + "/logits"] = lg
        out[name + "/argmax"] = lg.argmax(-1).astype(np.int64)
        out[name + "/margin"] = (top[:, -1] - top[:, -2]).astype(np.float32)
        if getattr(model, "ibs_matrix_generator", None) is not None and cfg.use_robust_ibs:
            # the reference's sign() decisions, compactly: PLI = |mean_t sign(dphi_t)| (D:613-630) is |k| / T with integer k, so
            # round(PLI * T) per (sample, band, channel pair) is exact in uint16.  The GPU test uses it to show that the samples
            # whose logits differ by more than the plain f32 gate are exactly those where a PLI entry differs.
            with torch.no_grad():
                conn = torch.cat([model.ibs_matrix_generator(x1[i:i + CHUNK], x2[i:i + CHUNK]) for i in range(0, N, CHUNK)])
            out[name + "/pli_counts"] = np.rint(conn[:, :, 1].numpy().astype(np.float64) * 1024).astype(np.uint16)
        print(name, lg.shape, "margin<4e-2:", int((out[name + "/margin"] < 4e-2).sum()), "class counts", np.bincount(lg.argmax(-1)))
    np.savez_compressed(REPO / "tests" / "golden" / "logits512.npz", **out)


if __name__ == "__main__":
    main()
