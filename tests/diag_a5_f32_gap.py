"""Manual (GPU box): where does the f32 path of a5_full leave the reference on the worst of the 512 logits512 samples?
Compares stage tensors of the HIP f32 path with the CPU oracle's on that sample (connectivity, synchrony tokens, spectrogram
tokens, temporal tokens, encoder output, cross-attention output, logits).  Usage: python tests/diag_a5_f32_gap.py [out.json]"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from eyegaze_multimodal_amd.data import randn_windows  # noqa: E402
from oracle import dual_eeg_oracle as O  # noqa: E402
from tests.helpers import GOLDEN  # noqa: E402
from tests.test_gpu_model import DEV, build, relerr  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03_a5_f32_gap.json"
    z = np.load(GOLDEN / "logits512.npz", allow_pickle=False)
    n, seed = int(z["n"]), int(z["seed"])
    _, kw, cfg, sd, model = build("a5_full", "f32")
    model.eval()
    x1, x2, _ = randn_windows(n, cfg.in_channels, 1024, seed=seed, num_classes=cfg.num_classes)
    with torch.no_grad():
        got = torch.cat([model(x1[i:i + 128].to(DEV), x2[i:i + 128].to(DEV))["logits"].float().cpu() for i in range(0, n, 128)]).numpy()
    err = np.abs(got - z["a5_full/logits"]).max(-1)
    order = np.argsort(-err)
    res = {"err_quantiles": {q: float(np.quantile(err, q)) for q in (0.5, 0.9, 0.99, 1.0)}, "worst_samples": [int(i) for i in order[:8]],
           "worst_errs": [float(err[i]) for i in order[:8]], "stages": []}
    for s in order[:3]:
        sl = slice(int(s), int(s) + 1)
        xa, xb = x1[sl].repeat(4, 1, 1), x2[sl].repeat(4, 1, 1)          # (a batch of 4 copies: any engine shape works)
        with torch.no_grad():
            st = {}
            ref = O.forward(xa, xb, sd, cfg, stages=st)
            out = model(xa.to(DEV), xb.to(DEV))
        eng = model.engine(4, 1024, torch.device(DEV))
        S, d, nib, C = eng.S, cfg.d_model, eng.n_ibs, cfg.in_channels
        x0 = eng.a["x0"].float().cpu().view(eng.NB, S, d)
        pos = sd["pos_embed.pos_embed.weight"][:S]
        row = {"sample": int(s), "dlogit": float((out["logits"].cpu() - ref["logits"]).abs().max()),
               "connectivity_max_abs": float((eng.a["ib_conn"].cpu()[:1][:, :, cfg.feature_indices] - st["connectivity"][:1]).abs().max()),
               "pli_entries_moved": int(((eng.a["ib_conn"].cpu()[:1][:, :, 1] - st["connectivity"][:1][:, :, 1]).abs() > 1e-4).sum()),
               "ibs_tokens_relerr": relerr(x0[:1, 1:1 + nib] - pos[1:1 + nib], st["ibs_tokens"][:1]),
               "spec_tokens_relerr": relerr(x0[:1, 1 + nib:1 + nib + C] - pos[1 + nib:1 + nib + C], st["spec1"][:1]),
               "temporal_tokens_relerr": relerr(x0[:1, 1 + nib + C:] - pos[1 + nib + C:], st["h1"][:1]),
               "encoder_out_relerr": relerr(eng.a["zn"].float().cpu().view(eng.NB, S, d)[:1], st["z1"][:1]),
               "cross_out_relerr": relerr(eng.a["zc"].float().cpu().view(eng.NB, S, d)[:1], st["z1c"][:1]),
               "ibs_tokens_max_abs_ref": float(st["ibs_tokens"][:1].abs().max()), "ibs_tokens_max_abs_err": float(((x0[:1, 1:1 + nib] - pos[1:1 + nib]) - st["ibs_tokens"][:1]).abs().max())}
        conn_h = eng.a["ib_conn"].cpu()[:1][:, :, cfg.feature_indices]
        d_ = (conn_h - st["connectivity"][:1]).abs()
        row["connectivity_max_abs_by_feature"] = [float(d_[:, :, j].max()) for j in range(d_.shape[2])]
        # token-axis variance of every matrix entry (what InstanceNorm1d divides by): tiny variances amplify rounding differences
        xx = st["connectivity"][:1].reshape(1, -1, C * C)
        row["min_token_axis_var"] = float(xx.var(1, unbiased=False).min())
        res["stages"].append(row)
        print(json.dumps(row), flush=True)
    Path(out_path).write_text(json.dumps(res, indent=1))
    print(json.dumps({k: v for k, v in res.items() if k != "stages"}))


if __name__ == "__main__":
    main()
