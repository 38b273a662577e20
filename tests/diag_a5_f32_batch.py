"""Manual (GPU box): is the f32 path of a5_full batch independent?  Sample 202 of the logits512 set differed from the reference by
2.2e-5 in a chunk of 128 and by 1.1e-6 in a batch of 4 copies (tests/diag_a5_f32_gap.py).  This runs the chunk twice (determinism)
and compares the stage tensors of that sample between the two batch shapes."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from eyegaze_multimodal_amd.data import randn_windows  # noqa: E402
from tests.helpers import GOLDEN  # noqa: E402
from tests.test_gpu_model import DEV, build  # noqa: E402


def stages(eng, row):
    a, S, d = eng.a, eng.S, eng.cfg.d_model
    out = {"ib_conn": a["ib_conn"][row].clone(), "x0": a["x0"].view(eng.NB, S, d)[row].clone()}
    for l in range(eng.cfg.num_layers):
        out[f"x{l + 1}"] = a[f"x{l + 1}"].view(eng.NB, S, d)[row].clone()
    out["zn"] = a["zn"].view(eng.NB, S, d)[row].clone()
    out["zc"] = a["zc"].view(eng.NB, S, d)[row].clone()
    out["logits"] = a["logits"][row].clone()
    return out


def main():
    z = np.load(GOLDEN / "logits512.npz", allow_pickle=False)
    n, seed = int(z["n"]), int(z["seed"])
    _, kw, cfg, sd, model = build("a5_full", "f32")
    model.eval()
    x1, x2, _ = randn_windows(n, cfg.in_channels, 1024, seed=seed, num_classes=cfg.num_classes)
    s, c0 = 202, 128
    with torch.no_grad():
        la = model(x1[c0:c0 + 128].to(DEV), x2[c0:c0 + 128].to(DEV))["logits"].clone()
        big = stages(model.engine(128, 1024, torch.device(DEV)), s - c0)
        lb = model(x1[c0:c0 + 128].to(DEV), x2[c0:c0 + 128].to(DEV))["logits"].clone()
        print(json.dumps({"chunk_run_twice_bitwise_equal": bool(torch.equal(la, lb)), "max_diff": float((la - lb).abs().max())}), flush=True)
        xa, xb = x1[s:s + 1].repeat(4, 1, 1), x2[s:s + 1].repeat(4, 1, 1)
        model(xa.to(DEV), xb.to(DEV))
        small = stages(model.engine(4, 1024, torch.device(DEV)), 0)
        # a chunk of 128 in which the sample sits at another position
        lc = model(x1[s - 5:s + 123].to(DEV), x2[s - 5:s + 123].to(DEV))["logits"].clone()
        other = stages(model.engine(128, 1024, torch.device(DEV)), 5)
    ref = z["a5_full/logits"][s]
    for k in big:
        da, db = (big[k].float() - small[k].float()).abs(), (big[k].float() - other[k].float()).abs()
        print(json.dumps({"stage": k, "max_abs_B128_vs_B4": float(da.max()), "max_abs_B128_vs_B128_other_position": float(db.max()),
                          "scale": float(big[k].float().abs().max())}), flush=True)
    print(json.dumps({"dlogit_vs_reference": {"B128": float(np.abs(big["logits"].cpu().numpy() - ref).max()),
                                              "B4": float(np.abs(small["logits"].cpu().numpy() - ref).max()),
                                              "B128_other_position": float(np.abs(other["logits"].cpu().numpy() - ref).max())}}))


if __name__ == "__main__":
    main()
