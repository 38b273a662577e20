"""Manual (GPU box): measures the achieved parity errors of the HIP path against every reference-generated fixture and writes
them as JSON (committed as profiles/r0N_parity_table.json; the gates in tests/ are set from these values).
Per fixture x input kind x compute dtype: max |dlogit|, |dloss|, argmax agreement, cls1/cls2 relative error, and in f32 the
worst per-parameter gradient error (norm-relative for every parameter, Frobenius-relative where the fixture holds full tensors).
Plus the 512-sample logits fixture (argmax agreement overall and on decided samples)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.helpers import ALL_CONFIGS, t  # noqa: E402
from tests.test_gpu_model import DEV, build, relerr  # noqa: E402
from tests.test_gpu_logits512 import NAMES as L512, run as run512  # noqa: E402


def fixture_rows():
    rows = []
    for name in ALL_CONFIGS:
        for dtype in ("f32", "bf16"):
            for kind in ("randn", "gen_eeg"):
                z, kw, cfg, sd, model = build(name, dtype)
                model.eval()
                x1, x2, labels = t(z[f"{kind}/eeg1"]).to(DEV), t(z[f"{kind}/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
                out = model(x1, x2, labels)
                loss = out["loss_ce"] + (out["loss_ibs_cls"] if "loss_ibs_cls" in out else 0.0)
                has_grad = True
                try:
                    loss.backward()
                except Exception as e:      # tiny_full in bf16: d_model/2 = 32 is below the bf16 K-tile of the backward-data GEMM
                    has_grad = False
                    print("no backward:", name, dtype, str(e)[:80], flush=True)
                torch.cuda.synchronize()
                got = out["logits"].detach().float().cpu().numpy()
                ref = z[f"{kind}/out/logits"]
                top2 = np.sort(ref, -1)
                row = dict(fixture=name, kind=kind, dtype=dtype, samples=int(ref.shape[0]),
                           max_abs_dlogit=float(np.abs(got - ref).max()), max_abs_logit=float(np.abs(ref).max()),
                           min_top2_margin=float((top2[:, -1] - top2[:, -2]).min()),
                           argmax_agree=float((got.argmax(-1) == z[f"{kind}/out/argmax"]).mean()),
                           dloss_ce=abs(float(out["loss_ce"]) - float(z[f"{kind}/out/loss_ce"])),
                           cls1_relerr=relerr(out["cls1"].detach().float().cpu(), z[f"{kind}/out/cls1"]),
                           cls2_relerr=relerr(out["cls2"].detach().float().cpu(), z[f"{kind}/out/cls2"]))
                if "ibs_logits" in out:
                    row["max_abs_dibs_logit"] = float(np.abs(out["ibs_logits"].detach().float().cpu().numpy() - z[f"{kind}/out/ibs_logits"]).max())
                if not has_grad:
                    rows.append(row)
                    print(json.dumps(row), flush=True)
                    continue
                names = [str(n) for n in z[f"{kind}/grad/names"]]
                params = dict(model.named_parameters())
                gscale = float(z[f"{kind}/grad/global_norm"])
                gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
                row["global_grad_norm_relerr"] = abs(gn - gscale) / gscale
                worst = (0.0, "")
                for n, refn in zip(names, z[f"{kind}/grad/norms"]):
                    if refn < 1e-3 * gscale:
                        continue
                    e = abs(float(params[n].grad.norm()) - refn) / refn
                    if e > worst[0]:
                        worst = (e, n)
                row["worst_param_grad_norm_relerr"], row["worst_param"] = worst
                wf = (0.0, "")
                for key in z.files:
                    if key.startswith(f"{kind}/grad/full/"):
                        n = key.split("/full/")[1]
                        r = torch.from_numpy(z[key]).double()
                        if float(r.norm()) < 1e-5 * gscale:
                            continue
                        e = float((params[n].grad.cpu().double() - r).norm() / r.norm())
                        if e > wf[0]:
                            wf = (e, n)
                if wf[1]:
                    row["worst_full_grad_frobenius_relerr"], row["worst_full_grad_param"] = wf
                rows.append(row)
                print(json.dumps(row), flush=True)
    return rows


def sign_flip_rows():
    """f32 mode, configurations with synchrony tokens: WHICH connectivity entries differ from the reference's, and by how much.
    PLI = |mean_t sign(dphi_t)| (D:613-630): one sample whose sign flips (dphi within rounding of 0 or +-pi: the radix-2 LDS FFT
    rounds differently from pocketfft) moves the signed mean by exactly 2/T (1/T through sign(0) = 0), so every PLI difference must sit on the
    lattice k/T up to fp32 noise -- that is measured here per fixture, entry by entry; wPLI (power-weighted, D:632-658) moves by 2 w_t.
    All other features (PLV, coherence, correlations, phase difference) are continuous and must agree to 1e-4."""
    rows = []
    T = 1024
    for name in ("a5_full", "b1_no_inorm", "b2_phase", "b3_amplitude", "tiny_full", "a5_c32"):
        for kind in ("randn", "gen_eeg"):
            z, kw, cfg, sd, model = build(name, "f32")
            model.eval()
            x1, x2, labels = t(z[f"{kind}/eeg1"]).to(DEV), t(z[f"{kind}/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
            with torch.no_grad():
                out = model(x1, x2, labels)
            torch.cuda.synchronize()
            eng = next(iter(model._engines.values()))
            conn = eng.a["ib_conn"].cpu().numpy()[:2][:, :, cfg.feature_indices].astype(np.float64)
            ref = z[f"{kind}/stage/connectivity"].astype(np.float64)
            row = dict(fixture=name, kind=kind, entries_per_feature=int(conn[:, :, 0].size),
                       max_abs_dlogit=float(np.abs(out["logits"].float().cpu().numpy() - z[f"{kind}/out/logits"]).max()))
            for j, f in enumerate(cfg.feature_indices):
                d = conn[:, :, j] - ref[:, :, j]
                key = {0: "plv", 1: "pli", 2: "wpli", 3: "coherence", 4: "power_corr", 5: "phase_diff", 6: "time_corr"}.get(f, f"f{f}")
                if f == 1:
                    k = d * T                                           # in units of 1/T: a +1 <-> -1 flip is 2, a flip through sign(0) = 0 is 1
                    moved = np.abs(d) > 1e-4
                    row["pli"] = dict(entries_moved=int(moved.sum()), max_abs_diff=float(np.abs(d).max()),
                                      max_shift_in_units_of_1_over_T=float(np.abs(np.round(k[moved])).max()) if moved.any() else 0.0,
                                      max_distance_from_lattice_in_units_of_1_over_T=float(np.abs(k[moved] - np.round(k[moved])).max()) if moved.any() else 0.0,
                                      max_abs_diff_of_unmoved=float(np.abs(d[~moved]).max()))
                elif f == 2:
                    moved = np.abs(d) > 1e-4
                    row["wpli"] = dict(entries_moved=int(moved.sum()), max_abs_diff=float(np.abs(d).max()),
                                       max_abs_diff_of_unmoved=float(np.abs(d[~moved]).max()))
                else:
                    row[key] = dict(max_abs_diff=float(np.abs(d).max()))
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


def logits512_rows(dtypes=("f32", "bf16", "fp16")):
    rows = []
    for name in L512:
        for dtype in dtypes:
            got, ref, am, margin = run512(name, dtype)
            decided = margin > 4e-2
            agree = got.argmax(-1) == am
            row = dict(fixture="logits512/" + name, dtype=dtype, samples=int(len(am)), max_abs_dlogit=float(np.abs(got - ref).max()),
                       argmax_agree=float(agree.mean()), decided=int(decided.sum()), argmax_agree_decided=float(agree[decided].mean()),
                       smallest_margin_of_a_disagreement=(float(margin[~agree].min()) if (~agree).any() else None),
                       largest_margin_of_a_disagreement=(float(margin[~agree].max()) if (~agree).any() else None))
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    out = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03_parity_table.json")
    res = {"device": torch.cuda.get_device_name(0), "logits512": logits512_rows(), "sign_flips": sign_flip_rows(),
           "fixtures": fixture_rows()}
    out.write_text(json.dumps(res, indent=1))
