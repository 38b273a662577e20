"""CPU: the oracle of the multimodal logit-fusion LOOP (oracle/multimodal_oracle.py::Stepper) against tests/golden/mm_loop.npz, which
oracle/make_golden_mm_loop.py produced by executing the REFERENCE's own train_one_epoch / MultimodalFusionModel /
get_linear_warmup_cosine_scheduler and optimizer construction (train_multimodal_fuzzy_fusion.py:106-179, 197-214, 395-543,
727-746) on CPU, image branch = the in-tree 2-D CNN.  This pins the loop (loss composition, shared clip, per-group AdamW,
per-step warm-up + cosine, epoch metrics); before round 3 it was "parity unpinned".  The GPU counterpart holds the HIP trainer to the
same fixture (tests/test_gpu_multimodal.py)."""
import ast
from pathlib import Path

import numpy as np
import torch

from oracle import dual_eeg_oracle as O
from oracle.multimodal_oracle import Stepper, synthetic_gaze_state

GOLD = Path(__file__).resolve().parent / "golden" / "mm_loop.npz"


def load_fixture():
    z = np.load(GOLD, allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    return z, meta


def fixture_models(meta):
    """the three parameter sets the fixture's run started from (seeded generators, no reference needed)"""
    from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion
    from eyegaze_multimodal_amd.image_encoder import GazeCNNEncoder
    cfg = O.ModelCfg(**meta["eeg_kw"])
    eeg_sd = O.synthetic_state_dict(cfg, meta["weight_seed"])
    gaze = GazeCNNEncoder(num_classes=3, d_model=meta["eeg_kw"]["d_model"], compute_dtype="f32")
    gaze.load_state_dict(synthetic_gaze_state(gaze, meta["gaze_seed"]))
    fusion = FuzzyGatingFusion(num_classes=3, mode=meta["config"]["fusion"]["mode"], eps_temp=meta["config"]["fusion"]["eps_temp"])
    return cfg, eeg_sd, gaze, fusion


def fixture_batches(meta):
    from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import synth_multimodal
    data = synth_multimodal(meta["n"], 8, 1024, 64, 16, 3, seed=meta["data_seed"])
    b = meta["batch"]
    return [tuple(t_[i:i + b] for t_ in data) for i in range(0, meta["n"], b)]


def test_oracle_loop_reproduces_the_reference_loop():
    torch.set_num_threads(4)
    z, meta = load_fixture()
    cfg, eeg_sd, gaze, fusion = fixture_models(meta)
    t, fz = meta["config"]["training"], meta["config"]["fusion"]
    spe = meta["n"] // meta["batch"]
    fus_sd = {k: v.clone() for k, v in fusion.state_dict().items()}
    st = Stepper(gaze, cfg, eeg_sd, fus_sd, fz["mode"], t["encoder_learning_rate"], t["fusion_learning_rate"], t["weight_decay"],
                 t["max_grad_norm"], (t["lambda_aux_img"], t["lambda_aux_eeg"], t["lambda_reg"]), (fz["temp_reg_min"], fz["temp_reg_max"]),
                 warmup_steps=t["warmup_epochs"] * spe, total_steps=t["epochs"] * spe)
    init_gaze = {k: v.detach().clone() for k, v in st.gaze.named_parameters()}
    batches = fixture_batches(meta)
    lrs, norms = [], []
    for epoch in range(meta["epochs"]):
        acc = {k: [] for k in ("loss", "loss_ce", "loss_aux_img", "loss_aux_eeg", "loss_reg")}
        alphas, preds, labs = [], [], []
        for b in batches:
            r = st.step(*b)
            for k in acc:
                acc[k].append(float(r[k]))
            alphas.append(r["alpha"].numpy())
            preds.append(r["fused"].argmax(-1).numpy())
            labs.append(b[4].numpy())
            lrs.append(r["lrs"])
            norms.append(min(float(r["norm"]), t["max_grad_norm"]))      # the fixture holds the norm AFTER the clip
        for k in acc:
            assert abs(np.mean(acc[k]) - float(z[f"epoch{epoch}/{k}"])) < 3e-6, (epoch, k, np.mean(acc[k]), float(z[f"epoch{epoch}/{k}"]))
        al = np.concatenate(alphas)
        assert abs(al.mean() - float(z[f"epoch{epoch}/alpha_mean"])) < 2e-6 and abs(al.std() - float(z[f"epoch{epoch}/alpha_std"])) < 2e-6
        assert abs((np.concatenate(preds) == np.concatenate(labs)).mean() - float(z[f"epoch{epoch}/accuracy"])) < 1e-12
    np.testing.assert_allclose(np.asarray(lrs), z["lrs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.asarray(norms), z["clipped_grad_norm"], rtol=2e-4)
    # parameters after the 8 steps: per-tensor size of the update and its first elements
    names = [str(n) for n in z["param_names"]]
    for i, n in enumerate(names):
        grp, key = n.split(".", 1)
        if grp == "eeg_encoder":
            delta = st.eeg[key].detach() - eeg_sd[key]
        elif grp == "gaze_encoder":
            delta = dict(st.gaze.named_parameters())[key].detach() - init_gaze[key]
        else:
            delta = st.fus[key].detach() - fus_sd[key]
            np.testing.assert_allclose(st.fus[key].detach().double().numpy(), z["final/" + n], rtol=0, atol=2e-6, err_msg=n)
        dn = float(z["delta_norm"][i])
        assert abs(float(delta.double().norm()) - dn) <= 2e-3 * dn + 1e-7, (n, float(delta.double().norm()), dn)
        head = delta.reshape(-1)[:16].double().numpy()
        # Adam's first steps move every element by about lr whatever its gradient: elements with a near-zero gradient flip sign on
        # rounding noise, so the element-wise check carries an lr-sized absolute allowance on top of the tensor-level norm check
        np.testing.assert_allclose(head, z["delta_head"][i][:len(head)], rtol=5e-3, atol=3e-5 * (10 if grp == "fusion" else 1), err_msg=n)


def test_fixture_schedule_is_warmup_then_cosine():
    """the per-step learning rates the reference's LambdaLR produced == the product's closed form (warmup_cosine_factor)"""
    from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import warmup_cosine_factor
    z, meta = load_fixture()
    t = meta["config"]["training"]
    spe = meta["n"] // meta["batch"]
    for step, row in enumerate(z["lrs"]):
        f = warmup_cosine_factor(step, t["warmup_epochs"] * spe, t["epochs"] * spe)
        np.testing.assert_allclose(row, [t["encoder_learning_rate"] * f, t["encoder_learning_rate"] * f, t["fusion_learning_rate"] * f],
                                   rtol=1e-12, atol=1e-18)
