"""Multimodal logit-fusion loop (BASELINE configs[4]): HIP image CNN branch + HIP DualEEGTransformer + HIP fuzzy gate,
one global clip, per-group AdamW, per-step warm-up + cosine, fp16 with dynamic loss scaling.
Checker: oracle/multimodal_oracle.py (plain torch on the CPU); its EEG and fuzzy parts are pinned by reference fixtures."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import DualEEGTransformer  # noqa: E402
from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion  # noqa: E402
from eyegaze_multimodal_amd.image_encoder import GazeCNNEncoder  # noqa: E402
from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import (MultimodalFusionModel, MultimodalTrainer, synth_multimodal,  # noqa: E402
                                                                    warmup_cosine_factor)
from oracle import dual_eeg_oracle as O  # noqa: E402
from oracle.multimodal_oracle import Stepper, image_logits  # noqa: E402

DEV = "cuda"
KW = dict(in_channels=8, num_classes=3, max_len=256, use_spectrogram=True, use_ibs=False, use_cross_attention=True,
          d_model=64, num_layers=2, num_heads=2, d_ff=128)


def make(dtype, seed=3):
    torch.manual_seed(seed)
    eeg = DualEEGTransformer(**KW, compute_dtype=dtype)
    gaze = GazeCNNEncoder(num_classes=3, d_model=64, compute_dtype=dtype)
    fusion = FuzzyGatingFusion(num_classes=3, mode="full")
    return MultimodalFusionModel(gaze, eeg, fusion)


def test_image_branch_matches_torch_cnn():
    model = make("f32")
    cpu = copy.deepcopy(model.gaze_encoder).eval()
    B, F_, W_ = 6, 64, 16
    img1, img2, *_ = synth_multimodal(B, 8, 1024, F_, W_, 3, seed=1)
    enc = model.gaze_encoder.to(DEV)
    eng = enc.engine(B, F_, W_, DEV)
    got = eng.forward(img1.to(DEV), img2.to(DEV), train=False).cpu()
    with torch.no_grad():
        ref = image_logits(cpu, img1, img2)
    assert float((got - ref).abs().max()) < 2e-5
    # backward from a fixed logit gradient vs torch autograd through the CPU modules
    gl = torch.randn(B, 3, generator=torch.Generator().manual_seed(2))
    eng.backward(gl.to(DEV))
    torch.cuda.synchronize()
    for p in cpu.parameters():
        p.grad = None
    (image_logits(cpu, img1, img2) * gl).sum().backward()
    fp = enc._flat
    for (n, p), q in zip(enc.named_parameters(), cpu.parameters()):
        g = fp.grad[fp.offsets[n]: fp.offsets[n] + p.numel()].view(p.shape).cpu().double()
        r = q.grad.double()
        assert float((g - r).norm() / r.norm().clamp_min(1e-12)) < 1e-3, n


def test_f32_steps_match_the_cpu_restatement():
    model = make("f32")
    B, F_, W_ = 8, 64, 16
    img1, img2, x1, x2, y = synth_multimodal(B, 8, 1024, F_, W_, 3, seed=5)
    hp = dict(encoder_lr=2e-4, fusion_lr=2e-3, weight_decay=0.01, max_grad_norm=1.0)
    cpu_gaze = copy.deepcopy(model.gaze_encoder)
    eeg_sd = {k: v.clone() for k, v in model.eeg_encoder.state_dict().items()}
    fus_sd = {k: v.clone() for k, v in model.fusion.state_dict().items()}
    ref = Stepper(cpu_gaze, O.ModelCfg(**KW), eeg_sd, fus_sd, "full", hp["encoder_lr"], hp["fusion_lr"], hp["weight_decay"],
                  hp["max_grad_norm"], (0.3, 0.3, 0.1), (0.5, 5.0), warmup_steps=2, total_steps=10)
    tr = MultimodalTrainer(model, DEV, **hp, warmup_steps=2, total_steps=10)
    d = lambda t_: t_.to(DEV)
    for step in range(3):
        out = tr.train_step(d(img1), d(img2), d(x1), d(x2), d(y), dropout=False)
        r = ref.step(img1, img2, x1, x2, y)
        torch.cuda.synchronize()
        assert abs(float(out["loss"]) - float(r["loss"])) < 2e-5, (step, float(out["loss"]), float(r["loss"]))
        np.testing.assert_allclose(out["alpha"].cpu().numpy(), r["alpha"].numpy(), atol=2e-5)
        np.testing.assert_allclose(out["fused_logits"].cpu().numpy(), r["fused"].numpy(), atol=5e-5)
        # gradients of all three parameter sets (un-clipped)
        efp = model.eeg_encoder._flat
        for n, p in model.eeg_encoder.named_parameters():
            g = efp.grad[efp.offsets[n]: efp.offsets[n] + p.numel()].view(p.shape).cpu().double()
            rr = r["grads"]["eeg"][n].double()
            if float(rr.norm()) < 1e-7:
                continue
            assert float((g - rr).norm() / rr.norm()) < 3e-3, (step, n)
        gfp = model.gaze_encoder._flat
        gg = torch.cat([gfp.grad[gfp.offsets[n]: gfp.offsets[n] + p.numel()] for n, p in model.gaze_encoder.named_parameters()]).cpu().double()
        assert float((gg - r["grads"]["gaze"].double()).norm() / r["grads"]["gaze"].double().norm()) < 3e-3
        st = model.eeg_encoder.engine(B, 1024, DEV).read_state()
        assert abs(st.grad_norm - float(r["norm"])) < 2e-3 * float(r["norm"])
        assert abs(st.lr - hp["encoder_lr"] * warmup_cosine_factor(step, 2, 10)) < 1e-9      # lr travels as fp32
    # parameters after three steps (first steps of Adam move every element by ~lr: compare the update's size and direction)
    for n, p in model.fusion.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), ref.fus[n].detach().numpy(), rtol=0, atol=2e-4, err_msg=n)
    delta_hip = torch.cat([(p.detach().cpu() - eeg_sd[n]).reshape(-1) for n, p in model.eeg_encoder.named_parameters()]).double()
    delta_ref = torch.cat([(ref.eeg[n].detach() - eeg_sd[n]).reshape(-1) for n, _ in model.eeg_encoder.named_parameters()]).double()
    cos = float((delta_hip * delta_ref).sum() / (delta_hip.norm() * delta_ref.norm()))
    assert cos > 0.98 and abs(float(delta_hip.norm() / delta_ref.norm()) - 1) < 0.05, (cos, float(delta_hip.norm()), float(delta_ref.norm()))


def test_fp16_loop_learns_with_loss_scaling():
    model = make("fp16")
    n, B, F_, W_ = 256, 32, 64, 16
    img1, img2, x1, x2, y = synth_multimodal(n, 8, 1024, F_, W_, 3, seed=9)
    tr = MultimodalTrainer(model, DEV, encoder_lr=5e-4, fusion_lr=5e-3, max_grad_norm=1.0, warmup_steps=4, total_steps=64)
    d = lambda t_: t_.to(DEV)
    losses = []
    for step in range(64):
        j = torch.arange(step * B, (step + 1) * B) % n
        out = tr.train_step(d(img1[j]), d(img2[j]), d(x1[j]), d(x2[j]), d(y[j]))
        losses.append(float(out["loss"]))
    st = model.eeg_encoder.engine(B, 1024, DEV).read_state()
    assert st.scaler_on == 1 and np.isfinite(st.loss_scale) and st.loss_scale >= 1.0
    assert st.opt_steps + st.skipped == 64 and st.skipped <= 8, (st.opt_steps, st.skipped)
    assert np.isfinite(losses).all() and np.mean(losses[-8:]) < 0.7 * np.mean(losses[:8]), (losses[:4], losses[-4:])
    ev = tr.evaluate([(d(img1[:64]), d(img2[:64]), d(x1[:64]), d(x2[:64]), d(y[:64]))])
    assert ev["accuracy"] > 0.6, ev
    assert 0.0 <= ev["alpha_mean"] <= 1.0


def test_warmup_cosine_equals_lambdalr():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    import math
    ws, ts = 5, 40
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: (s / max(1, ws)) if s < ws else max(0.0, 0.5 * (1 + math.cos(math.pi * (s - ws) / max(1, ts - ws)))))
    for s in range(45):
        assert abs(opt.param_groups[0]["lr"] - warmup_cosine_factor(s, ws, ts)) < 1e-12
        opt.step()
        sched.step()


def test_f32_loop_matches_the_reference_loop_fixture():
    """tests/golden/mm_loop.npz = the REFERENCE's own train_one_epoch + optimizer / scheduler construction run on CPU
    (oracle/make_golden_mm_loop.py): the HIP trainer must reproduce its epoch losses, per-step learning rates, clipped norms and
    parameter updates over the same 2 x 4 steps."""
    from tests.test_mm_loop_golden import fixture_batches, fixture_models, load_fixture
    z, meta = load_fixture()
    cfg, eeg_sd, gaze, fusion = fixture_models(meta)
    eeg = DualEEGTransformer(**meta["eeg_kw"], compute_dtype="f32")
    eeg.load_state_dict(eeg_sd)
    model = MultimodalFusionModel(gaze, eeg, fusion)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    t, fz = meta["config"]["training"], meta["config"]["fusion"]
    spe = meta["n"] // meta["batch"]
    tr = MultimodalTrainer(model, DEV, encoder_lr=t["encoder_learning_rate"], fusion_lr=t["fusion_learning_rate"],
                           weight_decay=t["weight_decay"], max_grad_norm=t["max_grad_norm"], lambda_aux_img=t["lambda_aux_img"],
                           lambda_aux_eeg=t["lambda_aux_eeg"], lambda_reg=t["lambda_reg"], temp_reg_min=fz["temp_reg_min"],
                           temp_reg_max=fz["temp_reg_max"], warmup_steps=t["warmup_epochs"] * spe, total_steps=t["epochs"] * spe)
    batches = fixture_batches(meta)
    step = 0
    for epoch in range(meta["epochs"]):
        acc = {k: [] for k in ("loss", "loss_ce", "loss_aux_img", "loss_aux_eeg", "loss_reg")}
        alphas = []
        for b in batches:
            out = tr.train_step(*(x.to(DEV) for x in b), dropout=False)
            st = model.eeg_encoder.engine(meta["batch"], 1024, DEV).read_state()
            assert abs(st.lr - float(z["lrs"][step][0])) <= 1e-7 * max(1e-9, float(z["lrs"][step][0])) + 1e-12, (step, st.lr)
            assert abs(min(st.grad_norm, t["max_grad_norm"]) - float(z["clipped_grad_norm"][step])) < 2e-3, (step, st.grad_norm)
            for k in acc:
                acc[k].append(float(out[k]))
            alphas.append(out["alpha"].cpu().numpy())
            step += 1
        for k in acc:
            assert abs(np.mean(acc[k]) - float(z[f"epoch{epoch}/{k}"])) < 1e-4, (epoch, k, np.mean(acc[k]), float(z[f"epoch{epoch}/{k}"]))
        assert abs(np.concatenate(alphas).mean() - float(z[f"epoch{epoch}/alpha_mean"])) < 2e-5
    final = model.state_dict()
    worst = 0.0
    for i, n in enumerate(str(x) for x in z["param_names"]):
        if n.endswith("k_proj.bias"):
            continue      # the key bias has NO gradient (soft-max is shift-invariant): Adam turns its rounding noise into lr-sized steps
        delta = (final[n].detach().cpu() - init[n].cpu()).double()
        dn = float(z["delta_norm"][i])
        worst = max(worst, abs(float(delta.norm()) - dn) / max(dn, 1e-9))
        assert abs(float(delta.norm()) - dn) <= 2e-2 * dn + 1e-6, (n, float(delta.norm()), dn)
        if n.startswith("fusion."):
            np.testing.assert_allclose(final[n].detach().cpu().double().numpy(), z["final/" + n], rtol=0, atol=2e-4, err_msg=n)


@pytest.mark.parametrize("dtype", ["f32", "fp16"])
@pytest.mark.parametrize("mode", ["full", "no_temperature", "no_fuzzification", "fixed_weights"])
def test_fused_loss_path_equals_the_autograd_path(dtype, mode, monkeypatch):
    """eg_fusion_loop_loss + the gate kernels (default) against the torch autograd graph they replace (EYEGAZE_MM_FUSED_LOSS=0):
    same losses, alpha, fused logits and -- after three steps with dropout off -- the same parameters of all three sets."""
    res = {}
    B, F_, W_ = 8, 64, 16
    img1, img2, x1, x2, y = synth_multimodal(B, 8, 1024, F_, W_, 3, seed=5)
    d = lambda t_: t_.to(DEV)
    for flag in ("1", "0"):
        monkeypatch.setenv("EYEGAZE_MM_FUSED_LOSS", flag)
        torch.manual_seed(3)
        eeg = DualEEGTransformer(**KW, compute_dtype=dtype)
        gaze = GazeCNNEncoder(num_classes=3, d_model=64, compute_dtype=dtype)
        fusion = FuzzyGatingFusion(num_classes=3, mode=mode)
        with torch.no_grad():                      # temperatures outside [0.5, 5]: the regulariser and its gradient are active
            fusion.tau_img.fill_(6.0)
            fusion.tau_eeg.fill_(-2.0)
        model = MultimodalFusionModel(gaze, eeg, fusion)
        tr = MultimodalTrainer(model, DEV, encoder_lr=2e-4, fusion_lr=2e-3, warmup_steps=1, total_steps=10)
        assert tr.fused_loss == (flag == "1")
        outs = [tr.train_step(d(img1), d(img2), d(x1), d(x2), d(y), dropout=False) for _ in range(3)]
        torch.cuda.synchronize()
        res[flag] = (outs, [f.clone() for f in (eeg._flat.flat, gaze._flat.flat, tr.fus.flat)],
                     model.eeg_encoder.engine(B, 1024, DEV).read_state())
    for a, b in zip(res["1"][0], res["0"][0]):
        for k in ("loss", "loss_ce", "loss_aux_img", "loss_aux_eeg", "loss_reg"):
            assert abs(float(a[k]) - float(b[k])) < (2e-6 if dtype == "f32" else 2e-3) * max(1.0, abs(float(b[k]))), k
        np.testing.assert_allclose(a["alpha"].cpu().numpy(), b["alpha"].cpu().numpy(), atol=1e-6 if dtype == "f32" else 2e-3)
    assert float(res["1"][0][0]["loss_reg"]) > 0.5            # the regulariser really was active
    for fa, fb in zip(res["1"][1], res["0"][1]):
        num, den = float((fa - fb).double().norm()), float(fb.double().norm())
        assert num <= (2e-6 if dtype == "f32" else 2e-3) * den, (num, den)
    assert res["1"][2].opt_steps == res["0"][2].opt_steps
