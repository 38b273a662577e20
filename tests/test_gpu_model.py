"""GPU parity tests, model level: the HIP DualEEGTransformer against the golden fixtures emitted by the
reference (tests/golden, see oracle/make_golden.py) and against the CPU oracle on fresh seeded inputs.

Tolerances = at most 2x the errors MEASURED on the MI355X (profiles/r02_parity_table.json, written by tests/parity_table.py;
SURVEY.md 8c evidence: reference fp32 vs fp64 differs by ~2e-7 on logits):
  bf16 compute (bf16 storage, fp32 accumulate):  measured max |dlogit| 1.74e-2 over all fixtures -> gate 3e-2; argmax equal
      wherever the reference's top-2 margin exceeds 4e-2; stage tensors rtol 3e-2 of their max-abs.
  f32 compute (exact-fp32 MFMA): measured max |dlogit| 1.8e-6 -> gate 4e-6 (SURVEY 8c proposed 1e-5), argmax on every sample;
      synchrony tokens included on these fixtures (1.8e-6; over 512 samples the sign()-based PLI / wPLI add up to 2.2e-5, which
      tests/test_gpu_logits512.py attributes to the flipped entries by injecting the reference's connectivity).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW  # noqa: E402
from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from oracle import dual_eeg_oracle as O  # noqa: E402
from tests.helpers import load_golden, t  # noqa: E402

DEV = "cuda"
BASE_CONFIGS = ["cfg1_a1_2class", "cfg2_concat", "cfg3_xattn", "tiny_a1"]
TOKEN_CONFIGS = ["cfg5_a2_spec", "a3_ibs_scalar", "a5_full", "b1_no_inorm", "b2_phase", "b3_amplitude", "tiny_full", "a5_c32"]
ALL = BASE_CONFIGS + TOKEN_CONFIGS


def _check_token_stages(z, kind, cfg, eng, tight):
    """STFT image, connectivity matrices and token rows against the reference's stage tensors."""
    C = cfg.in_channels
    if cfg.use_spectrogram:
        lm = eng.a["spimg"].cpu().numpy()[: 2 * C]
        # compared in the magnitude domain: log() of a near-zero bin amplifies fp32 noise of the 128-point DFT
        np.testing.assert_allclose(np.exp(lm), np.exp(z[f"{kind}/stage/logmag1"]), atol=2e-5, rtol=2e-4)
        spec = eng.a["x0"].float().cpu().view(eng.NB, eng.S, -1)[:2, 1 + eng.n_ibs: 1 + eng.n_ibs + C]
        pos = eng.fp.flat[eng.fp.offsets["pos_embed.pos_embed.weight"]:].view(-1)[: eng.S * cfg.d_model].view(eng.S, -1).cpu()
        got = spec - pos[1 + eng.n_ibs: 1 + eng.n_ibs + C]
        assert relerr(got, z[f"{kind}/stage/spec1"]) < (2e-3 if tight else 3e-2)
    if cfg.use_ibs and cfg.use_robust_ibs:
        conn = eng.a["ib_conn"].cpu().numpy()[:2][:, :, cfg.feature_indices]
        ref = z[f"{kind}/stage/connectivity"]
        T = 1024
        for j, f in enumerate(cfg.feature_indices):
            dlt = np.abs(conn[:, :, j] - ref[:, :, j])
            if f in (1, 2):   # sign()-based: one flipped sample of T moves PLI by 2/T
                assert (dlt > 1e-4).mean() < 5e-3 and dlt.max() < 6.5 / T, (f, dlt.max())
            else:
                assert dlt.max() < 1e-4, (f, dlt.max())
        tok = eng.a["x0"].float().cpu().view(eng.NB, eng.S, -1)[:2, 1: 1 + eng.n_ibs]
        pos = eng.fp.flat[eng.fp.offsets["pos_embed.pos_embed.weight"]:].view(-1)[: eng.S * cfg.d_model].view(eng.S, -1).cpu()
        got = tok - pos[1: 1 + eng.n_ibs]
        assert relerr(got, z[f"{kind}/stage/ibs_tokens"]) < (2e-2 if tight else 4e-2)


def build(name, dtype="bf16"):
    z, kw, cfg, sd = load_golden(name)
    model = DualEEGTransformer(**kw, compute_dtype=dtype)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd, strict=True)
    return z, kw, cfg, sd, model.to(DEV)


def relerr(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["randn", "gen_eeg"])
def test_eval_forward_matches_reference(name, kind):
    z, kw, cfg, sd, model = build(name)
    model.eval()
    x1, x2, labels = t(z[f"{kind}/eeg1"]).to(DEV), t(z[f"{kind}/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
    with torch.no_grad():
        out = model(x1, x2, labels)
    torch.cuda.synchronize()
    ref_logits = z[f"{kind}/out/logits"]
    got = out["logits"].cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref_logits).max()
    assert err <= 3e-2, f"logits err {err}"
    top2 = np.sort(ref_logits, -1)
    decided = (top2[:, -1] - top2[:, -2]) > 4e-2
    assert (got.argmax(-1)[decided] == z[f"{kind}/out/argmax"][decided]).all()
    assert abs(float(out["loss_ce"]) - float(z[f"{kind}/out/loss_ce"])) < 2e-2
    for k in ("cls1", "cls2", "ibs_token"):
        if k in out:
            assert relerr(out[k].cpu(), z[f"{kind}/out/{k}"]) < 3e-2, k
    if "ibs_logits" in out:
        assert np.abs(out["ibs_logits"].cpu().numpy() - z[f"{kind}/out/ibs_logits"]).max() < 3e-2
        assert abs(float(out["loss_ibs_cls"]) - float(z[f"{kind}/out/loss_ibs_cls"])) < 2e-2
    eng = next(iter(model._engines.values()))
    _check_token_stages(z, kind, cfg, eng, tight=False)
    NB, S, d = eng.NB, eng.S, cfg.d_model
    h1 = eng.a["h1"].float().cpu().view(NB, eng.T2, d)[:2]
    assert relerr(h1, z[f"{kind}/stage/h1"]) < 2e-2
    zn = eng.a["zn"].float().cpu().view(NB, S, d)[:2]
    assert relerr(zn, z[f"{kind}/stage/z1"]) < 3e-2
    if cfg.use_cross_attention:
        zc = eng.a["zc"].float().cpu().view(NB, S, d)[:2]
        assert relerr(zc, z[f"{kind}/stage/zc1"]) < 3e-2


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["randn", "gen_eeg"])
def test_f32_forward_and_gradients_are_tight(name, kind):
    """compute_dtype='f32' (exact-fp32 MFMA + fmaf attention): logits within 4e-6 of the reference's fp32 CPU
    result (measured 1.8e-6), argmax bit-exact on every sample, every parameter gradient within 1e-3 relative (Frobenius;
    measured 6.6e-4).  Configurations with synchrony tokens keep the logit gate and get 2e-2 on gradients: the sign()-based PLI /
    wPLI features are discontinuous (one flipped sample of T=1024 moves an entry by 2e-3: gradient of the tokenizer's instance-norm
    gain 1.45e-2) and the radix-2 LDS FFT rounds differently from pocketfft."""
    z, kw, cfg, sd, model = build(name, "f32")
    model.eval()
    x1, x2, labels = t(z[f"{kind}/eeg1"]).to(DEV), t(z[f"{kind}/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
    out = model(x1, x2, labels)
    loss = out["loss_ce"] + (out["loss_ibs_cls"] if "loss_ibs_cls" in out else 0.0)
    loss.backward()
    torch.cuda.synchronize()
    ibs = cfg.use_ibs
    # logits: measured <= 1.8e-6 on every fixture, synchrony tokens included (profiles/r03_parity_table.json: on these 4-sample
    # fixtures NO PLI entry moves at C = 8 and 1-4 of 6 144 move by exactly 2/T at C = 32), so one gate; the sign()-based features
    # still loosen the GRADIENT gate of the tokenizer's instance-norm gain (1.45e-2 measured)
    ltol, gtol = (4e-6, 2e-2) if ibs else (4e-6, 1e-3)
    got = out["logits"].detach().cpu().numpy()
    assert np.abs(got - z[f"{kind}/out/logits"]).max() <= ltol
    assert (got.argmax(-1) == z[f"{kind}/out/argmax"]).all()
    assert abs(float(out["loss_ce"]) - float(z[f"{kind}/out/loss_ce"])) < ltol
    eng = next(iter(model._engines.values()))
    _check_token_stages(z, kind, cfg, eng, tight=True)
    for k in ("cls1", "cls2"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), z[f"{kind}/out/{k}"], rtol=10 * ltol, atol=ltol)
    names = [str(n) for n in z[f"{kind}/grad/names"]]
    params = dict(model.named_parameters())
    gscale = float(z[f"{kind}/grad/global_norm"])
    for n, ref in zip(names, z[f"{kind}/grad/norms"]):
        got_n = float(params[n].grad.norm())
        assert abs(got_n - ref) <= 2 * gtol * ref + 1e-6 * gscale, (n, got_n, ref)
    for key in z.files:
        if key.startswith(f"{kind}/grad/full/"):
            n = key.split("/full/")[1]
            ref = torch.from_numpy(z[key]).double()
            g = params[n].grad.cpu().double()
            if float(ref.norm()) < 1e-5 * gscale:  # k_proj.bias: mathematically zero (soft-max shift invariance)
                assert float(g.norm()) < 1e-5 * gscale, n
                continue
            assert float((g - ref).norm() / ref.norm()) < gtol, n


def _oracle_grads(cfg, sd, x1, x2, labels, round_bf16):
    rb = (lambda v: v.to(torch.bfloat16).float()) if round_bf16 else (lambda v: v)
    params = {k: rb(v).clone().requires_grad_(True) for k, v in sd.items()}
    out = O.forward(rb(x1), rb(x2), params, cfg, labels)
    out["loss_ce"].backward()
    return {k: p.grad for k, p in params.items()}


@pytest.mark.parametrize("name", BASE_CONFIGS)
def test_bf16_gradients_track_the_oracle(name):
    """bf16 compute.  The fp32 oracle's own gradients move by 2-35 % (Frobenius) on these 4-sample fixtures when
    its weights are merely ROUNDED to bf16 (ReLU gates and soft-max saturation flip; measured in DESIGN.md), so a
    bf16 run cannot be held to a tight element-wise gate.  It is held to: same direction as the oracle evaluated on
    the bf16-rounded weights (cosine >= 0.8, norm within 25 % for every tensor carrying >= 2 % of the global norm;
    smaller tensors are noise-dominated), global norm within 5 %.
    The tight gradient gate is test_f32_forward_and_gradients_are_tight."""
    z, kw, cfg, sd, model = build(name)
    model.eval()  # dropout inactive: deterministic gradients (SURVEY §7 'Dropout')
    kind = "randn"
    x1, x2, labels = t(z[f"{kind}/eeg1"]), t(z[f"{kind}/eeg2"]), t(z["labels"])
    out = model(x1.to(DEV), x2.to(DEV), labels.to(DEV))
    out["loss_ce"].backward()
    torch.cuda.synchronize()
    ref = _oracle_grads(cfg, sd, x1, x2, labels, round_bf16=True)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())))
    rn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
    assert abs(gn - rn) < 5e-2 * rn, (gn, rn)
    bad = []
    for n, p in model.named_parameters():
        g, r = p.grad.cpu().double(), ref[n].double()
        assert torch.isfinite(g).all(), n
        if float(r.norm()) < 2e-2 * rn:
            continue  # noise-dominated (k_proj.bias is even mathematically zero: soft-max shift invariance)
        cos = float((g * r).sum() / (g.norm() * r.norm()))
        ratio = float(g.norm() / r.norm())
        if cos < 0.8 or abs(ratio - 1) > 0.25:
            bad.append((n, round(cos, 3), round(ratio, 3)))
    assert not bad, bad[:10]


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("name", ["cfg3_xattn", "tiny_a1"])
def test_native_step_matches_reference(name, dtype):
    """forward + backward + clip(1.0) + AdamW(1e-4, .01) entirely in HIP vs the reference's one step."""
    z, kw, cfg, sd, model = build(name, dtype)
    model.eval()
    kind = "randn"
    x1, x2, labels = t(z[f"{kind}/eeg1"]).to(DEV), t(z[f"{kind}/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
    eng = model.engine(x1.shape[0], x1.shape[2], x1.device)
    opt = HipAdamW(model, lr=1e-4, weight_decay=0.01)
    opt.begin_step(eng, seed=1)
    eng.forward(x1, x2, labels, train=False)
    eng.backward(gloss=torch.ones(1, device=DEV))
    opt.step(eng)
    torch.cuda.synchronize()
    st = eng.read_state()
    ref_norm = float(z[f"{kind}/step/total_norm"])
    assert abs(st.grad_norm - ref_norm) < 5e-2 * ref_norm
    names = [str(n) for n in z[f"{kind}/grad/names"]]
    params = dict(model.named_parameters())
    l2 = np.array([float(params[n].detach().double().norm()) for n in names])
    np.testing.assert_allclose(l2, z[f"{kind}/step/param_l2"], rtol=2e-4, atol=5e-4)  # +-lr sign flips of ~zero gradients
    # AdamW's first step moves every element by ~lr regardless of the gradient's magnitude
    delta = np.array([float((params[n].detach().cpu() - sd[n]).double().norm()) for n in names])
    np.testing.assert_allclose(delta, z[f"{kind}/step/delta_l2"], rtol=0.1, atol=2.5e-3)  # lr*sqrt(numel) for ~zero-gradient tensors


def test_train_mode_step_runs_and_learns():
    """Train mode (dropout on): finite losses, masks change with the seed, and a few native steps on a fixed
    batch reduce the loss."""
    torch.manual_seed(0)
    kw = dict(in_channels=8, num_classes=3, max_len=256, use_spectrogram=False, use_ibs=False, use_cross_attention=True)
    model = DualEEGTransformer(**kw).to(DEV)
    model.train()
    g = torch.Generator().manual_seed(1)
    B = 16
    x1 = torch.randn(B, 8, 1024, generator=g).to(DEV)
    x2 = torch.randn(B, 8, 1024, generator=g).to(DEV)
    labels = (torch.arange(B) % 3).to(DEV)
    x1[labels == 1] += 0.5 * torch.sin(torch.arange(1024, device=DEV) * 0.3)
    x1[labels == 2] -= 0.5 * torch.sin(torch.arange(1024, device=DEV) * 0.1)
    eng = model.engine(B, 1024, x1.device)
    opt = HipAdamW(model, lr=3e-4)
    losses = []
    for step in range(30):
        opt.begin_step(eng, seed=100 + step)
        eng.forward(x1, x2, labels, train=True)
        eng.backward(gloss=torch.ones(1, device=DEV))
        opt.step(eng)
        losses.append(float(eng.a["loss"]))
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.05, losses
    # same seed -> identical forward; different seed -> different dropout masks
    opt.begin_step(eng, seed=7)
    eng.forward(x1, x2, labels, train=True)
    a = eng.a["logits"].clone()
    eng.forward(x1, x2, labels, train=True)
    b = eng.a["logits"].clone()
    eng.set_state(seed=8, lr=0.0, step=1)
    eng.forward(x1, x2, labels, train=True)
    c = eng.a["logits"].clone()
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_autograd_loop_like_reference():
    """The reference's own loop shape (train_art.py:175-222): zero_grad / forward / loss / backward /
    clip_grad_norm_ / torch AdamW on model.parameters() works against the HIP module."""
    torch.manual_seed(0)
    kw = dict(in_channels=8, num_classes=2, max_len=256, use_spectrogram=False, use_ibs=False)
    model = DualEEGTransformer(**kw).to(DEV)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
    g = torch.Generator().manual_seed(1)
    x1, x2 = torch.randn(8, 8, 1024, generator=g).to(DEV), torch.randn(8, 8, 1024, generator=g).to(DEV)
    labels = (torch.arange(8) % 2).to(DEV)
    before = model.classifier[3].weight.detach().clone()
    for _ in range(2):
        opt.zero_grad()
        out = model(x1, x2, labels)
        loss = out["loss_ce"] + 0.1 * model.compute_symmetry_loss(out["cls1"], out["cls2"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
    assert torch.isfinite(loss)
    assert not torch.equal(before, model.classifier[3].weight.detach())


def test_cpu_tensors_are_refused():
    model = DualEEGTransformer(in_channels=8, max_len=256, use_spectrogram=False, use_ibs=False)
    with pytest.raises(L.EgError):
        model(torch.zeros(2, 8, 1024), torch.zeros(2, 8, 1024))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_grouped_weight_gradients_equal_per_product_path(dtype, monkeypatch):
    """The one-launch grouped weight-gradient path (used at training batch sizes) produces the same gradients as the
    per-product launches the small fixtures exercise."""
    from eyegaze_multimodal_amd.engine import Engine
    z, kw, cfg, sd, model = build("cfg3_xattn", dtype)
    x1, x2, labels = t(z["randn/eeg1"]).to(DEV), t(z["randn/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
    one = torch.ones(1, device=DEV)
    grads = []
    for min_rows in (1 << 30, 0):
        monkeypatch.setattr(Engine, "GROUP_MIN_ROWS", min_rows)
        model._engines.clear()
        eng = model.engine(x1.shape[0], x1.shape[2], x1.device)
        eng.set_state(seed=5, lr=0.0, step=1)
        eng.forward(x1, x2, labels, train=True)      # dropout on: the masked-gradient buffers are exercised too
        eng.backward(gloss=one)
        torch.cuda.synchronize()
        assert (eng._wgrad_group_plan() is not None) == (min_rows == 0)
        grads.append(model._flat.grad.clone())
    ref, got = grads
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    # same products, different split counts: fp32 summation order only
    assert err <= 2e-5 * float(ref.abs().max()) + 1e-7, err


def test_multimodal_fusion_step_reaches_the_hip_backward():
    """Config 5's logit-level fusion step (train_multimodal_fuzzy_fusion.py:432-470) around the HIP EEG model: the gradient
    of the fused loss w.r.t. the EEG logits must enter the HIP backward.  The image branch is a stand-in (fixed logits; the
    reference's ViT needs timm + downloaded weights, absent here).  Check: classifier.3.bias.grad == column sums of the
    oracle's d loss / d eeg_logits evaluated at the model's own logits."""
    import torch.nn.functional as F
    from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion
    from oracle import fuzzy_oracle as FO
    z, kw, cfg, sd, model = build("cfg3_xattn", "f32")
    model.eval()
    x1, x2, labels = t(z["gen_eeg/eeg1"]).to(DEV), t(z["gen_eeg/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
    fusion = FuzzyGatingFusion(num_classes=3, mode="full").to(DEV)
    img_logits = torch.tensor([[2.0, -1.0, 0.5], [0.1, 0.0, -0.1], [-1.0, 3.0, 0.0], [0.3, 0.2, 0.1]], device=DEV, requires_grad=True)
    out = model(x1, x2)
    fused, alpha, aux = fusion(img_logits, out["logits"])
    T_i, T_e = aux["temperatures"]["img"], aux["temperatures"]["eeg"]
    loss = (F.cross_entropy(fused, labels) + 0.3 * F.cross_entropy(img_logits / T_i, labels)
            + 0.3 * F.cross_entropy(out["logits"] / T_e, labels) + 0.1 * fusion.compute_temperature_regularization())
    loss.backward()
    torch.cuda.synchronize()
    # oracle on CPU at the same logits
    p = {n: v.detach().cpu().clone().requires_grad_(True) for n, v in fusion.named_parameters()}
    zi = img_logits.detach().cpu().clone().requires_grad_(True)
    ze = out["logits"].detach().cpu().clone().requires_grad_(True)
    ref_loss, _, _ = FO.fusion_loop_loss(zi, ze, labels.cpu(), p, "full")
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    np.testing.assert_allclose(img_logits.grad.cpu().numpy(), zi.grad.numpy(), atol=2e-6)
    got = dict(model.named_parameters())["classifier.3.bias"].grad.cpu().numpy()
    np.testing.assert_allclose(got, ze.grad.sum(0).numpy(), atol=2e-6)
    for n, q in fusion.named_parameters():
        np.testing.assert_allclose(q.grad.cpu().numpy(), p[n].grad.numpy(), atol=2e-6, err_msg=n)
    # and the encoder received it too
    assert float(dict(model.named_parameters())["encoder.layers.0.mha.q_proj.weight"].grad.abs().sum()) > 0


@pytest.mark.parametrize("name,dtype", [("cfg3_xattn", "f32"), ("cfg2_concat", "f32"), ("cfg3_xattn", "bf16")])
def test_train_mode_matches_oracle_with_identical_masks(name, dtype):
    """TRAIN mode, dropout on.  The oracle is fed exactly the masks the kernels draw (tests/helpers.py replicates the
    counter hash on the CPU), so every dropout site, its scaling, its index convention and its backward are checked exactly
    instead of statistically: loss, logits and all gradients as in the eval-mode gates."""
    from tests.helpers import hip_dropout_override
    z, kw, cfg, sd, model = build(name, dtype)
    model.train()
    kind, seed = "gen_eeg", 0x1234_5678_9ABC
    x1c, x2c, labc = t(z[f"{kind}/eeg1"]), t(z[f"{kind}/eeg2"]), t(z["labels"])
    B = x1c.shape[0]
    eng = model.engine(B, x1c.shape[2], torch.device(DEV))
    eng.set_state(seed=seed, lr=0.0, step=1)
    eng.forward(x1c.to(DEV), x2c.to(DEV), labc.to(DEV), train=True)
    eng.backward(gloss=torch.ones(1, device=DEV))
    torch.cuda.synchronize()
    got_logits, got_loss = eng.a["logits"].cpu().numpy(), float(eng.a["loss"])
    # oracle with the same masks
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.DROPOUT_OVERRIDE = hip_dropout_override(seed, B, cfg.num_layers, O.CTX)
    try:
        out = O.forward(x1c, x2c, params, cfg, labc, train=True)
        out["loss_ce"].backward()
    finally:
        O.DROPOUT_OVERRIDE = None
    ref_logits = out["logits"].detach().numpy()
    ltol, gtol = (2e-4, 2e-3) if dtype == "f32" else (5e-2, None)
    assert np.abs(got_logits - ref_logits).max() <= ltol, np.abs(got_logits - ref_logits).max()
    assert abs(got_loss - float(out["loss_ce"])) <= ltol
    # masks really were active: train-mode logits differ from the eval fixture
    assert np.abs(ref_logits - z[f"{kind}/out/logits"]).max() > 1e-3
    fp = model._flat
    gnorm_ref = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None)))
    gflat = fp.grad.cpu().double()
    assert abs(float(gflat.norm()) / gnorm_ref - 1) < (5e-3 if dtype == "f32" else 8e-2)
    if gtol:
        for n, p in zip(fp.names, fp.params):
            ref = params[n].grad
            if ref is None:
                continue
            g = gflat[fp.offsets[n]: fp.offsets[n] + p.numel()].view(p.shape)
            if float(ref.norm()) < 1e-5 * gnorm_ref:
                assert float(g.norm()) < 1e-4 * gnorm_ref, n
                continue
            assert float((g - ref.double()).norm() / ref.double().norm()) < gtol, (n, float((g - ref.double()).norm() / ref.double().norm()))


def test_auxiliary_losses_and_their_gradients():
    """csrc/aux.hip against (i) the values the reference produced on the fixture's fixed (ibs, cls1, cls2, labels) and (ii) the
    oracle's autograd gradients; larger random batches against the oracle; the no-positive-pair case returns 0 with zero
    gradient."""
    z, kw, cfg, sd, model = build("cfg3_xattn", "f32")
    ibs, c1, c2 = (t(z[f"aux/{k}"]) for k in ("ibs", "cls1", "cls2"))
    lab = t(z["aux/labels"])
    d = lambda v: v.to(DEV).requires_grad_(True)
    assert abs(float(model.compute_symmetry_loss(d(c1), d(c2))) - float(z["aux/sym"])) < 1e-6
    assert abs(float(model.compute_ibs_alignment_loss(d(ibs), d(c1), d(c2))) - float(z["aux/align"])) < 1e-5
    assert abs(float(model.compute_ibs_contrastive_loss(d(ibs), lab.to(DEV))) - float(z["aux/contrastive"])) < 1e-5
    g = torch.Generator().manual_seed(2)
    for B, D in ((ibs.shape[0], ibs.shape[1]), (64, 256), (256, 256), (300, 96)):
        if B == ibs.shape[0]:
            a, b1, b2, y = ibs, c1, c2, lab
        else:
            a, b1, b2 = (torch.randn(B, D, generator=g) for _ in range(3))
            b1 = b1 + 0.5 * a                     # some alignment, so the soft-max is not flat
            y = torch.randint(0, 3, (B,), generator=g)
        for kind in ("sym", "infonce", "supcon"):
            ra, r1, r2 = (v.clone().requires_grad_(True) for v in (a, b1, b2))
            ga, g1, g2 = d(a), d(b1), d(b2)
            if kind == "sym":
                ref, got = O.symmetry_loss(r1, r2), model.compute_symmetry_loss(g1, g2)
            elif kind == "infonce":
                ref, got = O.ibs_alignment_loss(ra, r1, r2), model.compute_ibs_alignment_loss(ga, g1, g2)
            else:
                ref, got = O.ibs_contrastive_loss(ra, y), model.compute_ibs_contrastive_loss(ga, y.to(DEV))
            (3.0 * ref).backward()
            (3.0 * got).backward()                # a non-unit upstream gradient
            assert abs(float(got) - float(ref)) < 2e-5 * max(1.0, abs(float(ref))), (kind, B, float(got), float(ref))
            for name, rv, gv in (("ibs", ra, ga), ("cls1", r1, g1), ("cls2", r2, g2)):
                if rv.grad is None:
                    assert gv.grad is None or float(gv.grad.abs().max()) == 0.0
                    continue
                err = float((gv.grad.cpu() - rv.grad).abs().max())
                assert err < 2e-5 * max(1e-3, float(rv.grad.abs().max())) + 1e-8, (kind, name, B, err)
    # no row has a same-label partner -> loss 0, gradient 0 (D:1346-1349)
    x = d(ibs[:3])
    l0 = model.compute_ibs_contrastive_loss(x, torch.tensor([0, 1, 2], device=DEV))
    l0.backward()
    assert float(l0) == 0.0 and float(x.grad.abs().max()) == 0.0
