"""
TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build container, where the
reference is mounted read-only at /root/reference; never on the GPU box, never from the product path.

It imports the reference's own model file by path (the same way the reference's train script does,
4_Experiments/scripts/train_art.py:31-44), loads deterministic synthetic weights
(oracle.dual_eeg_oracle.synthetic_state_dict) into it, runs it on seeded inputs on CPU/fp32 and writes
small `.npz` fixtures to tests/golden/.  The fixtures hold data only (inputs, expected outputs,
per-parameter gradient norms, post-step parameter checksums); no reference source is copied.

Usage:  python oracle/make_golden.py            # regenerate every fixture
"""
from __future__ import annotations

import importlib.util
import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("EYEGAZE_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REPO))
from oracle.dual_eeg_oracle import ModelCfg, synthetic_state_dict, zscore_window  # noqa: E402


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    sys.path.insert(0, str(REF / "3_Models" / "backbones"))
    model_mod = _load("dual_eeg_transformer", REF / "3_Models" / "backbones" / "dual_eeg_transformer.py")
    gen_mod = _load("two_EEG_fusion", REF / "1_Data" / "processed" / "two_EEG_fusion.py")
    fz_mod = _load("fuzzy_gating_fusion", REF / "3_Models" / "fusion" / "fuzzy_gating_fusion.py")
    return model_mod, gen_mod, fz_mod


# name -> (ModelCfg kwargs).  C=8, T=1024 everywhere (BASELINE.json shape); max_len = 1024//4 (train_art.py:368)
BASE = dict(in_channels=8, max_len=256)
CONFIGS = {
    # BASELINE configs[0]: A1 temporal-only, 2 classes
    "cfg1_a1_2class": dict(BASE, num_classes=2, use_spectrogram=False, use_ibs=False, use_cross_attention=True),
    # configs[1]: concat fusion, no cross-attention
    "cfg2_concat": dict(BASE, num_classes=3, use_spectrogram=False, use_ibs=False, use_cross_attention=False),
    # configs[2]/[3]: cross-attention fusion
    "cfg3_xattn": dict(BASE, num_classes=3, use_spectrogram=False, use_ibs=False, use_cross_attention=True),
    # configs[4]: + STFT image -> 2-D CNN tokens (A2)
    "cfg5_a2_spec": dict(BASE, num_classes=3, use_spectrogram=True, use_ibs=False, use_cross_attention=True),
    # A3 scalar IBS
    "a3_ibs_scalar": dict(BASE, num_classes=3, use_spectrogram=False, use_ibs=True, use_robust_ibs=False),
    # A5 full (reference default flags)
    "a5_full": dict(BASE, num_classes=3, use_spectrogram=True, use_ibs=True, use_robust_ibs=True),
    # B-family tokenizer ablations
    "b1_no_inorm": dict(BASE, num_classes=3, use_spectrogram=False, use_ibs=True, ibs_instance_norm=False),
    "b2_phase": dict(BASE, num_classes=3, use_spectrogram=False, use_ibs=True, ibs_feature_type="phase"),
    "b3_amplitude": dict(BASE, num_classes=3, use_spectrogram=False, use_ibs=True, ibs_feature_type="amplitude"),
    # small-dims model (cheap full-gradient fixture: every gradient tensor stored)
    "tiny_full": dict(in_channels=8, max_len=256, num_classes=3, d_model=64, num_layers=2, num_heads=2, d_ff=128,
                      use_spectrogram=True, use_ibs=True, use_robust_ibs=True),
    "tiny_a1": dict(in_channels=8, max_len=256, num_classes=3, d_model=64, num_layers=2, num_heads=2, d_ff=128,
                    use_spectrogram=False, use_ibs=False, use_cross_attention=True),
}
WEIGHT_SEED = 20260128
B, T = 4, 1024


def make_inputs(gen_mod, kind: str, C: int):
    if kind == "randn":
        # throughput-run generator of SURVEY §8d: randn(seed 1234) then per-window global z-score
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(2, B, C, T, generator=g).numpy()
        x = np.stack([[zscore_window(w) for w in s] for s in x])
    else:
        # reference's own synthesiser 1_Data/processed/two_EEG_fusion.py:31-49; stream 2 shares half of stream 1
        x = np.zeros((2, B, C, T), np.float32)
        for b in range(B):
            a = gen_mod.gen_eeg(C=C, T=T, sample_rate=256.0, mode="mixed", noise_std=0.1, num_components=3, seed=100 + b)
            c = gen_mod.gen_eeg(C=C, T=T, sample_rate=256.0, mode="mixed", noise_std=0.1, num_components=3, seed=200 + b)
            x[0, b] = zscore_window(a)
            x[1, b] = zscore_window(0.5 * a + 0.5 * c)
    return torch.from_numpy(x[0]), torch.from_numpy(x[1])


def run_config(name, kw, model_mod, gen_mod, out_dir: Path):
    cfg = ModelCfg(**kw)
    sd = synthetic_state_dict(cfg, WEIGHT_SEED)
    model = model_mod.DualEEGTransformer(**kw)
    ref_keys = list(model.state_dict().keys())
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert ref_keys == list(sd.keys()), "state_dict key order differs from the reference"
    model.eval()
    labels = torch.tensor([i % cfg.num_classes for i in range(B)])
    blob = {"labels": labels.numpy(), "state_keys": np.array(ref_keys)}
    tiny = name.startswith("tiny")
    for kind in ("randn", "gen_eeg"):
        x1, x2 = make_inputs(gen_mod, kind, cfg.in_channels)
        stages = {}
        hooks = []

        def grab(key):
            def fn(mod, inp, out):
                stages[key] = out
            return fn
        seen = {"n": 0}

        def grab_twice(key):
            def fn(mod, inp, out):
                seen[key] = seen.get(key, 0) + 1
                stages[f"{key}{seen[key]}"] = out
            return fn
        hooks.append(model.temporal_conv.register_forward_hook(grab_twice("h")))
        hooks.append(model.encoder.register_forward_hook(grab_twice("z")))
        if cfg.use_spectrogram:
            hooks.append(model.spectrogram_generator.register_forward_hook(grab_twice("spec")))
        if cfg.use_ibs and cfg.use_robust_ibs:
            hooks.append(model.ibs_matrix_generator.register_forward_hook(grab("connectivity")))
            hooks.append(model.ibs_tokenizer.register_forward_hook(grab("ibs_tokens")))
        if cfg.use_ibs and not cfg.use_robust_ibs:
            hooks.append(model.ibs_generator.register_forward_hook(grab("ibs_scalar_token")))
        if cfg.use_cross_attention:
            hooks.append(model.cross_attn.register_forward_hook(grab("zc")))
        model.zero_grad()
        out = model(x1, x2, labels)
        loss = out["loss_ce"] + (1.0 * out["loss_ibs_cls"] if "loss_ibs_cls" in out else 0.0)
        loss.backward()
        for h in hooks:
            h.remove()
        pre = kind + "/"
        blob[pre + "eeg1"], blob[pre + "eeg2"] = x1.numpy(), x2.numpy()
        for k, v in out.items():
            blob[pre + "out/" + k] = v.detach().numpy()
        blob[pre + "out/argmax"] = out["logits"].argmax(-1).numpy()
        blob[pre + "total_loss"] = loss.detach().numpy()
        # stage intermediates: first two samples only (keeps fixtures small); stream-2 copies only for randn
        keep2 = kind == "randn"
        for k, v in stages.items():
            if isinstance(v, tuple):
                for i, t in enumerate(v):
                    if i == 0 or keep2:
                        blob[pre + f"stage/{k}{i + 1}"] = t.detach().numpy()[:2].copy()
            elif not k.endswith("2") or keep2 and k in ("h2",):
                blob[pre + "stage/" + k] = v.detach().numpy()[:2].copy()
        if cfg.use_spectrogram:  # STFT log-magnitude image, recomputed with the reference's own call
            with torch.no_grad():
                sg = model.spectrogram_generator
                st = torch.stft(x1.reshape(-1, T), n_fft=sg.n_fft, hop_length=sg.hop_length, window=sg.window,
                                return_complex=True, center=True)
                blob[pre + "stage/logmag1"] = torch.log(st.abs()[:, :sg.freq_bins] + 1e-8).numpy()[:2 * cfg.in_channels].copy()
        # gradients (eval mode => dropout inactive => deterministic)
        names = [n for n, _ in model.named_parameters()]
        gn = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for _, p in model.named_parameters()], np.float64)
        blob[pre + "grad/names"] = np.array(names)
        blob[pre + "grad/norms"] = gn
        blob[pre + "grad/global_norm"] = np.array(np.sqrt((gn ** 2).sum()))
        for n, p in model.named_parameters():
            if (tiny and kind == "randn") or n in ("classifier.3.weight", "classifier.3.bias", "cls_token", "encoder.norm.weight",
                             "temporal_conv.convs.0.bias", "encoder.layers.0.mha.q_proj.bias",
                             "encoder.layers.5.ffn.linear2.bias", "cross_attn.norm.weight",
                             "symmetric_fusion.proj.bias", "ibs_classifier.3.weight",
                             "spectrogram_generator.spec_conv.0.weight", "ibs_tokenizer.instance_norm.weight"):
                if p.grad is not None:
                    blob[pre + "grad/full/" + n] = p.grad.numpy().copy()
        if kind == "randn":
            # one optimiser step exactly as the loop does it (train_art.py:221-222, 401-405)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
            total = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            opt.step()
            blob[pre + "step/total_norm"] = np.array(float(total))
            blob[pre + "step/param_sum"] = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
            blob[pre + "step/param_l2"] = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
            blob[pre + "step/delta_l2"] = np.array([float((p.detach() - sd[n]).double().norm()) for n, p in model.named_parameters()])
            model.load_state_dict(sd, strict=True)
    # aux losses on fixed vectors (D:1255-1371)
    g = torch.Generator().manual_seed(7)
    ibs = torch.randn(6, cfg.d_model, generator=g)
    c1 = torch.randn(6, cfg.d_model, generator=g)
    c2 = torch.randn(6, cfg.d_model, generator=g)
    lab = torch.tensor([0, 1, 2, 0, 1, 1]) % cfg.num_classes
    blob["aux/ibs"], blob["aux/cls1"], blob["aux/cls2"], blob["aux/labels"] = ibs.numpy(), c1.numpy(), c2.numpy(), lab.numpy()
    blob["aux/sym"] = model.compute_symmetry_loss(c1, c2).detach().numpy()
    blob["aux/align"] = model.compute_ibs_alignment_loss(ibs, c1, c2).detach().numpy()
    blob["aux/contrastive"] = model.compute_ibs_contrastive_loss(ibs, lab).detach().numpy()
    blob["aux/contrastive_nopos"] = model.compute_ibs_contrastive_loss(ibs[:3], torch.tensor([0, 1, 2])).detach().numpy()
    # default initialisation under torch.manual_seed(42): per-tensor checksums pin the RNG consumption order
    torch.manual_seed(42)
    fresh = model_mod.DualEEGTransformer(**kw)
    blob["init42/sum"] = np.array([float(v.double().sum()) for v in fresh.state_dict().values()])
    blob["init42/abs"] = np.array([float(v.double().abs().sum()) for v in fresh.state_dict().values()])
    blob["cfg_json"] = np.array(repr(kw))
    path = out_dir / f"{name}.npz"
    np.savez_compressed(path, **{k: (v if isinstance(v, np.ndarray) else np.asarray(v)) for k, v in blob.items()})
    print(f"{name}: {path.stat().st_size / 1e6:.2f} MB, params={sum(p.numel() for p in model.parameters())}, "
          f"logits[0]={out['logits'][0].detach().numpy()}")


def run_fuzzy(fz_mod, out_dir: Path):
    """FuzzyGatingFusion known answers (3_Models/fusion/fuzzy_gating_fusion.py:297-390) for all 4 modes."""
    blob = {}
    g = torch.Generator().manual_seed(11)
    zi = torch.randn(16, 3, generator=g) * 2
    ze = torch.randn(16, 3, generator=g) * 2
    zi[0] = 0.0; ze[0] = 0.0                    # uniform/uniform
    zi[1] = torch.tensor([10.0, 0, 0]); ze[1] = 0.0
    zi[2] = 0.0; ze[2] = torch.tensor([10.0, 0, 0])
    blob["z_img"], blob["z_eeg"] = zi.numpy(), ze.numpy()
    modes = []
    for mode in fz_mod.FuzzyGatingFusion.VALID_MODES:
        m = fz_mod.FuzzyGatingFusion(num_classes=3, mode=mode)
        m.eval()
        with torch.no_grad():
            res = m(zi, ze)
        modes.append(mode)
        blob[f"{mode}/z_fused"] = res[0].numpy()
        blob[f"{mode}/alpha"] = res[1].numpy()
        for n, p in m.state_dict().items():
            blob[f"{mode}/state/{n}"] = p.numpy()
    blob["modes"] = np.array(modes)
    np.savez_compressed(out_dir / "fuzzy_gating.npz", **blob)
    print("fuzzy modes:", modes)


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    out_dir = REPO / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    model_mod, gen_mod, fz_mod = load_reference()
    only = sys.argv[1:]
    for name, kw in CONFIGS.items():
        if only and name not in only:
            continue
        run_config(name, kw, model_mod, gen_mod, out_dir)
    if not only or "fuzzy" in only:
        run_fuzzy(fz_mod, out_dir)


if __name__ == "__main__":
    main()
