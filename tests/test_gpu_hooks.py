"""GPU parity tests for the analysis-hook contract: the forward hooks the reference's analysis
code registers (5_Metrics/eeg_metrics.py:195-205 capture, :335-343 band masking, :433-452 attention probabilities) must see
and cause on the HIP module what they see and cause on the reference.  Expected values: tests/golden/hooks.npz, produced
from the reference by oracle/make_golden_hooks.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import GOLDEN, load_golden, t  # noqa: E402
from tests.test_gpu_model import DEV, build  # noqa: E402

HZ = np.load(GOLDEN / "hooks.npz", allow_pickle=False)


@pytest.mark.parametrize("dtype,atol", [("f32", 2e-5), ("bf16", 2e-2)])
def test_cross_attention_probability_hook(dtype, atol):
    z, kw, cfg, sd, model = build("cfg3_xattn", dtype)
    model.eval()
    x1, x2 = t(z["gen_eeg/eeg1"]).to(DEV), t(z["gen_eeg/eeg2"]).to(DEV)
    seen = []
    h = model.cross_attn.cross_attn.dropout.register_forward_hook(lambda m, i, o: seen.append(i[0].detach().cpu()))
    with torch.no_grad():
        out = model(x1, x2)
    h.remove()
    assert len(seen) == 2 and tuple(seen[0].shape) == (4, 8, 65, 65)      # direction 1 (q = stream 1), then direction 2
    got = torch.stack(seen)[:, :2].numpy()
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=2e-3 if dtype == "bf16" else 1e-5)
    np.testing.assert_allclose(got, HZ["xattn_probs"], atol=atol)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), HZ["xattn_logits"], atol=3e-2 if dtype == "bf16" else 1e-4)
    # no hook registered -> the carrier is skipped entirely
    seen.clear()
    with torch.no_grad():
        model(x1, x2)
    assert not seen
    # encoder self-attention modules honour the same contract (once per stream)
    h = model.encoder.layers[0].mha.dropout.register_forward_hook(lambda m, i, o: seen.append(i[0].detach().cpu()))
    with torch.no_grad():
        model(x1, x2)
    h.remove()
    assert len(seen) == 2
    np.testing.assert_allclose(torch.stack(seen)[:, :1].numpy(), HZ["self0_probs"], atol=atol)


def _conn_close(got, ref):
    T = 1024
    for f in range(7):
        dlt = np.abs(got[:, :, f] - ref[:, :, f])
        if f in (1, 2):   # sign()-based: one flipped sample of T moves PLI by 2/T
            assert (dlt > 1e-4).mean() < 5e-3 and dlt.max() < 6.5 / T, (f, dlt.max())
        else:
            assert dlt.max() < 1e-4, (f, dlt.max())


def test_ibs_matrix_generator_hooks():
    z, kw, cfg, sd, model = build("a5_full", "f32")
    model.eval()
    x1, x2 = t(z["gen_eeg/eeg1"]).to(DEV), t(z["gen_eeg/eeg2"]).to(DEV)
    # capture hook (IBSMatrixExtractor): output [B, 6, 7, C, C]
    seen = []
    h = model.ibs_matrix_generator.register_forward_hook(lambda m, i, o: seen.append(o.detach().cpu().numpy()))
    with torch.no_grad():
        base = model(x1, x2)["logits"].cpu().numpy()
        direct = model.ibs_matrix_generator(x1, x2).cpu().numpy()       # stand-alone call, as the reference allows
    h.remove()
    assert len(seen) == 2 and seen[0].shape == (4, 6, 7, 8, 8)
    _conn_close(seen[0], HZ["ibs_conn"])
    assert np.array_equal(direct, seen[0])
    np.testing.assert_allclose(base, HZ["ibs_base_logits"], atol=2e-4)
    # band-masking hook (eeg_metrics.py:335-343) edits the output in place and returns it
    for band in range(6):
        def mask(m, i, o, band=band):
            o[:, band, :, :, :] = 0
            return o
        h = model.ibs_matrix_generator.register_forward_hook(mask)
        with torch.no_grad():
            got = model(x1, x2)["logits"].cpu().numpy()
        h.remove()
        np.testing.assert_allclose(got, HZ["ibs_masked_logits"][band], atol=2e-4, err_msg=f"band {band}")
    # replacement-returning hook
    h = model.ibs_matrix_generator.register_forward_hook(lambda m, i, o: o * 0.5)
    with torch.no_grad():
        got = model(x1, x2)["logits"].cpu().numpy()
    h.remove()
    np.testing.assert_allclose(got, HZ["ibs_halved_logits"], atol=2e-4)
    # hooks removed -> baseline again
    with torch.no_grad():
        np.testing.assert_allclose(model(x1, x2)["logits"].cpu().numpy(), base, atol=1e-6)


@pytest.mark.parametrize("dtype,rtol", [("f32", 2e-3), ("bf16", 6e-2)])
def test_gradcam_hooks_on_spec_conv(dtype, rtol):
    """The reference's GradCAM class (eeg_metrics.py:742-830) on the HIP module: forward hook + full backward hook on
    spectrogram_generator.spec_conv[3], parameters frozen, inputs marked requires_grad, score = sum of predicted logits.
    Expected tensors: tests/golden/gradcam.npz (oracle/make_golden_gradcam.py, from the reference)."""
    gz = np.load(GOLDEN / "gradcam.npz", allow_pickle=False)
    z, kw, cfg, sd, model = build("cfg5_a2_spec", dtype)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    x1 = t(z["gen_eeg/eeg1"]).to(DEV).requires_grad_(True)
    x2 = t(z["gen_eeg/eeg2"]).to(DEV).requires_grad_(True)
    acts, grads = [], []
    layer = model.spectrogram_generator.spec_conv[3]
    h1 = layer.register_forward_hook(lambda m, i, o: acts.append(o.detach()))
    h2 = layer.register_full_backward_hook(lambda m, gi, go: grads.append(go[0].detach()))
    model.zero_grad()
    out = model(x1, x2)
    logits = out["logits"]
    pred = logits.argmax(-1)
    assert (pred.cpu().numpy() == gz["pred"]).all()
    logits[torch.arange(logits.shape[0], device=DEV), pred].sum().backward()
    h1.remove(), h2.remove()
    assert len(acts) == 2 and len(grads) == 2 and tuple(acts[0].shape) == tuple(gz["shape"])
    C = cfg.in_channels
    for s_ in range(2):
        a, ra = acts[s_][:C].cpu().numpy(), gz["act"][s_]
        assert np.abs(a - ra).max() <= rtol * np.abs(ra).max(), (s_, np.abs(a - ra).max())
        g, rg = grads[s_][:C].cpu().numpy(), gz["grad_in_hook_order"][s_]           # stream 2 first, as under autograd
        if dtype == "f32":
            assert np.abs(g - rg).max() <= rtol * np.abs(rg).max() + 1e-9, (s_, np.abs(g - rg).max(), np.abs(rg).max())
        else:   # bf16 gradients through six encoder layers: held to direction and scale, as the parameter gradients are
            cos = float((g * rg).sum() / (np.linalg.norm(g) * np.linalg.norm(rg)))
            assert cos > 0.8 and 0.7 < np.linalg.norm(g) / np.linalg.norm(rg) < 1.4, (s_, cos)
    # the CAM the reference's generate_cam computes from them (gradients reversed into forward order)
    w = torch.stack(grads).mean(dim=(3, 4), keepdim=True)
    cam = torch.relu((w * torch.stack(acts[::-1])).sum(2))[:, :C].cpu().numpy()
    ref = gz["cam_stream2_then_1"]
    if dtype == "f32":
        assert np.abs(cam - ref).max() <= 2 * rtol * np.abs(ref).max() + 1e-9
    else:
        assert float(np.corrcoef(cam.ravel(), ref.ravel())[0, 1]) > 0.8
    # without hooks the carriers are skipped and frozen parameters get no gradient
    assert all(p.grad is None for p in model.parameters()) and x1.grad is None
