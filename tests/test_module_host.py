"""CPU-only host-logic tests of the drop-in module: constructor / state_dict contract,
initialisation order, flat parameter buffer mechanics, and refusal to run without a device."""
import numpy as np
import pytest
import torch

from eyegaze_multimodal_amd import DualEEGTransformer, EgError
from eyegaze_multimodal_amd.engine import FlatParams
from oracle.dual_eeg_oracle import ModelCfg, state_shapes
from tests.helpers import ALL_CONFIGS, load_golden


@pytest.mark.parametrize("name", ALL_CONFIGS)
def test_state_dict_contract(name):
    z, kw, cfg, sd = load_golden(name)
    model = DualEEGTransformer(**kw)
    msd = model.state_dict()
    assert list(msd.keys()) == [str(k) for k in z["state_keys"]]
    for k, v in sd.items():
        assert tuple(msd[k].shape) == tuple(v.shape), k
    model.load_state_dict(sd, strict=True)
    assert [n for n, _ in model.named_parameters()] == [str(n) for n in z["randn/grad/names"]]


@pytest.mark.parametrize("name", ["cfg3_xattn", "a5_full", "a3_ibs_scalar", "tiny_full"])
def test_default_init_consumes_rng_like_the_reference(name):
    z, kw, cfg, sd = load_golden(name)
    torch.manual_seed(42)
    model = DualEEGTransformer(**kw)
    sums = np.array([float(v.double().sum()) for v in model.state_dict().values()])
    absum = np.array([float(v.double().abs().sum()) for v in model.state_dict().values()])
    np.testing.assert_allclose(sums, z["init42/sum"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(absum, z["init42/abs"], rtol=1e-6, atol=1e-6)


def test_reference_default_ctor_shapes():
    """Default kwargs = the reference's (in_channels=62 ...); the yaml's C=32 model has 140 tensors / 8,110,022 params."""
    m = DualEEGTransformer(in_channels=32, max_len=256)
    assert sum(p.numel() for p in m.parameters()) == 8110022
    assert len(m.state_dict()) == 140
    assert hasattr(m, "ibs_matrix_generator") and hasattr(m.cross_attn.cross_attn, "dropout")
    assert isinstance(m.spectrogram_generator.spec_conv[3], torch.nn.Conv2d)


def test_flat_params_are_views_and_survive_load_state_dict():
    z, kw, cfg, sd = load_golden("tiny_a1")
    model = DualEEGTransformer(**kw)
    fp: FlatParams = model._flat
    fp.ensure(torch.device("cpu"))
    base = fp.flat.data_ptr()
    for n, p in model.named_parameters():
        assert p.data_ptr() == base + 4 * fp.offsets[n]
        assert fp.offsets[n] % 4 == 0
    model.load_state_dict(sd, strict=True)
    o = fp.offsets["classifier.3.bias"]
    assert torch.equal(fp.flat[o:o + 3], sd["classifier.3.bias"])
    flat_before = fp.flat
    fp.ensure(torch.device("cpu"))
    assert fp.flat is flat_before  # idempotent
    model.float()  # nn.Module._apply keeps .data in place for no-op casts; views must still be intact or rebuilt
    fp.ensure(torch.device("cpu"))
    for n, p in model.named_parameters():
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * fp.offsets[n]


def test_cpu_forward_is_refused():
    model = DualEEGTransformer(in_channels=8, max_len=256, use_spectrogram=False, use_ibs=False)
    with pytest.raises(EgError, match="no CPU fallback"):
        model(torch.zeros(2, 8, 1024), torch.zeros(2, 8, 1024))


def test_bad_dtype_name():
    with pytest.raises(ValueError):
        DualEEGTransformer(in_channels=8, compute_dtype="fp8")


def test_aux_losses_refuse_cpu_tensors():
    """The auxiliary losses are HIP kernels too (csrc/aux.hip): no CPU fallback.  Their parity is in tests/test_gpu_model.py."""
    from eyegaze_multimodal_amd._lib import EgError
    z, kw, cfg, sd = load_golden("cfg3_xattn")
    m = DualEEGTransformer(**kw)
    ibs, c1, c2 = (torch.from_numpy(z[f"aux/{k}"]) for k in ("ibs", "cls1", "cls2"))
    for fn in (lambda: m.compute_symmetry_loss(c1, c2), lambda: m.compute_ibs_alignment_loss(ibs, c1, c2),
               lambda: m.compute_ibs_contrastive_loss(ibs, torch.from_numpy(z["aux/labels"]))):
        with pytest.raises(EgError):
            fn()
