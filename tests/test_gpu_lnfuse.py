"""LayerNorm as the tail of the launch that completes its input rows (csrc/common.h eg_epilogue_layernorm256):
  * eg_attn_block_fwd with ln_out (norm1, A:293) and eg_ffn_chain with ln_out (norm2, A:295) against the same launch followed by
    eg_layernorm_fwd: every other output bit-identical, the normalised rows within one rounding step of the storage type and the
    statistics within 2e-6 relative (the row sums are formed in another order);
  * against fp64 LayerNorm of the stored rows;
  * a whole training step with the fused norms against EYEGAZE_LN_FUSE=0: logits, loss and gradients agree to rounding."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import call, ptr  # noqa: E402
from tests.test_gpu_ops import DEV, DT  # noqa: E402

D = 256


def ln_standalone(x, gamma, beta, dtype):
    M = x.shape[0]
    y = torch.zeros_like(x)
    stats = torch.zeros(M, 2, device=DEV)
    call("eg_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), M, D, dtype, 0)
    torch.cuda.synchronize()
    return y, stats


def check_ln(got_y, got_st, rows, gamma, beta, dtype, rows_valid=None):
    """rows: the stored LayerNorm input [M, 256]"""
    ref_y, ref_st = ln_standalone(rows, gamma, beta, dtype)
    sel = slice(None) if rows_valid is None else rows_valid
    torch.testing.assert_close(got_st[sel], ref_st[sel], rtol=2e-6, atol=2e-6)
    ulp = 2.0 ** -7 if dtype == L.EG_BF16 else 2.0 ** -10       # one rounding step of the storage type, relative (just above a power of two)
    a, b = got_y[sel].float(), ref_y[sel].float()
    assert float(((a - b).abs() / b.abs().clamp_min(0.25)).max()) <= ulp * 1.01
    assert float((a != b).float().mean()) < 0.02                 # ... and rare
    x64 = rows[sel].double()
    mu, var = x64.mean(-1, keepdim=True), x64.var(-1, unbiased=False, keepdim=True)
    y64 = (x64 - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    torch.testing.assert_close(a.double(), y64, rtol=2 * ulp, atol=2 * ulp)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("case", [(8, 65, 0.1), (5, 73, 0.0), (2, 80, 0.1), (3, 16, 0.0), (520, 65, 0.1)])
def test_attention_block_with_norm1(case, dtype):
    from tests.test_gpu_attnblock import frag_weights, one_launch, operands
    NB, S, p = case
    o = operands(NB, S, dtype, seed=7)
    ref = one_launch(o, NB, S, dtype, p)
    g = torch.Generator().manual_seed(NB + S)
    gamma = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    t = DT[dtype]
    M = NB * S
    wqkv, wo = frag_weights(o, dtype)
    qkv = torch.full((M, 3 * D), 7.0, device=DEV, dtype=t)
    ctx, r1, y = (torch.full((M, D), 7.0, device=DEV, dtype=t) for _ in range(3))
    lse = torch.full((NB, 8, S), 7.0, device=DEV)
    st = torch.full((M, 2), 7.0, device=DEV)
    d = L.AttnBlockDesc()
    d.x, d.wqkv_frag, d.wo_frag, d.bqkv, d.bo = ptr(o["x"]), ptr(wqkv), ptr(wo), ptr(o["bqkv"]), ptr(o["bo"])
    d.qkv, d.ctx, d.lse, d.r1, d.state = ptr(qkv), ptr(ctx), ptr(lse), ptr(r1), ptr(o["st"])
    d.NB, d.S, d.d_model, d.num_heads, d.dtype = NB, S, D, 8, dtype
    d.attn_drop_p, d.attn_drop_site, d.out_drop_p, d.out_drop_site = p, 21, p, 22
    d.ln_gamma, d.ln_beta, d.ln_out, d.ln_stats = ptr(gamma), ptr(beta), ptr(y), ptr(st)
    call("eg_attn_block_fwd", C.byref(d), 0)
    torch.cuda.synchronize()
    for name, a, b in zip(("qkv", "lse", "ctx", "r1"), (qkv, lse, ctx, r1), ref):
        assert torch.equal(a, b), name                           # the norm changes nothing else
    check_ln(y, st, r1, gamma, beta, dtype)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("case", [(33280, 1024, 0.1), (4160, 1024, 0.0), (1000, 512, 0.1), (81, 128, 0.0)])
def test_ffn_chain_with_norm2(case, dtype):
    import tests.test_gpu_ffn as TF
    M, F, p = case
    o = TF.operands(M, F, dtype, seed=13)
    Href, Cref = TF.one_launch(o, M, F, dtype, "fwd", p)
    g = torch.Generator().manual_seed(M + F)
    gamma = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    t = DT[dtype]
    H = torch.full((M, F), 7.0, device=DEV, dtype=t)
    Cc, y = (torch.full((M, D), 7.0, device=DEV, dtype=t) for _ in range(2))
    st = torch.full((M, 2), 7.0, device=DEV)
    f = L.FfnDesc()
    w1f, w2f = TF.frag_pack(o["W1"], 3, dtype), TF.frag_pack(o["W2"], 5, dtype)
    f.A, f.W1, f.W2, f.H, f.C, f.state = ptr(o["A"]), ptr(w1f), ptr(w2f), ptr(H), ptr(Cc), ptr(o["st"])
    f.lda, f.ldh, f.ldc, f.ldg, f.ldr, f.M, f.F, f.dtype = D, F, D, F, D, M, F, dtype
    f.bias1, f.bias2, f.act1, f.residual = ptr(o["b1"]), ptr(o["b2"]), L.ACT_RELU, ptr(o["A"])
    f.drop_h_p, f.drop_h_site, f.drop_c1_p, f.drop_c1_site, f.drop_c2_p, f.drop_c2_site = p, 21, p, 22, p, 23
    f.ln_gamma, f.ln_beta, f.ln_out, f.ln_stats = ptr(gamma), ptr(beta), ptr(y), ptr(st)
    call("eg_ffn_chain", C.byref(f), 0)
    torch.cuda.synchronize()
    assert torch.equal(H, Href) and torch.equal(Cc, Cref)        # the norm changes nothing else
    check_ln(y, st, Cc, gamma, beta, dtype)


def test_fused_norm_argument_checks():
    f = L.FfnDesc()
    x = torch.zeros(128, D, device=DEV, dtype=torch.bfloat16)
    h = torch.zeros(128, 128, device=DEV, dtype=torch.bfloat16)
    f.A, f.W1, f.W2, f.H, f.C = ptr(x), ptr(h), ptr(h), ptr(h), ptr(x)
    f.lda, f.ldh, f.ldc, f.M, f.F, f.dtype = D, 128, D, 128, 128, L.EG_BF16
    f.ln_out = ptr(x)                                            # no gain / bias, not the forward form
    with pytest.raises(L.EgError, match="fused LayerNorm"):
        call("eg_ffn_chain", C.byref(f), 0)


@pytest.mark.parametrize("name,dtype", [("cfg3_xattn", "bf16"), ("cfg3_xattn", "fp16")])
def test_training_step_with_and_without_the_fused_norms(name, dtype, monkeypatch):
    from eyegaze_multimodal_amd import HipAdamW
    from tests.helpers import t
    from tests.test_gpu_model import build
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("EYEGAZE_LN_FUSE", flag)
        z, kw, cfg, sd, model = build(name, dtype)
        model.train()
        x1, x2, labels = t(z["randn/eeg1"]).to(DEV), t(z["randn/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
        eng = model.engine(x1.shape[0], x1.shape[2], torch.device(DEV))
        assert eng.ln_fuse == (flag == "1") and eng.attn_block and eng.fuse_ffn
        opt = HipAdamW(model)
        opt.begin_step(eng, seed=11)
        before = L.CALLS
        eng.forward(x1, x2, labels, train=True)
        nfwd = L.CALLS - before
        eng.backward(gloss=torch.ones(1, device=DEV))
        torch.cuda.synchronize()
        res[flag] = (eng.a["logits"].float().clone(), eng.a["loss"].float().clone(), model._flat.grad.clone(), nfwd)
    assert res["0"][3] - res["1"][3] == 2 * cfg.num_layers      # two launches fewer per encoder layer
    lg1, ls1, g1, _ = res["1"]
    lg0, ls0, g0, _ = res["0"]
    assert torch.isfinite(g1).all()
    tol = 2e-2 if dtype == "bf16" else 4e-3
    torch.testing.assert_close(lg1, lg0, rtol=tol, atol=tol)
    torch.testing.assert_close(ls1, ls0, rtol=tol, atol=tol)
    # The forward pass agrees to the dtype's rounding (fp16: 0.01 % of the first normalised rows differ by one step, logits by 5e-4);
    # the GRADIENTS of this 4-sample fixture at random initialisation move by percents under any such perturbation (the fp32 oracle's
    # own gradients move 2-35 % when its weights are rounded to bf16: test_gpu_model.test_bf16_gradients_track_the_oracle).
    # Measured: 3.3e-2 (bf16), 2.3e-2 (fp16); gates = 1.25 x measured.
    rel = float((g1 - g0).norm() / g0.norm())
    print(f"fused-norm gradient distance {dtype}: {rel:.3e}")
    assert rel < (4.2e-2 if dtype == "bf16" else 2.9e-2), rel
