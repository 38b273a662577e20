"""eg_attn_block_fwd (csrc/attnblock.hip): the attention half of an encoder layer as ONE launch with a workgroup per window
(q|k|v projection A:203-205, attention core A:206-212, out-proj + dropout + residual A:213 / A:292-293).
  * bit-identical to the three launches it replaces (eg_gemm_nt -> eg_attention_fwd -> eg_gemm_nt) on q|k|v, lse, ctx and r1,
    dropout on and off, S in {16 .. 80}, bf16 and fp16;
  * against fp64 torch on the rounded operands (catches an error common to both paths);
  * through the engine: a whole training step (forward, backward, every gradient) is bit-identical with the block on and off."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import call, ptr  # noqa: E402
from tests.test_gpu_ops import DEV, DT, dev_state, gemm_nt  # noqa: E402

D, H = 256, 8


def operands(NB, S, dtype, seed):
    g = torch.Generator().manual_seed(seed + 31 * NB + S)
    t = DT[dtype]
    M = NB * S
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    o = dict(x=mk(M, D).to(t).to(DEV), wq=mk(D, D, sc=0.08).to(DEV), wk=mk(D, D, sc=0.08).to(DEV), wv=mk(D, D, sc=0.08).to(DEV),
             wo=mk(D, D, sc=0.06).to(DEV), bqkv=(mk(3 * D, sc=0.1)).to(DEV), bo=mk(D, sc=0.1).to(DEV), st=dev_state(seed=91 + seed))
    return o


def frag_weights(o, dtype):
    """wqkv / wo in eg_attn_block_fwd's fragment order, written by eg_pack_table from the fp32 parameters (modes 7 / 8)"""
    wqkv = torch.zeros(3 * D * D, device=DEV, dtype=DT[dtype])
    wo = torch.zeros(D * D, device=DEV, dtype=DT[dtype])
    e = (L.PackEntry * 4)()
    for i, (src, dst, mode, part) in enumerate(((o["wq"], wqkv, 7, 0), (o["wk"], wqkv, 7, 1), (o["wv"], wqkv, 7, 2), (o["wo"], wo, 8, 0))):
        e[i].src, e[i].dst, e[i].rows, e[i].cols, e[i].ldd, e[i].mode, e[i].blk0, e[i].nblk = ptr(src), ptr(dst), D, D, part, mode, 32 * i, 32
    tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).to(DEV)
    call("eg_pack_table", ptr(tab), 4, 128, dtype, 0)
    torch.cuda.synchronize()
    return wqkv, wo


def three_launches(o, NB, S, dtype, p):
    t = DT[dtype]
    M = NB * S
    w = torch.cat([o["wq"], o["wk"], o["wv"]]).to(t).contiguous()
    qkv = gemm_nt(o["x"], w, M, 3 * D, D, dtype, bias=o["bqkv"])
    ctx = torch.zeros(M, D, device=DEV, dtype=t)
    lse = torch.zeros(NB, H, S, device=DEV)
    call("eg_attention_fwd", ptr(qkv), ptr(ctx), ptr(lse), NB, S, H, 0, dtype, p, 21, ptr(o["st"]), 0)
    r1 = gemm_nt(ctx, o["wo"].to(t).contiguous(), M, D, D, dtype, bias=o["bo"], residual=o["x"], drop1=(p, 22), state=o["st"])
    torch.cuda.synchronize()
    return qkv, lse, ctx, r1


def one_launch(o, NB, S, dtype, p):
    t = DT[dtype]
    M = NB * S
    wqkv, wo = frag_weights(o, dtype)
    qkv = torch.full((M, 3 * D), 7.0, device=DEV, dtype=t)
    ctx = torch.full((M, D), 7.0, device=DEV, dtype=t)
    r1 = torch.full((M, D), 7.0, device=DEV, dtype=t)
    lse = torch.full((NB, H, S), 7.0, device=DEV)
    d = L.AttnBlockDesc()
    d.x, d.wqkv_frag, d.wo_frag, d.bqkv, d.bo = ptr(o["x"]), ptr(wqkv), ptr(wo), ptr(o["bqkv"]), ptr(o["bo"])
    d.qkv, d.ctx, d.lse, d.r1, d.state = ptr(qkv), ptr(ctx), ptr(lse), ptr(r1), ptr(o["st"])
    d.NB, d.S, d.d_model, d.num_heads, d.dtype = NB, S, D, H, dtype
    d.attn_drop_p, d.attn_drop_site, d.out_drop_p, d.out_drop_site = p, 21, p, 22
    call("eg_attn_block_fwd", C.byref(d), 0)
    torch.cuda.synchronize()
    return qkv, lse, ctx, r1


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("case", [(3, 65, 0.0), (8, 65, 0.1), (5, 73, 0.1), (2, 80, 0.1), (4, 49, 0.0), (3, 16, 0.1), (1, 1, 0.0), (520, 65, 0.1)])
def test_block_is_bit_identical_to_the_three_launches(case, dtype):
    NB, S, p = case
    o = operands(NB, S, dtype, seed=3)
    ref = three_launches(o, NB, S, dtype, p)
    got = one_launch(o, NB, S, dtype, p)
    for name, a, b in zip(("qkv", "lse", "ctx", "r1"), got, ref):
        assert torch.isfinite(a.float()).all(), name
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()), int((a != b).sum()))


@pytest.mark.parametrize("S", [65, 80])
def test_block_matches_fp64(S):
    dtype, NB = L.EG_BF16, 4
    o = operands(NB, S, dtype, seed=5)
    qkv, lse, ctx, r1 = one_launch(o, NB, S, dtype, 0.0)
    t = DT[dtype]
    x = o["x"].double().cpu()
    w = torch.cat([o["wq"], o["wk"], o["wv"]]).to(t).double().cpu()
    qkv_ref = x @ w.T + o["bqkv"].double().cpu()
    torch.testing.assert_close(qkv.double().cpu(), qkv_ref, rtol=1e-2, atol=2e-2)
    qr = qkv.double().cpu().view(NB, S, 3, H, 32)            # downstream stages from the STORED (rounded) q|k|v
    q, k, v = qr[:, :, 0].transpose(1, 2), qr[:, :, 1].transpose(1, 2), qr[:, :, 2].transpose(1, 2)
    sc = q @ k.transpose(-1, -2) / math.sqrt(32)
    torch.testing.assert_close(lse.double().cpu(), torch.logsumexp(sc, -1), rtol=1e-5, atol=1e-5)
    ctx_ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(NB * S, D)
    torch.testing.assert_close(ctx.double().cpu(), ctx_ref, rtol=1e-2, atol=1e-2)
    r1_ref = x + ctx.double().cpu() @ o["wo"].to(t).double().cpu().T + o["bo"].double().cpu()
    torch.testing.assert_close(r1.double().cpu(), r1_ref, rtol=1e-2, atol=2e-2)


def test_block_argument_checks():
    d = L.AttnBlockDesc()
    with pytest.raises(L.EgError, match="null operand"):
        call("eg_attn_block_fwd", C.byref(d), 0)
    assert L.lib().eg_attn_block_ok(65, 256, 8, L.EG_BF16) == 1
    assert L.lib().eg_attn_block_ok(81, 256, 8, L.EG_BF16) == 0        # longer windows keep the three-launch path
    assert L.lib().eg_attn_block_ok(65, 256, 8, L.EG_F32) == 0
    assert L.lib().eg_attn_block_ok(65, 128, 4, L.EG_BF16) == 0


@pytest.mark.parametrize("name,dtype", [("cfg3_xattn", "bf16"), ("cfg5_a2_spec", "bf16"), ("cfg3_xattn", "fp16")])
def test_training_step_is_bit_identical_with_and_without_the_block(name, dtype, monkeypatch):
    """whole engine step in TRAIN mode (dropout on): logits, loss and every gradient must not depend on whether the attention half
    ran as one launch or as three"""
    from eyegaze_multimodal_amd import HipAdamW
    from tests.helpers import t
    from tests.test_gpu_model import build
    res = {}
    monkeypatch.setenv("EYEGAZE_LN_FUSE", "0")       # (the fused LayerNorm sums its statistics in another order: tests/test_gpu_lnfuse.py)
    for flag in ("1", "0"):
        monkeypatch.setenv("EYEGAZE_ATTN_BLOCK", flag)
        z, kw, cfg, sd, model = build(name, dtype)
        model.train()
        x1, x2, labels = t(z["randn/eeg1"]).to(DEV), t(z["randn/eeg2"]).to(DEV), t(z["labels"]).to(DEV)
        eng = model.engine(x1.shape[0], x1.shape[2], torch.device(DEV))
        assert eng.attn_block == (flag == "1")
        opt = HipAdamW(model)
        opt.begin_step(eng, seed=11)
        eng.forward(x1, x2, labels, train=True)
        eng.backward(gloss=torch.ones(1, device=DEV))
        torch.cuda.synchronize()
        res[flag] = (eng.a["logits"].clone(), eng.a["loss"].clone(), model._flat.grad.clone(), eng.a["r1_0"].clone(), eng.a["ctx0"].clone())
    for a, b in zip(res["1"], res["0"]):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
