"""GPU parity tests, op level: every C-ABI entry point against a plain torch fp32/fp64 CPU statement of the same
op on the same seeded inputs.  bf16 kernels are compared with the reference evaluated on the bf16-rounded
inputs (so only accumulation order / output rounding differ); tolerances are written next to each check."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import GemmDesc, GemmTNDesc, StepState, call, ptr, rowmap  # noqa: E402

DEV = "cuda"
DT = {L.EG_BF16: torch.bfloat16, L.EG_F16: torch.float16, L.EG_F32: torch.float32}


def dev_state(seed=1234, lr=1e-4, step=1, grad_scale=1.0):
    st = StepState(seed & 0xFFFFFFFF, seed >> 32, lr, 1 - 0.9 ** step, 1 - 0.999 ** step, grad_scale, 1.0, 0.0, 1.0)
    host = torch.zeros(L.STATE_WORDS, dtype=torch.int32)
    C.memmove(host.data_ptr(), C.addressof(st), C.sizeof(st))
    return host.to(DEV)


def read_state(t):
    st = StepState()
    h = t.cpu()
    C.memmove(C.addressof(st), h.data_ptr(), C.sizeof(st))
    return st


def gemm_nt(A, W, M, N, K, dtype, *, out=None, a=None, c=None, r=None, p=None, bias=None, residual=None, gate=None,
            out_pre=None, act=0, drop1=(0.0, 0), drop2=(0.0, 0), gate_scale=1.0, state=None, ldw=None):
    if out is None:
        out = torch.zeros(M, N, device=DEV, dtype=DT[dtype])
    d = GemmDesc()
    d.A, d.W, d.C = ptr(A), ptr(W), ptr(out)
    d.bias, d.residual, d.gate, d.out_pre = ptr(bias) or None, ptr(residual) or None, ptr(gate) or None, ptr(out_pre) or None
    d.state = ptr(state) or None
    d.a, d.c = a or rowmap(K), c or rowmap(N)
    d.r, d.p = r or d.c, p or d.c
    d.M, d.N, d.K, d.ldw, d.act, d.dtype = M, N, K, ldw or K, act, dtype
    d.drop1_p, d.drop1_site = drop1
    d.drop2_p, d.drop2_site = drop2
    d.gate_scale = gate_scale
    call("eg_gemm_nt", C.byref(d), 0)
    torch.cuda.synchronize()
    return out


def tol(dtype, k=1.0):
    # bf16: output rounding 2^-9 relative + fp32 accumulation; f32: accumulation order only
    return dict(rtol=1.0e-2 * k, atol=2e-2 * k) if dtype != L.EG_F32 else dict(rtol=2e-5 * k, atol=2e-5 * k)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
@pytest.mark.parametrize("shape", [(300, 136, 128), (520, 768, 256), (128, 128, 64), (33, 8, 1024), (2500, 64, 384), (1100, 40, 128)])
def test_gemm_nt_plain(dtype, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(DT[dtype])
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DT[dtype])
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(DT[dtype])
    ref = torch.relu(A.double() @ W.double().T + bias.double()) + res.double()
    Ad, Wd, bd, rd = A.to(DEV), W.to(DEV), bias.to(DEV), res.to(DEV)
    out = gemm_nt(Ad, Wd, M, N, K, dtype, bias=bd, residual=rd, act=L.ACT_RELU)
    torch.testing.assert_close(out.cpu().double(), ref, **tol(dtype))


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
def test_gemm_nt_asymmetric_identity(dtype):
    """A = I with an asymmetric W catches a transposed accumulator write (guide: A=I check)."""
    K = N = 128
    M = 128
    A = torch.eye(M, K).to(DT[dtype])
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125).to(DT[dtype])
    Ad, Wd = A.to(DEV), W.to(DEV)
    out = gemm_nt(Ad, Wd, M, N, K, dtype)
    torch.testing.assert_close(out.cpu().float(), W.float().T.contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
def test_conv1d_as_grouped_gemm(dtype):
    """Strided Conv1d(k=25, s=4, p=12)+ReLU on channel-last padded rows == F.conv1d (D:154-171)."""
    NB, Cc, T, dm, k, s = 3, 8, 256, 64, 25, 4
    pad = k // 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(NB, Cc, T, generator=g)
    w = torch.randn(dm, Cc, k, generator=g) / math.sqrt(Cc * k)
    b = torch.randn(dm, generator=g) * 0.1
    bk = 64 if dtype != L.EG_F32 else 32
    K0 = (k * Cc + bk - 1) // bk * bk
    T1 = (T + 2 * pad - k) // s + 1
    Tp = max(T + 2 * pad, s * (T1 - 1) + K0 // Cc + 1)
    Tp = (Tp + 7) // 8 * 8
    xt = torch.zeros(NB, Tp, Cc, device=DEV, dtype=DT[dtype])
    xdev, wdev, bdev = x.to(DEV), w.to(DEV), b.to(DEV)
    call("eg_window_pack", ptr(xdev), ptr(xt), NB, Cc, T, Cc, pad, Tp, dtype, 0)
    ref_xt = torch.zeros(NB, Tp, Cc)
    ref_xt[:, pad:pad + T] = x.transpose(1, 2)
    torch.testing.assert_close(xt.cpu().float(), ref_xt.to(DT[dtype]).float(), rtol=0, atol=0)
    wp = torch.zeros(dm, K0, device=DEV, dtype=DT[dtype])
    call("eg_pack_conv_weight", ptr(wdev), ptr(wp), dm, Cc, k, Cc, K0, dtype, 0)
    out = gemm_nt(xt, wp, NB * T1, dm, K0, dtype, a=rowmap(s * Cc, Tp * Cc, T1), bias=bdev, act=L.ACT_RELU)
    xr = x.to(DT[dtype]).double()
    wr = w.to(DT[dtype]).double()
    ref = torch.relu(torch.nn.functional.conv1d(xr, wr, b.double(), stride=s, padding=pad)).transpose(1, 2).reshape(NB * T1, dm)
    torch.testing.assert_close(out.cpu().double(), ref, **tol(dtype))


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
def test_gemm_nt_narrow_tile_is_bit_identical_to_the_square_tile(dtype):
    """N <= 64 with M >= 1024 runs on 256 x 64 tiles (gemm_nt_kernel NARROW: the spectrogram / image convolutions); below 1024 rows the
    same product runs on the 128 x 128 tile.  Same k-ordered MFMA chains and epilogue: the first 1000 rows of a 3000-row launch must
    equal a 1000-row launch bit for bit -- bias, ReLU, gate, both dropouts, second output and residual included."""
    M, Ms, N, K = 3000, 1000, 64, 384
    g = torch.Generator().manual_seed(31)
    A = torch.randn(M, K, generator=g).to(DT[dtype]).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DT[dtype]).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DT[dtype]).to(DEV)
    gate = torch.randn(M, N, generator=g).to(DT[dtype]).to(DEV)
    st = dev_state(seed=123)
    outs = []
    for m in (M, Ms):
        pre = torch.zeros(m, N, device=DEV, dtype=DT[dtype])
        out = gemm_nt(A[:m].contiguous(), W, m, N, K, dtype, bias=bias, residual=res[:m].contiguous(), gate=gate[:m].contiguous(),
                      gate_scale=1.25, out_pre=pre, act=L.ACT_RELU, drop1=(0.1, 5), drop2=(0.2, 6), state=st)
        outs.append((out, pre))
    assert torch.equal(outs[0][0][:Ms], outs[1][0]) and torch.equal(outs[0][1][:Ms], outs[1][1])
    assert torch.isfinite(outs[0][0].float()).all() and float(outs[0][0][Ms:].float().abs().sum()) > 0


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
def test_gemm_nt_dropout_and_gate(dtype):
    M, N, K = 256, 256, 128
    g = torch.Generator().manual_seed(9)
    A = torch.randn(M, K, generator=g).to(DT[dtype]).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DT[dtype]).to(DEV)
    st = dev_state(seed=77)
    base = gemm_nt(A, W, M, N, K, dtype)
    d1 = gemm_nt(A, W, M, N, K, dtype, drop1=(0.1, 5), state=st)
    d1b = gemm_nt(A, W, M, N, K, dtype, drop1=(0.1, 5), state=st)
    assert torch.equal(d1, d1b)  # counter based: reproducible
    kept = (d1 != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01, kept
    m = d1 != 0
    torch.testing.assert_close(d1[m].float(), (base[m].float() / 0.9), rtol=1e-2, atol=1e-2)
    d2 = gemm_nt(A, W, M, N, K, dtype, drop1=(0.1, 5), drop2=(0.1, 6), state=st)
    assert abs((d2 != 0).float().mean().item() - 0.81) < 0.012
    other = gemm_nt(A, W, M, N, K, dtype, drop1=(0.1, 5), state=dev_state(seed=78))
    assert not torch.equal(other, d1)
    # LayerNorm-backward's masked copy must use the very same mask (same site, idx = m*N+n)
    dy = torch.ones(M, N, device=DEV, dtype=DT[dtype])
    x = torch.randn(M, N, generator=g).to(DT[dtype]).to(DEV)
    stats = torch.zeros(M, 2, device=DEV)
    gam = torch.ones(N, device=DEV)
    y = torch.zeros_like(x)
    bet = torch.zeros(N, device=DEV)
    call("eg_layernorm_fwd", ptr(x), ptr(gam), ptr(bet), ptr(y), ptr(stats), M, N, dtype, 0)
    dx, dxd = torch.zeros_like(x), torch.zeros_like(x)
    part = torch.zeros(64 * 2 * N, device=DEV)
    dyr = torch.randn(M, N, generator=g).to(DT[dtype]).to(DEV)
    call("eg_layernorm_bwd", ptr(dyr), ptr(x), ptr(stats), ptr(gam), ptr(dx), ptr(dxd), ptr(part), 64, 64, M, N, dtype, 0.1, 5,
         0.0, 0, ptr(st), 0)
    torch.cuda.synchronize()
    assert torch.equal(dxd != 0, (d1 != 0) & (dx != 0)) or ((dxd != 0) ^ (d1 != 0)).float().mean() < 1e-3
    # gate: zero where gate <= 0, scaled elsewhere
    gate = torch.randn(M, N, generator=g).to(DT[dtype]).to(DEV)
    gg = gemm_nt(A, W, M, N, K, dtype, gate=gate, gate_scale=1.25)
    ref = torch.where(gate.float() > 0, base.float() * 1.25, torch.zeros_like(base.float()))
    torch.testing.assert_close(gg.float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
@pytest.mark.parametrize("shape", [(1000, 136, 200), (520, 768, 256), (37, 8, 64), (4096, 256, 1024)])
def test_gemm_tn(dtype, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(DT[dtype])
    X = torch.randn(M, K, generator=g).to(DT[dtype])
    for splits in (1, 7):
        part = torch.zeros(splits * N * K, device=DEV)
        d = GemmTNDesc()
        dYd, Xd = dY.to(DEV), X.to(DEV)
        d.dY, d.X, d.partial = ptr(dYd), ptr(Xd), ptr(part)
        d.y, d.x = rowmap(N), rowmap(K)
        d.M, d.N, d.K, d.splits, d.dtype = M, N, K, splits, dtype
        call("eg_gemm_tn", C.byref(d), 0)
        out = torch.zeros(N, K, device=DEV)
        call("eg_reduce_partials", ptr(part), ptr(out), N * K, splits, N * K, 0, 0)
        torch.cuda.synchronize()
        ref = dY.double().T @ X.double()
        torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-4 * math.sqrt(M))
    cs = torch.zeros(16 * N, device=DEV)
    call("eg_colsum", ptr(dYd), rowmap(N), M, N, ptr(cs), 16, dtype, 0)
    o = torch.zeros(N, device=DEV)
    call("eg_reduce_partials", ptr(cs), ptr(o), N, 16, N, 0, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(o.cpu().double(), dY.double().sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
@pytest.mark.parametrize("D", [256, 64])
def test_layernorm(dtype, D):
    M = 777
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.5).to(DT[dtype])
    gam, bet = torch.randn(D, generator=g) * 0.1 + 1, torch.randn(D, generator=g) * 0.1
    dy = torch.randn(M, D, generator=g).to(DT[dtype])
    xd, y, stats = x.to(DEV), torch.zeros(M, D, device=DEV, dtype=DT[dtype]), torch.zeros(M, 2, device=DEV)
    gamd, betd = gam.to(DEV), bet.to(DEV)  # keep device operands alive across the asynchronous launch
    call("eg_layernorm_fwd", ptr(xd), ptr(gamd), ptr(betd), ptr(y), ptr(stats), M, D, dtype, 0)
    xr = x.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    ref.backward(dy.double())
    torch.cuda.synchronize()
    torch.testing.assert_close(y.cpu().double(), ref.detach(), **tol(dtype))
    dx = torch.zeros_like(xd)
    nblk = 32
    part = torch.zeros(nblk * 2 * D, device=DEV)
    dyd = dy.to(DEV)
    call("eg_layernorm_bwd", ptr(dyd), ptr(xd), ptr(stats), ptr(gamd), ptr(dx), 0, ptr(part), nblk, nblk, M, D, dtype, 0.0, 0, 0.0,
         0, 0, 0)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    call("eg_reduce_partials", ptr(part), ptr(dg), D, nblk, 2 * D, 0, 0)
    call("eg_reduce_partials", ptr(part) + 4 * D, ptr(db), D, nblk, 2 * D, 0, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(dx.cpu().double(), xr.grad, **tol(dtype))
    torch.testing.assert_close(dg.cpu().double(), gr.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.cpu().double(), br.grad, rtol=1e-4, atol=1e-3)


def _attn_ref(qkv, NB, S, H, kv_shift, dO=None):
    D = H * 32
    x = qkv.double().reshape(NB, S, 3, H, 32)
    q = x[:, :, 0].permute(0, 2, 1, 3)
    idx = (torch.arange(NB) + kv_shift) % NB
    k = x[idx][:, :, 1].permute(0, 2, 1, 3)
    v = x[idx][:, :, 2].permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) / math.sqrt(32)
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(NB * S, D)
    lse = torch.logsumexp(s, -1)
    return o, lse


@pytest.mark.parametrize("dt16", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("S,kv_shift", [(65, 0), (65, 3), (73, 0), (115, 3), (139, 0), (16, 0), (96, 3)])
def test_attention_fwd_bwd(S, kv_shift, dt16):
    NB, H = 6, 4
    D = H * 32
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(NB * S, 3 * D, generator=g).to(DT[dt16])
    dO = torch.randn(NB * S, D, generator=g).to(DT[dt16])
    qkvd, ctx, lse = qkv.to(DEV), torch.zeros(NB * S, D, device=DEV, dtype=DT[dt16]), torch.zeros(NB, H, S, device=DEV)
    call("eg_attention_fwd", ptr(qkvd), ptr(ctx), ptr(lse), NB, S, H, kv_shift, dt16, 0.0, 0, 0, 0)
    torch.cuda.synchronize()
    qr = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, NB, S, H, kv_shift)
    # P is rounded to bf16 before P*V (8 significant bits): |dO| <= 2^-9 * sum|p v| ~ 4e-3 * |v|
    torch.testing.assert_close(ctx.cpu().double(), o_ref.detach(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse.cpu().double(), lse_ref.detach(), rtol=1e-4, atol=1e-4)
    o_ref.backward(dO.double())
    dqkv = torch.zeros_like(qkvd)
    dOd = dO.to(DEV)
    call("eg_attention_bwd", ptr(qkvd), ptr(ctx), ptr(dOd), ptr(lse), ptr(dqkv), NB, S, H, kv_shift, dt16, 0.0, 0, 0, 0)
    torch.cuda.synchronize()
    got, ref = dqkv.cpu().double(), qr.grad
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err < 3e-2 * max(1.0, scale), (err, scale)
    # relative Frobenius error is the sharper statement
    assert ((got - ref).norm() / ref.norm()).item() < 2e-2


@pytest.mark.parametrize("S,kv_shift", [(65, 0), (115, 3), (139, 0)])
def test_attention_f32_exact(S, kv_shift):
    """EG_F32 attention (plain fmaf chains) against fp64: the tight-parity path."""
    NB, H = 4, 2
    D = H * 32
    g = torch.Generator().manual_seed(S + 1)
    qkv = torch.randn(NB * S, 3 * D, generator=g)
    dO = torch.randn(NB * S, D, generator=g)
    qkvd, ctx, lse = qkv.to(DEV), torch.zeros(NB * S, D, device=DEV), torch.zeros(NB, H, S, device=DEV)
    call("eg_attention_fwd", ptr(qkvd), ptr(ctx), ptr(lse), NB, S, H, kv_shift, L.EG_F32, 0.0, 0, 0, 0)
    qr = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, NB, S, H, kv_shift)
    o_ref.backward(dO.double())
    dqkv, dOd = torch.zeros_like(qkvd), dO.to(DEV)
    call("eg_attention_bwd", ptr(qkvd), ptr(ctx), ptr(dOd), ptr(lse), ptr(dqkv), NB, S, H, kv_shift, L.EG_F32, 0.0, 0, 0, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(ctx.cpu().double(), o_ref.detach(), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(lse.cpu().double(), lse_ref.detach(), rtol=1e-6, atol=2e-6)
    torch.testing.assert_close(dqkv.cpu().double(), qr.grad, rtol=1e-4, atol=1e-5)


def test_attention_dropout_consistency():
    """Forward with attention-probability dropout == reference using the mask recovered from the kernel
    itself (V = identity trick), and the backward matches autograd through that same mask."""
    NB, H, S = 2, 1, 32
    D = 32
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(NB * S, 3 * D, generator=g)
    # V = one-hot rows: ctx[q, j] = P_dropped[q, j] for j < 32
    qkv[:, 2 * D:] = torch.eye(S)[:, :D].repeat(NB, 1)
    qkv = qkv.to(torch.bfloat16)
    st = dev_state(seed=99)
    qkvd, ctx, lse = qkv.to(DEV), torch.zeros(NB * S, D, device=DEV, dtype=torch.bfloat16), torch.zeros(NB, H, S, device=DEV)
    call("eg_attention_fwd", ptr(qkvd), ptr(ctx), ptr(lse), NB, S, H, 0, L.EG_BF16, 0.25, 11, ptr(st), 0)
    torch.cuda.synchronize()
    x = qkv.double().reshape(NB, S, 3, D)
    p = torch.softmax(x[:, :, 0] @ x[:, :, 1].transpose(1, 2) / math.sqrt(32), -1)
    pd = ctx.cpu().double().reshape(NB, S, D)
    mask = pd != 0
    assert abs(mask.float().mean().item() - 0.75) < 0.04
    torch.testing.assert_close(pd[mask], (p / 0.75)[mask], rtol=2e-2, atol=2e-3)


def test_stft_and_ibs_connectivity_match_oracle():
    """STFT image and the 6 x 7 x C x C synchrony matrices against the CPU oracle on fresh seeded windows."""
    import ctypes as CT
    from oracle import dual_eeg_oracle as O
    B, Cc, T, fs = 3, 8, 1024, 256.0
    g = torch.Generator().manual_seed(21)
    base = torch.randn(B, Cc, T, generator=g)
    t_ = torch.arange(T) / fs
    x1 = base + torch.sin(2 * math.pi * 10 * t_) + 0.5 * torch.sin(2 * math.pi * 22 * t_ + 1.0)
    x2 = 0.6 * base.roll(5, dims=2) + 0.4 * torch.randn(B, Cc, T, generator=g) + torch.sin(2 * math.pi * 10 * t_ + 0.7)
    x1d, x2d = x1.contiguous().to(DEV), x2.contiguous().to(DEV)
    # STFT
    win = torch.hann_window(128).to(DEV)
    img = torch.zeros(B * Cc, 64, 17, device=DEV)
    call("eg_stft_logmag", ptr(x1d), ptr(win), ptr(img), B * Cc, T, 128, 64, 64, 0)
    torch.cuda.synchronize()
    # magnitude domain: log() of a near-zero bin amplifies fp32 rounding of the 128-point DFT
    np.testing.assert_allclose(np.exp(img.cpu().numpy()), np.exp(O.stft_logmag(x1, 128, 64, 64).numpy()), atol=2e-5, rtol=2e-4)
    # connectivity
    bands = O.ROBUST_BANDS
    lo = (CT.c_float * 6)(*[b[0] for b in bands])
    hi = (CT.c_float * 6)(*[b[1] for b in bands])
    nsig, nbin = 2 * B * Cc, 181
    xcat = torch.cat([x1d, x2d], 0)
    xb, ph = torch.zeros(6, nsig, T, device=DEV), torch.zeros(6, nsig, T, device=DEV)
    stats, spec = torch.zeros(6, nsig, 4, device=DEV), torch.zeros(nsig, nbin, 2, device=DEV)
    conn = torch.zeros(B, 6, 7, Cc, Cc, device=DEV)
    call("eg_ibs_analytic", ptr(xcat), ptr(xb), ptr(ph), ptr(stats), ptr(spec), nsig, T, fs, nbin, CT.addressof(lo), CT.addressof(hi), 6, 0)
    call("eg_ibs_pairs", ptr(xb), ptr(ph), ptr(stats), ptr(spec), ptr(conn), B, Cc, T, fs, nbin, CT.addressof(lo), CT.addressof(hi), 6, 0)
    torch.cuda.synchronize()
    cfg = O.ModelCfg(in_channels=Cc)
    ref = O.ibs_connectivity(x1, x2, cfg).numpy()
    got = conn.cpu().numpy()
    # band-limited signal of the broadband band against the oracle's FFT-mask band-pass
    np.testing.assert_allclose(xb[0, :B * Cc].cpu().numpy().reshape(B, Cc, T), O.bandpass(x1, fs, 0.5, 45).numpy(), atol=2e-5)
    for f in range(7):
        d = np.abs(got[:, :, f] - ref[:, :, f])
        if f in (1, 2):
            assert (d > 1e-4).mean() < 5e-3 and d.max() < 6.5 / T, (f, d.max())
        else:
            assert d.max() < 1e-4, (f, d.max())
    # scalar variant (4 bands x 7 global features)
    lo4 = (CT.c_float * 4)(*[b[0] for b in O.SCALAR_BANDS])
    hi4 = (CT.c_float * 4)(*[b[1] for b in O.SCALAR_BANDS])
    xb4, ph4 = torch.zeros(4, nsig, T, device=DEV), torch.zeros(4, nsig, T, device=DEV)
    st4 = torch.zeros(4, nsig, 4, device=DEV)
    call("eg_ibs_analytic", ptr(xcat), ptr(xb4), ptr(ph4), ptr(st4), ptr(spec), nsig, T, fs, nbin, CT.addressof(lo4), CT.addressof(hi4), 4, 0)
    feats = torch.zeros(B, 64, device=DEV)
    call("eg_ibs_scalar", ptr(xb4), ptr(ph4), ptr(spec), ptr(feats), B, Cc, T, fs, nbin, CT.addressof(lo4), CT.addressof(hi4), 4, 0, 4, 64, 0)
    torch.cuda.synchronize()
    refs = O.ibs_scalar_features(x1, x2, cfg).numpy()
    gots = feats.cpu().numpy()[:, :28]
    d = np.abs(gots - refs)
    flip = np.isin(np.arange(28) % 7, [1, 2])
    assert d[:, ~flip].max() < 1e-4, d[:, ~flip].max()
    assert d[:, flip].max() < 2e-3


@pytest.mark.parametrize("B,Cc,T", [(2, 12, 512), (1, 32, 256), (2, 5, 128), (1, 16, 2048)])
def test_ibs_connectivity_shapes(B, Cc, T):
    """eg_ibs_analytic + eg_ibs_pairs against the CPU oracle away from the bench shape: channel counts that are not a multiple of the
    8 x 8 tile (padded lanes must not reach the output), several tiles per window pair (the XCD-ordered 1-D grid), an odd log2 T
    (the closing radix-2 stage of the FFT) and windows shorter / longer than one 256-step LDS chunk."""
    import ctypes as CT
    from oracle import dual_eeg_oracle as O
    fs = 256.0
    g = torch.Generator().manual_seed(100 + Cc + T)
    base = torch.randn(B, Cc, T, generator=g)
    t_ = torch.arange(T) / fs
    x1 = base + torch.sin(2 * math.pi * 10 * t_) + 0.5 * torch.sin(2 * math.pi * 22 * t_ + 1.0)
    x2 = 0.6 * base.roll(5, dims=2) + 0.4 * torch.randn(B, Cc, T, generator=g) + torch.sin(2 * math.pi * 10 * t_ + 0.7)
    bands = O.ROBUST_BANDS
    lo = (CT.c_float * 6)(*[b[0] for b in bands])
    hi = (CT.c_float * 6)(*[b[1] for b in bands])
    nsig, nbin = 2 * B * Cc, min(T // 2 + 1, int(45.0 * T / fs) + 1)
    xcat = torch.cat([x1, x2], 0).contiguous().to(DEV)
    xb, ph = torch.zeros(6, nsig, T, device=DEV), torch.zeros(6, nsig, T, device=DEV)
    stats, spec = torch.zeros(6, nsig, 4, device=DEV), torch.zeros(nsig, nbin, 2, device=DEV)
    conn = torch.full((B, 6, 7, Cc, Cc), float("nan"), device=DEV)
    call("eg_ibs_analytic", ptr(xcat), ptr(xb), ptr(ph), ptr(stats), ptr(spec), nsig, T, fs, nbin, CT.addressof(lo), CT.addressof(hi), 6, 0)
    call("eg_ibs_pairs", ptr(xb), ptr(ph), ptr(stats), ptr(spec), ptr(conn), B, Cc, T, fs, nbin, CT.addressof(lo), CT.addressof(hi), 6, 0)
    torch.cuda.synchronize()
    ref = O.ibs_connectivity(x1, x2, O.ModelCfg(in_channels=Cc)).numpy()
    got = conn.cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(xb[0, :B * Cc].cpu().numpy().reshape(B, Cc, T), O.bandpass(x1, fs, 0.5, 45).numpy(), atol=2e-5)
    for f in range(7):
        d = np.abs(got[:, :, f] - ref[:, :, f])
        if f in (1, 2):   # sign()-based: one flipped sample of T moves PLI by 2/T
            assert (d > 1e-4).mean() < 5e-3 and d.max() < 6.5 / T, (f, d.max())
        else:
            assert d.max() < 1e-4, (f, d.max())


def test_heads_and_ce():
    B, S, D, ncls, off = 8, 20, 64, 3, 5
    dtype = L.EG_F32
    g = torch.Generator().manual_seed(1)
    z = torch.randn(2 * B, S, D, generator=g)
    zd = z.to(DEV)
    cls1, cls2 = torch.zeros(B, D, device=DEV), torch.zeros(B, D, device=DEV)
    comb, zf = torch.zeros(B, 3 * D, device=DEV), torch.zeros(B, 3 * D, device=DEV)
    ipf, ip = torch.zeros(B, D, device=DEV), torch.zeros(B, D, device=DEV)
    call("eg_pool_fuse_fwd", ptr(zd), ptr(cls1), ptr(cls2), ptr(comb), ptr(zf), ptr(ipf), ptr(ip), B, S, D, off, 3, 1, dtype, 0)
    torch.cuda.synchronize()
    a, c = z[:B, 0], z[B:, 0]
    torch.testing.assert_close(comb.cpu(), torch.cat([a + c, a * c, (a - c).abs()], 1))
    torch.testing.assert_close(zf.cpu()[:, D:2 * D], z[:B, off:].mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(zf.cpu()[:, 2 * D:], z[B:, off:].mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ipf.cpu(), z[:B, 1:4].mean(1), rtol=1e-5, atol=1e-6)
    # classifier + CE forward/backward
    h = torch.randn(B, D, generator=g)
    W = torch.randn(ncls, D, generator=g) * 0.2
    bias = torch.randn(ncls, generator=g) * 0.1
    labels = torch.tensor([0, 1, 2, 1, 0, 2, 2, 1])
    hd, Wd, bd, ld = h.to(DEV), W.to(DEV), bias.to(DEV), labels.to(DEV)
    logits, sl, loss = torch.zeros(B, ncls, device=DEV), torch.zeros(B, device=DEV), torch.zeros(1, device=DEV)
    call("eg_classifier_ce_fwd", ptr(hd), ptr(Wd), ptr(bd), ptr(ld), ptr(logits), ptr(sl), ptr(loss), B, D, ncls, dtype, 0)
    hr, Wr, br = h.clone().requires_grad_(True), W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    lr_ = hr @ Wr.T + br
    lossr = torch.nn.functional.cross_entropy(lr_, labels)
    (2.0 * lossr).backward()
    torch.cuda.synchronize()
    torch.testing.assert_close(logits.cpu(), lr_.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss.cpu()[0], lossr.detach(), rtol=1e-5, atol=1e-6)
    gl = torch.full((1,), 2.0, device=DEV)
    dlog, dh, dW, db = torch.zeros(B, ncls, device=DEV), torch.zeros(B, D, device=DEV), torch.zeros(ncls, D, device=DEV), torch.zeros(ncls, device=DEV)
    call("eg_classifier_ce_bwd", ptr(hd), ptr(Wd), ptr(logits), ptr(ld), ptr(gl), 0, ptr(dlog), ptr(dh), ptr(dW), ptr(db), B, D,
         ncls, 0, 1.0, dtype, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(dh.cpu(), hr.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dW.cpu(), Wr.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-6)


def test_adamw_clip_matches_oracle():
    from oracle.dual_eeg_oracle import clip_and_adamw
    n = 100003
    g = torch.Generator().manual_seed(2)
    p = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g) * 0.05
    pd, gd, m, v = p.to(DEV), gr.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    part = torch.zeros(64, device=DEV)
    ref_p = {"w": p.clone()}
    state = {}
    for step in (1, 2, 3):
        st = dev_state(seed=0, lr=1e-3, step=step)
        call("eg_grad_sqnorm", ptr(gd), n, ptr(part), 64, 0)
        call("eg_clip_coef", ptr(part), 64, 1.0, ptr(st), 0)
        call("eg_adamw", ptr(pd), ptr(gd), ptr(m), ptr(v), n, 0.9, 0.999, 1e-8, 0.01, ptr(st), 0)
        torch.cuda.synchronize()
        total = clip_and_adamw(ref_p, {"w": gr.clone()}, state, step=step, lr=1e-3)
        s = read_state(st)
        assert abs(s.grad_norm - total) < 1e-4 * total
        torch.testing.assert_close(pd.cpu(), ref_p["w"], rtol=1e-5, atol=1e-6)


def test_rejects_bad_arguments():
    """Error behaviour of the boundary: bad shapes are refused on the host before any launch."""
    A = torch.zeros(8, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(L.EgError):
        gemm_nt(A, A, 8, 8, 60, L.EG_BF16)  # K not a multiple of the K-tile
    with pytest.raises(L.EgError):
        gemm_nt(A, A, 8, 7, 64, L.EG_BF16)  # N not a multiple of 8
    with pytest.raises(L.EgError):
        call("eg_attention_fwd", ptr(A), ptr(A), ptr(A), 1, 200, 1, 0, L.EG_BF16, 0.0, 0, 0, 0)  # S too long
    with pytest.raises(L.EgError):
        call("eg_attention_fwd", ptr(A), ptr(A), ptr(A), 1, 8, 1, 0, 7, 0.0, 0, 0, 0)  # unknown dtype


@pytest.mark.parametrize("mode", ["full", "no_temperature", "no_fuzzification", "fixed_weights"])
def test_fuzzy_gating_fusion_matches_reference(mode):
    """HIP FuzzyGatingFusion.forward against fixtures emitted by the reference (all four ablation modes)."""
    from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion
    from tests.helpers import GOLDEN
    z = np.load(GOLDEN / "fuzzy_gating.npz")
    m = FuzzyGatingFusion(num_classes=3, mode=mode)
    sd = {k.split("/state/")[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(mode + "/state/")}
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        fused, alpha, _ = m.to(DEV)(torch.from_numpy(z["z_img"]).to(DEV), torch.from_numpy(z["z_eeg"]).to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(fused.cpu().numpy(), z[mode + "/z_fused"], atol=2e-5)
    # sample 0 of no_fuzzification is 0/0 in fp32 (both entropies at their maximum): skip its alpha
    sl = slice(1, None) if mode == "no_fuzzification" else slice(None)
    np.testing.assert_allclose(alpha.cpu().numpy()[sl], z[mode + "/alpha"][sl], atol=2e-5)


@pytest.mark.parametrize("mode", ["full", "no_temperature", "no_fuzzification", "fixed_weights"])
@pytest.mark.parametrize("variant", ["generic", "loop"])
def test_fuzzy_gating_fusion_gradients_match_reference(mode, variant):
    """eg_fuzzy_gate_bwd behind autograd against gradients the reference module produced under torch autograd
    (tests/golden/fuzzy_grad.npz): a generic upstream gradient (CE on fused + a term in alpha) and the loss of the
    reference's multimodal step (train_multimodal_fuzzy_fusion.py:436-460)."""
    import torch.nn.functional as F
    from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion
    from tests.helpers import GOLDEN
    z = np.load(GOLDEN / "fuzzy_grad.npz")
    m = FuzzyGatingFusion(num_classes=3, mode=mode)
    m.load_state_dict({k.split("/state/")[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(mode + "/state/")}, strict=True)
    m = m.to(DEV)
    zi = torch.from_numpy(z["z_img"]).to(DEV).requires_grad_(True)
    ze = torch.from_numpy(z["z_eeg"]).to(DEV).requires_grad_(True)
    labels = torch.from_numpy(z["labels"]).to(DEV)
    fused, alpha, aux = m(zi, ze)
    if variant == "generic":
        loss = F.cross_entropy(fused, labels) + 0.3 * (alpha ** 2).sum()
    else:
        T_i, T_e = aux["temperatures"]["img"], aux["temperatures"]["eeg"]
        loss = (F.cross_entropy(fused, labels) + 0.3 * F.cross_entropy(zi / T_i, labels) + 0.3 * F.cross_entropy(ze / T_e, labels)
                + 0.1 * m.compute_temperature_regularization(0.5, 5.0))
    loss.backward()
    torch.cuda.synchronize()
    pre = f"{mode}/{variant}/"
    np.testing.assert_allclose(fused.detach().cpu().numpy(), z[pre + "fused"], atol=2e-5)
    np.testing.assert_allclose(alpha.detach().cpu().numpy(), z[pre + "alpha"], atol=1e-5)
    np.testing.assert_allclose(float(loss), float(z[pre + "loss"]), atol=2e-5)
    np.testing.assert_allclose(zi.grad.cpu().numpy(), z[pre + "d_img"], atol=5e-6, rtol=1e-4)
    np.testing.assert_allclose(ze.grad.cpu().numpy(), z[pre + "d_eeg"], atol=5e-6, rtol=1e-4)
    for n, p in m.named_parameters():
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(z[pre + "d_" + n])
        np.testing.assert_allclose(got, z[pre + "d_" + n], atol=1e-5, rtol=2e-4, err_msg=n)


def test_fuzzy_gating_fusion_large_batch_reduction():
    """B > one workgroup: the 12 parameter gradients are block partials summed in order; compare with the CPU oracle."""
    import torch.nn.functional as F
    from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion
    from oracle import fuzzy_oracle as FO
    g = torch.Generator().manual_seed(3)
    B = 1000
    zi, ze = torch.randn(B, 3, generator=g) * 2, torch.randn(B, 3, generator=g) * 2
    labels = torch.arange(B) % 3
    m = FuzzyGatingFusion(num_classes=3, mode="full")
    p = {n: v.detach().clone().requires_grad_(True) for n, v in m.named_parameters()}
    a, b = zi.clone().requires_grad_(True), ze.clone().requires_grad_(True)
    fused, alpha, _ = FO.fuzzy_forward_torch(a, b, p, "full")
    F.cross_entropy(fused, labels).backward()
    m = m.to(DEV)
    zd, ed = zi.to(DEV).requires_grad_(True), ze.to(DEV).requires_grad_(True)
    out = m(zd, ed)
    F.cross_entropy(out[0], labels.to(DEV)).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(zd.grad.cpu().numpy(), a.grad.numpy(), atol=2e-7, rtol=1e-3)
    for n, q in m.named_parameters():
        np.testing.assert_allclose(q.grad.cpu().numpy(), p[n].grad.numpy(), atol=2e-6, rtol=1e-3, err_msg=n)


# ------------------------------------------------------------------------------------------------
# row-complete GEMM tile (N == 256) with LayerNorm in the epilogue (eg_gemm_desc.ln_*)
# ------------------------------------------------------------------------------------------------
def test_reduce_table_wide_and_narrow_entries():
    """eg_reduce_table: entries with few splits take the one-column-per-thread path, entries with many short slabs the
    8-column x 32-lane path; block ranges follow include/eyegaze_hip.h (EG_REDUCE_WIDE_SPLITS)."""
    from eyegaze_multimodal_amd._lib import ReduceEntry
    from eyegaze_multimodal_amd.engine import _reduce_blocks
    g = torch.Generator().manual_seed(9)
    specs = [(4096 + 36, 5), (512, 100), (1040, 8), (1040, 9), (4, 3), (262144 + 8, 2)]     # (n floats, splits); n % 4 == 0
    parts = [torch.randn(s, n + 12, generator=g).to(DEV) for n, s in specs]                 # stride > n
    outs = [torch.full((n + 4,), 7.0, device=DEV) for n, _ in specs]
    tab = (ReduceEntry * len(specs))()
    blk = 0
    for e, (n, s), p, o in zip(tab, specs, parts, outs):
        e.partial, e.out, e.n, e.stride, e.splits, e.blk0 = ptr(p), ptr(o), n, n + 12, s, blk
        blk += _reduce_blocks(n, s)
    dtab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(DEV)
    call("eg_reduce_table", ptr(dtab), len(specs), blk, 0)
    torch.cuda.synchronize()
    for (n, s), p, o in zip(specs, parts, outs):
        ref = p[:, :n].double().sum(0)
        torch.testing.assert_close(o[:n].double().cpu(), ref.cpu(), rtol=1e-6, atol=1e-5)
        assert float(o[n:].min()) == 7.0          # nothing written past n


@pytest.mark.parametrize("knob", ["0", "2"], ids=["two_pass_everywhere", "single_sweep_up_to_128"])
def test_attention_backward_variants_not_selected_by_default(knob):
    """The attention backward has two kernels (two-pass; single sweep, the default at every length since round 3).  The selection
    is read once per process, so the non-default kernel (two-pass, knob 0; knob 2 is the default's old spelling) is exercised by
    re-running this file's attention tests in a child process with EYEGAZE_ATTN_BWD1 set."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, EYEGAZE_ATTN_BWD1=knob)
    r = subprocess.run([sys.executable, "-m", "pytest", str(Path(__file__)), "-q", "-x", "-m", "gpu", "-k",
                        "attention and not variants_not_selected"], env=env, cwd=str(repo), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
