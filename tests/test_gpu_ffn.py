"""Chained feed-forward launch (csrc/ffn.hip: eg_ffn_chain) against the two eg_gemm_nt launches it replaces, and against
fp64 torch.  Both paths run the same k-ordered MFMA chains, epilogue order and dropout indices, so the stored hidden rows and
the block output must be BIT-IDENTICAL.  Shapes: the benchmark size (33 280 rows), ragged M (not a multiple of 160 or 16),
fewer rows than one workgroup, one and several hidden chunks, forward form (bias, ReLU, three dropout sites, residual = the
input rows) and backward-data form (gate from the saved hidden rows, separate residual), bf16 and fp16."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import FfnDesc, GemmDesc, call, ptr, rowmap  # noqa: E402
from tests.test_gpu_ops import dev_state  # noqa: E402

DEV = "cuda"
D = 256
TDT = {L.EG_BF16: torch.bfloat16, L.EG_F16: torch.float16}


def operands(M, F, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed + M + F)
    t = TDT[dtype]
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(t).to(DEV)
    return dict(A=mk(M, D, sc=0.5), W1=mk(F, D, sc=0.1), W2=mk(D, F, sc=0.05), b1=torch.randn(F, generator=g).to(DEV) * 0.1,
                b2=torch.randn(D, generator=g).to(DEV) * 0.1, R=mk(M, D), G=mk(M, F), st=dev_state(seed=77 + seed))


def frag_pack(w, mode, dtype):
    """The weight in eg_ffn_chain's fragment order, written by eg_pack_table from its fp32 image (modes 3-6)."""
    src = w.float().contiguous()
    dst = torch.zeros(src.numel(), device=DEV, dtype=TDT[dtype])
    e = (L.PackEntry * 1)()
    e[0].src, e[0].dst, e[0].rows, e[0].cols, e[0].ldd, e[0].mode, e[0].blk0, e[0].nblk = (
        ptr(src), ptr(dst), src.shape[0], src.shape[1], 0, mode, 0, src.numel() // 2048)
    tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).to(DEV)
    call("eg_pack_table", ptr(tab), 1, src.numel() // 2048, dtype, 0)
    torch.cuda.synchronize()
    return dst


def two_launches(o, M, F, dtype, mode, p):
    t = TDT[dtype]
    H = torch.full((M, F), 7.0, device=DEV, dtype=t)
    Cc = torch.full((M, D), 7.0, device=DEV, dtype=t)
    d = GemmDesc()
    d.A, d.W, d.C, d.state = ptr(o["A"]), ptr(o["W1"]), ptr(H), ptr(o["st"])
    d.a, d.c = rowmap(D), rowmap(F)
    d.r, d.p = d.c, d.c
    d.M, d.N, d.K, d.ldw, d.dtype = M, F, D, D, dtype
    if mode == "fwd":
        d.bias, d.act, d.drop1_p, d.drop1_site = ptr(o["b1"]), L.ACT_RELU, p, 21
    else:
        d.gate, d.gate_scale = ptr(o["G"]), 1.25
    call("eg_gemm_nt", C.byref(d), 0)
    e = GemmDesc()
    e.A, e.W, e.C, e.state = ptr(H), ptr(o["W2"]), ptr(Cc), ptr(o["st"])
    e.a, e.c = rowmap(F), rowmap(D)
    e.r, e.p = e.c, e.c
    e.M, e.N, e.K, e.ldw, e.dtype = M, D, F, F, dtype
    if mode == "fwd":
        e.bias, e.drop1_p, e.drop1_site, e.drop2_p, e.drop2_site = ptr(o["b2"]), p, 22, p, 23
        e.residual = ptr(o["A"])
    else:
        e.residual = ptr(o["R"])
    call("eg_gemm_nt", C.byref(e), 0)
    torch.cuda.synchronize()
    return H, Cc


def gate_bits(M, F):
    L.lib().eg_ffn_gate_bits_bytes.restype = C.c_int64
    return torch.zeros(L.lib().eg_ffn_gate_bits_bytes(M, F) // 8, device=DEV, dtype=torch.int64)


def one_launch(o, M, F, dtype, mode, p, bits_out=None, bits_in=None):
    t = TDT[dtype]
    H = torch.full((M, F), 7.0, device=DEV, dtype=t)
    Cc = torch.full((M, D), 7.0, device=DEV, dtype=t)
    d = FfnDesc()
    # role 1 from the [F, 256] matrix itself (mode 3) or from its transpose as the parameter stores it (mode 4); same for role 2
    w1f = frag_pack(o["W1"], 3, dtype) if mode == "fwd" else frag_pack(o["W1"].t().contiguous(), 4, dtype)
    w2f = frag_pack(o["W2"], 5, dtype) if mode == "fwd" else frag_pack(o["W2"].t().contiguous(), 6, dtype)
    d.A, d.W1, d.W2, d.H, d.C, d.state = ptr(o["A"]), ptr(w1f), ptr(w2f), ptr(H), ptr(Cc), ptr(o["st"])
    d.lda, d.ldh, d.ldc, d.ldg, d.ldr = D, F, D, F, D
    d.M, d.F, d.dtype = M, F, dtype
    if mode == "fwd":
        d.bias1, d.bias2, d.act1, d.residual = ptr(o["b1"]), ptr(o["b2"]), L.ACT_RELU, ptr(o["A"])
        d.drop_h_p, d.drop_h_site, d.drop_c1_p, d.drop_c1_site, d.drop_c2_p, d.drop_c2_site = p, 21, p, 22, p, 23
        d.gate_bits_out = ptr(bits_out) if bits_out is not None else None
    else:
        d.gate, d.gate_scale, d.residual = ptr(o["G"]), 1.25, ptr(o["R"])
        if bits_in is not None:
            d.gate, d.gate_bits_in = None, ptr(bits_in)
    call("eg_ffn_chain", C.byref(d), 0)
    torch.cuda.synchronize()
    return H, Cc


CASES = [
    # M, F, mode, dropout p
    (33280, 1024, "fwd", 0.1),
    (33280, 1024, "bwd", 0.0),
    (33280, 1024, "fwd", 0.0),
    (1037, 1024, "fwd", 0.2),       # ragged: 6 full workgroups + 77 rows
    (1037, 1024, "bwd", 0.0),
    (7, 128, "fwd", 0.1),           # less than one MFMA tile, one hidden chunk
    (161, 256, "bwd", 0.0),         # one row into the second workgroup
    (4160, 512, "fwd", 0.1),        # B = 32 at S = 65
]


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "M%d_F%d_%s_p%g" % c)
def test_ffn_chain_is_bit_identical_to_two_gemm_launches(case, dtype):
    M, F, mode, p = case
    o = operands(M, F, dtype, seed=3)
    H2, C2 = two_launches(o, M, F, dtype, mode, p)
    H1, C1 = one_launch(o, M, F, dtype, mode, p)
    assert torch.equal(H1, H2), float((H1.float() - H2.float()).abs().max())
    assert torch.equal(C1, C2), float((C1.float() - C2.float()).abs().max())


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", [c for c in CASES if c[2] == "fwd"], ids=lambda c: "M%d_F%d_%s_p%g" % c)
def test_gate_bits_written_by_the_forward_launch_equal_the_row_gate(case, dtype):
    """The forward launch leaves "stored hidden value > 0" as one bit per element; the backward launch that reads those bits
    must equal both the two-launch path and the one-launch path gated by the hidden rows themselves."""
    M, F, _, p = case
    o = operands(M, F, dtype, seed=11)
    if dtype == L.EG_F16:       # values whose fp32 result is positive but rounds to a zero half: the bit must follow the STORED value
        o["A"][: min(M, 3)] *= 1e-4
    bits = gate_bits(M, F)
    Hf, Cf = one_launch(o, M, F, dtype, "fwd", p, bits_out=bits)
    Hp, Cp = one_launch(o, M, F, dtype, "fwd", p)
    assert torch.equal(Hf, Hp) and torch.equal(Cf, Cp)          # writing the bits changes nothing else
    ob = dict(o, A=o["R"], G=Hf, R=o["A"])
    H2, C2 = two_launches(ob, M, F, dtype, "bwd", 0.0)
    Hr, Cr = one_launch(ob, M, F, dtype, "bwd", 0.0)
    Hb, Cb = one_launch(ob, M, F, dtype, "bwd", 0.0, bits_in=bits)
    assert torch.equal(Hr, H2) and torch.equal(Cr, C2)
    assert torch.equal(Hb, H2), float((Hb.float() - H2.float()).abs().max())
    assert torch.equal(Cb, C2), float((Cb.float() - C2.float()).abs().max())
    frac = float((Hf > 0).float().mean())
    assert 0.2 < frac < 0.6, frac                               # the gate is neither all-pass nor all-zero


@pytest.mark.parametrize("case", [c for c in CASES if c[3] == 0.0], ids=lambda c: "M%d_F%d_%s_p%g" % c)
def test_ffn_chain_matches_fp64(case):
    M, F, mode, p = case
    o = operands(M, F, L.EG_BF16, seed=5)
    H1, C1 = one_launch(o, M, F, L.EG_BF16, mode, p)
    A, W1, W2 = o["A"].double().cpu(), o["W1"].double().cpu(), o["W2"].double().cpu()
    if mode == "fwd":
        h = (A @ W1.T + o["b1"].double().cpu()).clamp_min(0)
    else:
        h = torch.where(o["G"].double().cpu() > 0, (A @ W1.T) * 1.25, torch.zeros(M, F, dtype=torch.float64))
    hq = h.to(torch.bfloat16).double()      # the stored hidden rows are what product 2 consumes
    c = hq @ W2.T + (o["b2"].double().cpu() + A if mode == "fwd" else o["R"].double().cpu())
    assert float((H1.double().cpu() - h).abs().max()) <= 0.02 + 0.008 * float(h.abs().max())
    assert float((C1.double().cpu() - c).abs().max()) <= 0.03 + 0.008 * float(c.abs().max())


def test_ffn_chain_argument_checks():
    d = FfnDesc()
    with pytest.raises(L.EgError):
        call("eg_ffn_chain", C.byref(d), 0)
    o = operands(16, 128, L.EG_BF16, seed=1)
    H = torch.empty(16, 128, device=DEV, dtype=torch.bfloat16)
    Cc = torch.empty(16, D, device=DEV, dtype=torch.bfloat16)
    d.A, d.W1, d.W2, d.H, d.C = ptr(o["A"]), ptr(frag_pack(o["W1"], 3, L.EG_BF16)), ptr(frag_pack(o["W2"], 5, L.EG_BF16)), ptr(H), ptr(Cc)
    d.lda, d.ldh, d.ldc, d.M, d.F, d.dtype = D, 128, D, 16, 100, L.EG_BF16        # F not a multiple of 128
    with pytest.raises(L.EgError):
        call("eg_ffn_chain", C.byref(d), 0)
    d.F, d.dtype = 128, L.EG_F32                                                    # fp32 keeps the two-launch path
    with pytest.raises(L.EgError):
        call("eg_ffn_chain", C.byref(d), 0)
