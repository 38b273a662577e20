"""Tall NT tile (csrc/tallgemm.hip: one wave per SIMD, weights in fragment order straight to registers, 7-stage activation ring)
against the wide tile: the same k-ordered MFMA chains and epilogue, so eg_gemm_nt must give BIT-IDENTICAL results whether or not
the descriptor carries the fragment-ordered weights that select the tall tile.  Shapes: conv-1 forward at the benchmark size
(overlapping rows, K = 6400, bias + ReLU + dropout + positional residual + second output), a backward-data phase (K = 1792,
gate), ragged M, the 8 / 9 / 10-row-tile variants, bf16 and fp16."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import GemmDesc, call, ptr, rowmap  # noqa: E402
from tests.test_gpu_ops import dev_state  # noqa: E402

DEV = "cuda"
N = 256
TDT = {L.EG_BF16: torch.bfloat16, L.EG_F16: torch.float16}
CASES = [
    # M, K, overlapping conv rows, act, drop, residual, gate, out_pre
    (32768, 6400, True, 1, 0.1, 1, 0, 1),      # conv-1 forward: 256 workgroups of 8 row tiles
    (35840, 1792, False, 0, 0.0, 0, 1, 0),     # conv-1 backward-data phase: 9 row tiles
    (40960, 1536, False, 0, 0.0, 1, 0, 0),     # 10 row tiles
    (2500, 1536, False, 1, 0.2, 1, 0, 0),      # ragged M, fewer workgroups than CUs
    (100000, 2048, False, 0, 0.0, 0, 0, 0),    # more than one round of 10-tile workgroups
]


def run(case, dtype, tall):
    M, K, conv, act, drop, residual, gate, out_pre = case
    t = TDT[dtype]
    g = torch.Generator(device="cpu").manual_seed(M + K)
    if conv:       # row m of window w starts at w * R0 + (m % T2) * 1024 and is K long (rows overlap)
        T2, stride = 64, 1024
        R0 = (T2 - 1) * stride + K
        A = (torch.randn((M // T2) * R0, generator=g) * 0.5).to(t).to(DEV)
        amap = rowmap(stride, R0, T2)
    else:
        A = (torch.randn(M, K, generator=g) * 0.5).to(t).to(DEV)
        amap = rowmap(K)
    W = (torch.randn(N, K, generator=g) * 0.05).to(t).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(t).to(DEV)
    G = torch.randn(M, N, generator=g).to(t).to(DEV)
    out = torch.full((M, N), 7.0, device=DEV, dtype=t)
    pre = torch.full((M, N), 7.0, device=DEV, dtype=t)
    Wf = torch.zeros(N * K, device=DEV, dtype=t)
    call("eg_frag_order_rows", ptr(W), ptr(Wf), K, K, 1, 0)
    d = GemmDesc()
    d.A, d.W, d.C, d.bias, d.state = ptr(A), ptr(W), ptr(out), ptr(b), ptr(dev_state(seed=99))
    d.residual = ptr(R) if residual else None
    d.gate = ptr(G) if gate else None
    d.out_pre = ptr(pre) if out_pre else None
    d.a, d.c = amap, rowmap(N)
    d.r, d.p = d.c, d.c
    d.M, d.N, d.K, d.ldw, d.act, d.dtype = M, N, K, K, act, dtype
    d.drop1_p, d.drop1_site, d.gate_scale = drop, 13, 1.25 if gate else 1.0
    d.W_frag = ptr(Wf) if tall else None
    route = L.lib().eg_gemm_nt_route(C.byref(d))
    call("eg_gemm_nt", C.byref(d), 0)
    torch.cuda.synchronize()
    return route, out, pre, (A, W, b, R, amap)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "M%d_K%d_conv%d_a%d_p%g_r%d_g%d_o%d" % c)
def test_tall_tile_is_bit_identical_to_the_wide_tile(case, dtype):
    r1, o1, p1, _ = run(case, dtype, tall=True)
    r0, o0, p0, _ = run(case, dtype, tall=False)
    assert r1 == 5 and r0 == 1, (r1, r0)           # EG_ROUTE_TALL / EG_ROUTE_WIDE
    assert torch.equal(o1, o0), float((o1.float() - o0.float()).abs().max())
    if case[7]:
        assert torch.equal(p1, p0)


def test_tall_tile_matches_fp64():
    case = (2500, 1536, False, 0, 0.0, 1, 0, 0)
    _, out, _, (A, W, b, R, _) = run(case, L.EG_BF16, tall=True)
    ref = A.double().cpu() @ W.double().cpu().T + b.double().cpu() + R.double().cpu()
    assert float((out.double().cpu() - ref).abs().max()) <= 0.03 + 0.008 * float(ref.abs().max())
