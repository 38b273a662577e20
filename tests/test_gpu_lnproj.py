"""eg_ln_bwd_proj (csrc/lnproj.hip): LayerNorm backward and the backward-data product that consumes it as ONE launch over 80-row tiles.
  * dx and dx_drop bit-identical to eg_layernorm_bwd, dC bit-identical to eg_gemm_nt on that dx_drop, with dropout on and off,
    ragged M, bf16 and fp16;
  * the gain / bias gradient partials reduce to eg_layernorm_bwd's sums (another grouping of the rows: to fp32 rounding);
  * the capacity of the partial buffer is checked on the host."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import call, ptr  # noqa: E402
from tests.test_gpu_ffn import frag_pack  # noqa: E402
from tests.test_gpu_ops import DEV, DT, dev_state, gemm_nt  # noqa: E402

D = 256


def setup(M, dtype, seed):
    g = torch.Generator().manual_seed(seed + M)
    t = DT[dtype]
    x = torch.randn(M, D, generator=g).to(t).to(DEV)
    dy = (torch.randn(M, D, generator=g) * 0.3).to(t).to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    W = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV)            # out_proj.weight [out, in] (fp32 master)
    y = torch.zeros_like(x)
    stats = torch.zeros(M, 2, device=DEV)
    call("eg_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), M, D, dtype, 0)
    return x, dy, gamma, stats, W, dev_state(seed=7 + seed)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", [(33280, 0.1, 0.0), (4160, 0.1, 0.1), (1000, 0.0, 0.0), (81, 0.1, 0.0), (80, 0.0, 0.0), (7, 0.1, 0.0)])
def test_fused_launch_equals_the_two_launches(case, dtype):
    M, p1, p2 = case
    t = DT[dtype]
    x, dy, gamma, stats, W, st = setup(M, dtype, seed=3)
    # reference: eg_layernorm_bwd, then dC = dx_drop * W  (W^T as the NT product's [N, K] operand)
    nblk = 64
    dx_r, dxd_r = torch.zeros_like(x), torch.zeros_like(x)
    part_r = torch.zeros(nblk, 2 * D, device=DEV)
    call("eg_layernorm_bwd", ptr(dy), ptr(x), ptr(stats), ptr(gamma), ptr(dx_r), ptr(dxd_r), ptr(part_r), nblk, nblk, M, D, dtype,
         p1, 5, p2, 6, ptr(st), 0)
    WT = W.t().contiguous().to(t)                                       # [N = in, K = out]
    dC_r = gemm_nt(dxd_r, WT, M, D, D, dtype)
    # fused
    wf = frag_pack(W, 6, dtype)                                         # role-2 fragment order of W^T from the [out, in] parameter
    nb = L.lib().eg_ln_bwd_proj_blocks(M)
    dx, dxd, dC = (torch.full((M, D), 7.0, device=DEV, dtype=t) for _ in range(3))
    part = torch.full((nb, 2 * D), 7.0, device=DEV)
    d = L.LnBwdProjDesc()
    d.dy, d.x, d.stats, d.gamma, d.W_frag = ptr(dy), ptr(x), ptr(stats), ptr(gamma), ptr(wf)
    d.dx, d.dx_drop, d.dC, d.partial, d.state = ptr(dx), ptr(dxd), ptr(dC), ptr(part), ptr(st)
    d.M, d.d_model, d.dtype, d.partial_capacity_blocks = M, D, dtype, nb
    d.drop1_p, d.drop1_site, d.drop2_p, d.drop2_site = p1, 5, p2, 6
    call("eg_ln_bwd_proj", C.byref(d), 0)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_r), float((dx.float() - dx_r.float()).abs().max())
    assert torch.equal(dxd, dxd_r)
    assert torch.equal(dC, dC_r), float((dC.float() - dC_r.float()).abs().max())
    torch.testing.assert_close(part.sum(0), part_r.sum(0), rtol=2e-5, atol=2e-4 * math.sqrt(M))
    if p1 > 0:
        assert 0.05 < float((dxd == 0).float().mean()) < 0.3


def test_partial_capacity_is_checked_on_the_host():
    M = 1000
    x, dy, gamma, stats, W, st = setup(M, L.EG_BF16, seed=1)
    wf = frag_pack(W, 6, L.EG_BF16)
    buf = torch.zeros(M, D, device=DEV, dtype=torch.bfloat16)
    part = torch.zeros(4, 2 * D, device=DEV)
    d = L.LnBwdProjDesc()
    d.dy, d.x, d.stats, d.gamma, d.W_frag = ptr(dy), ptr(x), ptr(stats), ptr(gamma), ptr(wf)
    d.dx, d.dx_drop, d.dC, d.partial = ptr(buf), ptr(buf), ptr(buf), ptr(part)
    d.M, d.d_model, d.dtype, d.partial_capacity_blocks = M, D, L.EG_BF16, 4
    with pytest.raises(L.EgError, match="partial buffer holds 4 rows"):
        call("eg_ln_bwd_proj", C.byref(d), 0)
