"""256 x 256 grouped weight-gradient tile (eg_gemm_tn_grouped256) against the 128 x 128 grouped kernel and fp64 torch.
Both accumulate every output element over the reduction rows in the same order (64-row stages, 32-deep MFMAs), so for equal
row splits the partial slabs -- weights and fused bias sums -- must be BIT-IDENTICAL.  Shapes: the encoder's four products
(q|k|v with three parameter parts, out-proj, linear1, linear2), the benchmark row count and a ragged one."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import call, ptr  # noqa: E402

DEV = "cuda"
PROBS = [(768, 256, 3), (256, 256, 1), (256, 1024, 1), (1024, 256, 1)]      # N, K, parameter parts


def run(entry, tile, M, splits, dtype, ops):
    tdt = torch.bfloat16 if dtype == L.EG_BF16 else torch.float16
    total = sum(N * K + N for N, K, _ in PROBS)
    partial = torch.full((splits * total,), 7.0, device=DEV)
    tp = (L.TNProblem * len(PROBS))()
    blk, off = 0, 0
    for e, (N, K, parts), (dY, X) in zip(tp, PROBS, ops):
        e.dY, e.X, e.partial = ptr(dY), ptr(X), ptr(partial) + 4 * off
        e.ldy, e.ldx, e.N, e.K, e.part_rows, e.has_bias, e.blk0 = N, K, N, K, N // parts, 1, blk
        blk += (N // tile) * (K // tile) * splits
        off += splits * (N * K + N)
    tab = torch.frombuffer(bytearray(bytes(tp)), dtype=torch.uint8).to(DEV)
    call(entry, ptr(tab), len(PROBS), blk, M, splits, dtype, 0)
    torch.cuda.synchronize()
    return partial


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("M,splits", [(33280, 3), (1000, 3), (4160, 5), (70, 2)])
def test_tn256_partials_are_bit_identical_to_the_128_tile(M, splits, dtype):
    tdt = torch.bfloat16 if dtype == L.EG_BF16 else torch.float16
    g = torch.Generator(device="cpu").manual_seed(M)
    ops = [((torch.randn(M, N, generator=g) * 0.3).to(tdt).to(DEV), (torch.randn(M, K, generator=g) * 0.5).to(tdt).to(DEV))
           for N, K, _ in PROBS]
    a = run("eg_gemm_tn_grouped", 128, M, splits, dtype, ops)
    b = run("eg_gemm_tn_grouped256", 256, M, splits, dtype, ops)
    assert torch.equal(a, b), float((a - b).abs().max())
    # and the summed slabs are the weight / bias gradients (fp64 reference on the rounded operands)
    off = 0
    for (N, K, parts), (dY, X) in zip(PROBS, ops):
        slab = N * K + N
        tot = b[off:off + splits * slab].view(splits, slab).double().sum(0).cpu()
        off += splits * slab
        dW = dY.double().cpu().T @ X.double().cpu()
        db = dY.double().cpu().sum(0)
        P = N // parts
        for i in range(parts):      # part i: [P x K weights | P bias sums]
            w = tot[i * (P * K + P): i * (P * K + P) + P * K].view(P, K)
            bb = tot[i * (P * K + P) + P * K: (i + 1) * (P * K + P)]
            assert float((w - dW[i * P:(i + 1) * P]).abs().max()) <= 1e-3 * max(1.0, float(dW.abs().max()))
            assert float((bb - db[i * P:(i + 1) * P]).abs().max()) <= 1e-3 * max(1.0, float(db.abs().max()))


@pytest.mark.parametrize("M,splits", [(32768, 10), (1000, 4)])
def test_single_problem_tn256_with_overlapping_conv_rows(M, splits):
    """eg_gemm_tn(tile = 256) on the strided-conv operand layout (X rows overlap: row m starts 1024 elements after row m - 1 inside
    a window, 6400 elements long) gives the same partial slabs as tile = 128."""
    from eyegaze_multimodal_amd._lib import GemmTNDesc, rowmap
    N, K, T2, stride, R0 = 256, 6400, 64, 1024, 64 * 1024 + 6400
    groups = (M + T2 - 1) // T2
    g = torch.Generator(device="cpu").manual_seed(7)
    X = (torch.randn(groups * R0, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    dY = (torch.randn(M, N, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    outs = []
    for tile in (128, 256):
        part = torch.full((splits * N * K,), 7.0, device=DEV)
        d = GemmTNDesc()
        d.dY, d.X, d.partial = ptr(dY), ptr(X), ptr(part)
        d.y, d.x = rowmap(N), rowmap(stride, R0, T2)
        d.M, d.N, d.K, d.splits, d.dtype, d.tile = M, N, K, splits, L.EG_BF16, tile
        call("eg_gemm_tn", C.byref(d), 0)
        torch.cuda.synchronize()
        outs.append(part)
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    rows = torch.stack([X[(m // T2) * R0 + (m % T2) * stride:(m // T2) * R0 + (m % T2) * stride + K] for m in range(min(M, 1000))])
    if M <= 1000:
        ref = dY.double().cpu().T @ rows.double().cpu()
        got = outs[1].view(splits, N, K).double().sum(0).cpu()
        assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
