"""Direct checks of the image / spectrogram branch's small kernels at the shapes of the full-size workloads (the model fixtures reach
them only at B = 4, where e.g. the two-stage reduction of eg_unpack_conv2d_wgrad never runs):
  * eg_unpack_conv2d_wgrad: dW[n][c][ky][kx] = sum over splits of partial[s][n][(ky * 4 + kx) * Cin + c], short and long reductions;
  * eg_spec_avgpool_fwd / _bwd (D:118-123): 4 x 4 adaptive average pooling of the post-ReLU conv-2 image and its gradient;
  * eg_colsum at N = 64 (the convolutions' bias gradients) and at the encoder's widths;
  * eg_conv2d_wgrad_flat: the conv-2 weight gradient as a flat correlation, against autograd's grad_weight on the same 16-bit operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import _lib as L  # noqa: E402
from eyegaze_multimodal_amd._lib import call, ptr, rowmap  # noqa: E402
from tests.test_gpu_ops import DEV, DT  # noqa: E402


@pytest.mark.parametrize("splits", [1, 3, 64, 65, 200, 768])
def test_unpack_conv2d_wgrad(splits):
    N, Cin = 64, 32
    K = 12 * Cin
    g = torch.Generator().manual_seed(splits)
    partial = torch.randn(splits, N, K, generator=g).to(DEV)
    ref = partial.double().sum(0).view(N, 3, 4, Cin)[:, :, :3, :].permute(0, 3, 1, 2).contiguous()      # [n][c][ky][kx]
    dW = torch.full((N, Cin, 3, 3), 7.0, device=DEV)
    call("eg_unpack_conv2d_wgrad", ptr(partial), ptr(dW), splits, N, Cin, 0)       # (partial is scratch: reduced in place when long)
    torch.cuda.synchronize()
    torch.testing.assert_close(dW.double(), ref, rtol=2e-5, atol=2e-5 * max(1.0, splits ** 0.5))


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16, L.EG_F32])
@pytest.mark.parametrize("shape", [(5, 16, 16), (3, 32, 8), (2, 8, 24)])
def test_spec_avgpool_forward_and_backward(shape, dtype):
    nimg, Hp, Wp = shape
    t = DT[dtype]
    g = torch.Generator().manual_seed(Hp * 100 + Wp)
    out2 = torch.relu(torch.randn(nimg, Hp + 2, Wp, 64, generator=g)).to(t).to(DEV)         # rows Hp, Hp + 1 are the layout's pad rows
    pooled = torch.full((nimg, 64 * 16), 7.0, device=DEV, dtype=t)
    call("eg_spec_avgpool_fwd", ptr(out2), ptr(pooled), nimg, Hp, Wp, dtype, 0)
    torch.cuda.synchronize()
    x = out2[:, :Hp].double().permute(0, 3, 1, 2)                                            # [n][c][y][x]
    ref = torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(nimg, 64 * 16)
    tol = dict(rtol=1e-2, atol=1e-2) if dtype != L.EG_F32 else dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pooled.double(), ref, **tol)
    dpooled = torch.randn(nimg, 64 * 16, generator=g).to(t).to(DEV)
    d2 = torch.zeros(nimg, Hp + 2, Wp + 4, 64, device=DEV, dtype=t)
    call("eg_spec_avgpool_bwd", ptr(out2), ptr(dpooled), ptr(d2), nimg, Hp, Wp, dtype, 0)
    torch.cuda.synchronize()
    wy, wx = Hp // 4, Wp // 4
    gfull = dpooled.double().view(nimg, 64, 4, 4).repeat_interleave(wy, 2).repeat_interleave(wx, 3) / (wy * wx)   # [n][c][Hp][Wp]
    refd = torch.where(x > 0, gfull, torch.zeros_like(gfull)).permute(0, 2, 3, 1)                                   # [n][y][x][c]
    torch.testing.assert_close(d2[:, 1:Hp + 1, 1:Wp + 1].double(), refd, **tol)
    assert float(d2[:, 0].abs().sum()) == 0 and float(d2[:, :, 0].abs().sum()) == 0         # the pad frame is not written


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F32])
@pytest.mark.parametrize("shape", [(20000, 64, 512), (4099, 64, 16), (3000, 256, 64), (1000, 136, 7), (513, 1024, 512)])
def test_colsum(shape, dtype):
    M, N, nblk = shape
    g = torch.Generator().manual_seed(M + N)
    Y = torch.randn(M, N, generator=g).to(DT[dtype]).to(DEV)
    part = torch.full((nblk, N), 7.0, device=DEV)
    call("eg_colsum", ptr(Y), rowmap(N), M, N, ptr(part), nblk, dtype, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(part.double().sum(0), Y.double().sum(0), rtol=1e-4, atol=2e-3 * (M ** 0.5) / 30)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("shape,want", [((6, 8, 8), 3), ((40, 32, 8), 64), ((3, 8, 4), 1), ((9, 16, 24), 7), ((2, 4, 12), 512)])
def test_conv2d_wgrad_flat(shape, want, dtype):
    """dW of Conv2d(32, 64, 3, padding=1) from the padded pixel rows the forward / pooling kernels keep (p1 [nimg, Hp+2, Wp+4, 32],
    d2 [nimg, Hp+2, Wp+4, 64], image at rows 1..Hp, pixels 1..Wp), against torch's grad_weight in float64 on the same rounded
    operands; splits that end inside a stage, a last split shorter than the others and more splits than 256-pixel stages."""
    nimg, Hp, Wp = shape
    t = DT[dtype]
    rowpx = Wp + 4
    g = torch.Generator().manual_seed(nimg * 1000 + Hp * 10 + Wp)
    x = torch.randn(nimg, Hp, Wp, 32, generator=g).to(t)                     # conv-2 input (post-pool activations), channel-last
    dy = (torch.randn(nimg, Hp, Wp, 64, generator=g) * 0.1).to(t)            # gradient of the conv-2 output
    Q = nimg * (Hp + 2) * rowpx
    p1 = torch.zeros(Q + 4 * rowpx, 32, dtype=t)
    d2 = torch.zeros(Q + 4 * rowpx, 64, dtype=t)
    p1[:Q].view(nimg, Hp + 2, rowpx, 32)[:, 1:Hp + 1, 1:Wp + 1] = x
    d2[:Q].view(nimg, Hp + 2, rowpx, 64)[:, 1:Hp + 1, 1:Wp + 1] = dy
    splits = L.lib().eg_conv2d_wgrad_flat_splits(Q, want)
    assert 1 <= splits <= want
    partial = torch.full((splits, 64, 384), float("nan"), device=DEV)
    p1d, d2d = p1.to(DEV), d2.to(DEV)
    bpart = torch.full((splits, 64), float("nan"), device=DEV)
    call("eg_conv2d_wgrad_flat", ptr(d2d), ptr(p1d), ptr(partial), ptr(bpart), Q, Q + 4 * rowpx, rowpx, splits, dtype, 0)
    dW = torch.full((64, 32, 3, 3), 7.0, device=DEV)
    call("eg_unpack_conv2d_wgrad", ptr(partial), ptr(dW), splits, 64, 32, 0)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (64, 32, 3, 3), dy.double().permute(0, 3, 1, 2), padding=1)
    scale = float(ref.abs().max())
    assert float((dW.double().cpu() - ref).abs().max()) < 2e-5 * scale + 1e-6, float((dW.double().cpu() - ref).abs().max()) / scale
    refb = dy.double().sum((0, 1, 2))                                       # grad_bias = column sums of the gradient rows
    torch.testing.assert_close(bpart.double().sum(0).cpu(), refb, rtol=2e-5, atol=2e-5 * float(refb.abs().max()) + 1e-6)
    # without the bias slot the same weight-gradient slabs come out
    partial2 = torch.full_like(partial, float("nan"))
    call("eg_conv2d_wgrad_flat", ptr(d2d), ptr(p1d), ptr(partial2), 0, Q, Q + 4 * rowpx, rowpx, splits, dtype, 0)
    dW2 = torch.zeros_like(dW)
    call("eg_unpack_conv2d_wgrad", ptr(partial2), ptr(dW2), splits, 64, 32, 0)
    torch.cuda.synchronize()
    assert torch.equal(dW, dW2)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("shape,splits", [((6, 8, 8), 3), ((40, 32, 8), 512), ((3, 8, 4), 1), ((9, 16, 24), 7)])
def test_conv2d_flat_forward(shape, splits, dtype):
    """Conv2d(32, 64, 3, padding=1) + ReLU from the padded pixel rows against torch's conv2d in float64 on the same rounded operands
    (weights packed by eg_pack_conv2d_weight); every row of out2 [nimg, Hp+2, Wp, 64] is written, the image rows are compared."""
    nimg, Hp, Wp = shape
    t = DT[dtype]
    rowpx = Wp + 4
    g = torch.Generator().manual_seed(nimg * 1000 + Hp * 10 + Wp + 1)
    x = torch.randn(nimg, Hp, Wp, 32, generator=g).to(t)
    wt = (torch.randn(64, 32, 3, 3, generator=g) * 0.1)
    bias = torch.randn(64, generator=g)
    Q = nimg * (Hp + 2) * rowpx
    p1 = torch.zeros(Q + 4 * rowpx, 32, dtype=t)
    p1[:Q].view(nimg, Hp + 2, rowpx, 32)[:, 1:Hp + 1, 1:Wp + 1] = x
    wd = torch.zeros(64, 384, device=DEV, dtype=t)
    call("eg_pack_conv2d_weight", ptr(wt.to(DEV)), ptr(wd), 64, 32, 0, dtype, 0)
    out2 = torch.full((nimg, Hp + 2, Wp, 64), 7.0, device=DEV, dtype=t)
    p1d, bd = p1.to(DEV), bias.to(DEV)
    call("eg_conv2d_flat", ptr(p1d), ptr(wd), ptr(bd), ptr(out2), Q, Q + 4 * rowpx, rowpx, Wp, 32, 64, L.ACT_RELU, splits, dtype, 0)
    torch.cuda.synchronize()
    ref = torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wd.cpu().double().view(64, 3, 4, 32)[:, :, :3].permute(0, 3, 1, 2),
                                                bias.double(), padding=1)).permute(0, 2, 3, 1)
    # the flat index places output pixel (y, x) of the reference's padded convolution one row and one pixel up-left of its window's
    # origin: out2 row y holds the window whose top-left corner is padded row y, i.e. image row y (the layout eg_spec_avgpool_fwd reads)
    got = out2[:, :Hp].double().cpu()
    tol = 2e-2 if dtype == L.EG_BF16 else 3e-3
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    assert float((out2[:, Hp:].float() - 7.0).abs().min()) > 0            # the layout's two pad rows are written too (values unused)


@pytest.mark.parametrize("dtype", [L.EG_BF16, L.EG_F16])
@pytest.mark.parametrize("shape,splits", [((6, 8, 8), 3), ((40, 32, 8), 512), ((3, 8, 4), 1), ((9, 16, 24), 7)])
def test_conv2d_flat_backward_data(shape, splits, dtype):
    """grad_input of Conv2d(32, 64, 3, padding=1): dp1 [nimg, Hp+2, Wp, 32] from the padded gradient rows d2 and the transposed weight
    packing, against torch's conv2d_input in float64 on the same rounded operands."""
    nimg, Hp, Wp = shape
    t = DT[dtype]
    rowpx = Wp + 4
    g = torch.Generator().manual_seed(nimg * 1000 + Hp * 10 + Wp + 2)
    dy = torch.randn(nimg, Hp, Wp, 64, generator=g).to(t)
    wt = (torch.randn(64, 32, 3, 3, generator=g) * 0.1)
    Q = nimg * (Hp + 2) * rowpx
    d2 = torch.zeros(Q + 4 * rowpx, 64, dtype=t)
    d2[:Q].view(nimg, Hp + 2, rowpx, 64)[:, 1:Hp + 1, 1:Wp + 1] = dy
    wT = torch.zeros(32, 768, device=DEV, dtype=t)
    call("eg_pack_conv2d_weight", ptr(wt.to(DEV)), ptr(wT), 64, 32, 1, dtype, 0)
    dp1 = torch.full((nimg, Hp + 2, Wp, 32), 7.0, device=DEV, dtype=t)
    d2d = d2.to(DEV)
    call("eg_conv2d_flat", ptr(d2d), ptr(wT), 0, ptr(dp1), Q, Q + 4 * rowpx, rowpx, Wp, 64, 32, L.ACT_NONE, splits, dtype, 0)
    torch.cuda.synchronize()
    # the rounded weights, back in [n][c][ky][kx]: the transposed packing holds w[n][c][2-ky][2-kx] at [c][(ky*4+kx)*64 + n]
    wr = wT.cpu().double().view(32, 3, 4, 64)[:, :, :3].flip(1, 2).permute(3, 0, 1, 2)
    ref = torch.nn.grad.conv2d_input((nimg, 32, Hp, Wp), wr, dy.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    tol = 2e-2 if dtype == L.EG_BF16 else 3e-3
    torch.testing.assert_close(dp1[:, :Hp].double().cpu(), ref, rtol=tol, atol=tol * float(ref.abs().max()))
