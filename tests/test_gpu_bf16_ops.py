"""GPU: operator-level parity of the bf16 path's kernels (conv algorithm 12, BASELINE configs[3]) through the C ABI against
a CPU restatement of their stated semantics: operands rounded to bf16 (activated input, weights), products accumulated
in fp32 (here: fp64, so the bound is the kernel's own fp32 accumulation), stored tensors rounded to bf16.
Tolerances: a bf16 output must equal bf16(reference) up to ONE unit in the last place of bf16 (<= 2^-7 relative: a value
within fp32 accumulation noise of a rounding boundary may round the other way) plus that accumulation noise (1e-4 of
the output scale at K <= 2304); fp32 outputs within 2e-4 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    return torch.device("cuda:0")


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _ref_operand(x_nhwc_f32, in_mode, sc, sh):
    x = x_nhwc_f32.permute(0, 3, 1, 2)
    if in_mode == 1:  # the kernels evaluate the affine as ONE fp32 fma: exact product + sum in fp64, rounded once
        x = F.relu((x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float())
    return _bf(x)


def _ulp_close(out_bf16, ref_f64, scale):
    """|out - ref| <= one bf16 ulp of max(|ref|, tiny) + accumulation noise"""
    o = out_bf16.to(torch.float64)
    err = (o - ref_f64).abs()
    bound = ref_f64.abs() * 2.0 ** -7 + 1e-4 * scale
    bad = err > bound
    return int(bad.sum()), float((err / (ref_f64.abs() + 1e-3 * scale)).max())


CONV_CASES = [
    # N, H, W, cin, cout, ks, in_mode, x dtype, out dtype, pooled copy
    (2, 32, 32, 64, 64, 3, 1, "bf16", "bf16", True),     # full tiles, BatchNorm + ReLU on load, pooled raw output
    (1, 30, 40, 128, 128, 3, 1, "bf16", "bf16", False),  # ragged tiles, 4 chunks, 2 output blocks
    (2, 24, 48, 64, 128, 3, 0, "bf16", "bf16", False),   # plain input (data-gradient form)
    (1, 12, 20, 128, 256, 3, 1, "bf16", "bf16", False),  # 3x3 heads
    (1, 30, 40, 256, 65, 1, 1, "bf16", "f32", False),    # pointwise, ragged output channels
    (1, 14, 18, 136, 256, 1, 0, "f32", "bf16", False),   # pointwise data gradient: 133 (+3 zero) fp32 channels in
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_bf16_matches_its_semantics(case):
    from semantic_superpoint_amd import lib as L
    N, H, W, cin, cout, ks, in_mode, xdt, odt, pooled = case
    dev = _dev()
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 7 * CONV_CASES.index(case))
    x = torch.randn(N, H, W, cin, generator=g)
    if xdt == "bf16":
        x = _bf(x)
    w = torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    gamma = torch.randn(cout, generator=g)
    xin = _ref_operand(_bf(x) if xdt == "f32" and in_mode == 0 else x, in_mode, sc, sh)
    ref = F.conv2d(xin.double(), _bf(w).double(), b.double(), padding=ks // 2).permute(0, 2, 3, 1).contiguous()
    stats = torch.zeros(32, 2 * cout, dtype=torch.float64, device=dev)
    xd = (x.to(torch.bfloat16) if xdt == "bf16" else x).to(dev).contiguous()
    res = L.op_conv_bf16(xd, w.to(dev), b.to(dev), ks, in_mode=in_mode, in_scale=sc.to(dev), in_shift=sh.to(dev), stats=stats,
                         out_f32=(odt == "f32"), pool_gamma=gamma.to(dev) if pooled else None)
    out, pool = res if pooled else (res, None)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    if odt == "f32":
        err = float((out.cpu().double() - ref).abs().max() / scale)
        assert err < 2e-4, err
        stored = out.cpu().double()
    else:
        nbad, worst = _ulp_close(out.cpu(), ref, scale)
        assert nbad == 0, (nbad, worst)
        stored = out.cpu().double()
    # statistics = sum / sum of squares of the STORED tensor
    s = stats.cpu().sum(0)
    n = N * H * W
    assert torch.allclose(s[:cout] / n, stored.reshape(-1, cout).mean(0), rtol=1e-4, atol=1e-5)
    assert torch.allclose(s[cout:] / n, (stored.reshape(-1, cout) ** 2).mean(0), rtol=1e-4, atol=1e-5)
    if pooled:
        st = stored.permute(0, 3, 1, 2)
        pmax, pmin = F.max_pool2d(st, 2), -F.max_pool2d(-st, 2)
        want = torch.where(gamma.view(1, -1, 1, 1) >= 0, pmax, pmin).permute(0, 2, 3, 1)
        assert torch.equal(pool.cpu().double(), want)


def test_conv_bf16_data_gradient_form():
    """transpose_flip: the convolution with the mirrored / transposed weight image = conv2d_input of the forward layer"""
    from semantic_superpoint_amd import lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    N, H, W, cin, cout = 2, 20, 24, 64, 128     # forward layer cin -> cout; the data gradient maps cout -> cin
    dy = _bf(torch.randn(N, H, W, cout, generator=g))
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cout * 9)
    ref = torch.nn.grad.conv2d_input((N, cin, H, W), _bf(w).double(), dy.permute(0, 3, 1, 2).double(), padding=1)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    out = L.op_conv_bf16(dy.to(torch.bfloat16).to(dev), w.to(dev), None, 3, transpose_flip=True)
    torch.cuda.synchronize()
    nbad, worst = _ulp_close(out.cpu(), ref, float(ref.abs().max()))
    assert nbad == 0, (nbad, worst)


WGRAD_CASES = [
    # N, H, W, cin, cout, ks, in_mode, dy dtype
    (2, 32, 32, 64, 64, 3, 1, "bf16"),     # full tiles, BatchNorm + ReLU on load
    (1, 30, 40, 128, 128, 3, 1, "bf16"),   # ragged tiles, 2 x 2 slabs
    (2, 24, 48, 64, 128, 3, 0, "bf16"),    # plain input
    (1, 14, 18, 256, 136, 1, 1, "f32"),    # pointwise head, ragged output channels (133 + 3 zero)
    # maps with INTERIOR 16x16 tiles (halo inside the map): the staging path without per-slot coordinates / padding masks
    (1, 64, 80, 64, 64, 3, 1, "bf16"),     # 4 x 5 tiles, 2 x 3 of them interior
    (1, 56, 72, 128, 64, 3, 1, "bf16"),    # interior tiles beside ragged ones (56 = 3.5, 72 = 4.5 tiles), two input-channel blocks
    (1, 48, 48, 64, 64, 3, 0, "bf16"),     # plain input: interior tile without the activation pass
    (1, 48, 64, 256, 256, 1, 1, "f32"),    # pointwise, whole channel blocks: every tile is interior
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_wgrad_bf16_matches_its_semantics(case):
    from semantic_superpoint_amd import lib as L
    N, H, W, cin, cout, ks, in_mode, ddt = case
    dev = _dev()
    g = torch.Generator().manual_seed(11 + WGRAD_CASES.index(case))
    x = _bf(torch.randn(N, H, W, cin, generator=g))
    dy = torch.randn(N, H, W, cout, generator=g)
    if ddt == "bf16":
        dy = _bf(dy)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    xin = _ref_operand(x, in_mode, sc, sh)
    ref = torch.nn.grad.conv2d_weight(xin.double(), (cout, cin, ks, ks), _bf(dy).permute(0, 3, 1, 2).double(), padding=ks // 2)
    dyd = (dy.to(torch.bfloat16) if ddt == "bf16" else dy).to(dev).contiguous()
    dw = L.op_conv_wgrad_bf16(x.to(torch.bfloat16).to(dev), dyd, ks, in_mode=in_mode, in_scale=sc.to(dev), in_shift=sh.to(dev))
    torch.cuda.synchronize()
    err = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-4, err   # fp32 accumulation of exact bf16 products
    # accumulation: a second call adds to the gradient
    dw2 = L.op_conv_wgrad_bf16(x.to(torch.bfloat16).to(dev), dyd, ks, in_mode=in_mode, in_scale=sc.to(dev), in_shift=sh.to(dev), dw=dw.clone())
    torch.cuda.synchronize()
    assert float((dw2.cpu().double() - 2 * ref).abs().max() / ref.abs().max()) < 4e-4


@pytest.mark.parametrize("N,H,W,Cc,pool", [(2, 16, 24, 64, False), (2, 16, 24, 64, True), (1, 30, 40, 128, False), (2, 12, 8, 256, True)])
def test_bn_relu_pool_backward_on_bf16_tensors(N, H, W, Cc, pool):
    """BatchNorm(train) + ReLU (+ MaxPool2d(2)) backward with bf16 y / dOut / dY (fp32 arithmetic inside) against autograd in
    fp64 on the same bf16-valued tensors; dY is compared after its rounding to bf16 (one ulp on boundary cases)."""
    from semantic_superpoint_amd import lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(3 + H + Cc)
    y = _bf(torch.randn(N, H, W, Cc, generator=g))
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    dout = _bf(torch.randn(N, Ho, Wo, Cc, generator=g))
    gamma = torch.randn(Cc, generator=g)
    beta = torch.randn(Cc, generator=g) * 0.2
    yd = y.double().permute(0, 3, 1, 2).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = yd.mean(dim=(0, 2, 3)), yd.var(dim=(0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    z = (yd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1) * gd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
    a = F.relu(z)
    if pool:
        a = F.max_pool2d(a, 2)
    a.backward(dout.double().permute(0, 3, 1, 2))
    scale = (gamma.double() * invstd.detach()).float()
    shift = (beta.double() - mean.detach() * scale.double()).float()
    dy, dg, db, dbias = L.op_bn_bwd_bf16(y.to(torch.bfloat16).to(dev), dout.to(torch.bfloat16).to(dev), gamma.to(dev), scale.to(dev),
                                         shift.to(dev), mean.detach().float().to(dev), invstd.detach().float().to(dev), pool=pool)
    torch.cuda.synchronize()
    ref = yd.grad.permute(0, 2, 3, 1)
    nbad, worst = _ulp_close(dy.cpu(), ref, float(ref.abs().max()))
    # a ReLU gate / pool winner whose z sits within fp32 rounding of 0 (or of its neighbour) may go the other way: none expected here
    assert nbad <= 2, (nbad, worst)
    assert (dg.cpu().double() - gd.grad).abs().max() <= 1e-4 * float(gd.grad.abs().max())
    assert (db.cpu().double() - bd.grad).abs().max() <= 1e-4 * float(bd.grad.abs().max())
