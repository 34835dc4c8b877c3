"""GPU: operator-level parity of the HIP kernels (called through the C ABI) against plain PyTorch fp32
references computed on the CPU.  Tolerances: 2e-4 relative to the output scale (fp32, different
summation order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _ref_input(x_nhwc, in_mode, sc, sh):
    x = x_nhwc.permute(0, 3, 1, 2)
    if in_mode >= 1:
        x = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if in_mode == 2:
        x = F.max_pool2d(x, 2)
    return x


@pytest.fixture(params=[1, 0, 6, 10, 11], ids=["winograd_pipelined", "direct", "winograd_two_workgroups", "winograd_f4x4",
                                                "wgrad_f3x3_4x4"])
def conv_algo(request):
    """ssp_set_conv_algo: every convolution / weight-gradient operator test runs under the fp32 implementations
    (1 = default: software-pipelined Winograd; 0 = direct implicit GEMM; [2 = Winograd without the software pipeline, 5 = the
    pipelined kernel with LDS-staged weights and 3 / 7 / 8 = the bf16-operand experiments inside the Winograd kernels are compiled
    out of the shipped library: SSP_LEGACY_ALGOS];
    6 = the second-generation pipelined Winograd kernel: two independent 4-wave workgroups per CU;
    10 = Winograd F(4x4,3x3) (conv_wino4_kernel) on every 3x3 convolution it can run, F(2x2,3x3) elsewhere;
    11 = algorithm 1 with the Winograd F(3x3,4x4) weight gradient, wgrad_wino4_kernel (opt-in: correct, not faster))."""
    from semantic_superpoint_amd import lib as L
    L.set_conv_algo(request.param)
    yield request.param
    L.set_conv_algo(1)


CONV_CASES = [
    # N, H, W, cin, cout, ks, in_mode   (H,W = conv resolution; the input is 2H x 2W for in_mode 2)
    (2, 16, 64, 64, 64, 3, 1),    # wide tiles (8x32), full tiles
    (1, 12, 32, 64, 128, 3, 0),   # wide, partial tile rows, 2 cout blocks
    (2, 8, 32, 64, 64, 3, 2),     # wide + pooled input
    (2, 30, 40, 128, 128, 3, 1),  # narrow tiles (32x8): the 30x40 layers
    (1, 8, 24, 64, 128, 3, 2),    # narrow + pooled input
    (2, 6, 8, 128, 256, 3, 1),    # heads on a tiny map
    (2, 30, 40, 256, 65, 1, 1),   # convPb: 1x1, cout tail
    (1, 4, 6, 256, 256, 1, 1),    # convDb
    (1, 16, 32, 256, 133, 1, 0),  # 1x1 raw input, wide
    (3, 7, 9, 40, 200, 1, 1),     # 1x1: partial 32-channel chunk under BatchNorm-on-load, 7 n-tiles = two grouped problems, ragged pixels
    (1, 3, 5, 300, 20, 1, 0),     # 1x1: ten chunks, one partial n-tile, 15 pixels
    (1, 9, 11, 64, 64, 3, 1),     # odd map: partial 2x2 Winograd tiles at the right / bottom edge
    (2, 14, 20, 32, 64, 3, 0),    # 32 input channels (four 8-channel stages), narrow partial tiles
    (1, 8, 32, 48, 70, 3, 1),     # Cin = 48, cout tail (70 = 64 + 6)
    (3, 10, 64, 16, 64, 3, 0),    # Cin = 16: two stages per tile, many tiles per block
    (2, 50, 70, 64, 128, 3, 1),   # several F(4x4,3x3) tile blocks per image, partial at both edges (32x16 blocks: 50 = 3 x 16 + 2)
    (1, 70, 50, 32, 64, 3, 0),    # the 16x32 block shape of F(4x4,3x3) (70 = 2 x 32 + 6, 50 = 3 x 16 + 2)
]


@pytest.mark.parametrize("N,H,W,cin,cout,ks,mode", CONV_CASES)
def test_conv_forward(N, H, W, cin, cout, ks, mode, conv_algo):
    from semantic_superpoint_amd import lib as L
    dev = _dev()
    rs = np.random.RandomState(N * 1000 + H * 10 + cout + ks + mode)
    mul = 2 if mode == 2 else 1
    x = torch.from_numpy(rs.randn(N, H * mul, W * mul, cin).astype(np.float32))
    w = torch.from_numpy((rs.randn(cout, cin, ks, ks) / np.sqrt(cin * ks * ks)).astype(np.float32))
    b = torch.from_numpy(rs.randn(cout).astype(np.float32) * 0.1)
    sc = torch.from_numpy(rs.uniform(-1.5, 1.5, cin).astype(np.float32))  # negative scales too
    sh = torch.from_numpy(rs.uniform(-0.5, 0.5, cin).astype(np.float32))
    ref = F.conv2d(_ref_input(x, mode, sc, sh), w, b, padding=ks // 2).permute(0, 2, 3, 1).contiguous()
    stats = torch.zeros(L.NREP, 2 * cout, dtype=torch.float64, device=dev)
    out = L.op_conv(x.to(dev), w.to(dev), b.to(dev), ks, mode, sc.to(dev), sh.to(dev), stats)
    torch.cuda.synchronize()
    stats = stats.sum(0)
    assert _rel(out.cpu(), ref) < 2e-4
    s_ref = ref.double().sum(dim=(0, 1, 2))
    q_ref = (ref.double() ** 2).sum(dim=(0, 1, 2))
    assert (stats[:cout].cpu() - s_ref).abs().max() < 1e-3 * (s_ref.abs().max() + 1)
    assert (stats[cout:].cpu() - q_ref).abs().max() < 1e-4 * q_ref.abs().max()


@pytest.mark.parametrize("N,H,W,cin,cout,ks", [(2, 16, 32, 64, 64, 3), (1, 30, 40, 128, 256, 3), (2, 6, 8, 256, 65, 1),
                                                 (1, 8, 32, 64, 128, 3), (2, 30, 40, 256, 133, 1), (1, 5, 7, 256, 256, 1)])
def test_conv_dgrad(N, H, W, cin, cout, ks, conv_algo):
    """data gradient = conv with transposed/flipped weights; `cin/cout` are those of the FORWARD conv."""
    from semantic_superpoint_amd import lib as L
    dev = _dev()
    rs = np.random.RandomState(7 + cin + cout)
    cpad = (cout + 3) // 4 * 4
    dy = torch.zeros(N, H, W, cpad)
    dy[..., :cout] = torch.from_numpy(rs.randn(N, H, W, cout).astype(np.float32))
    w = torch.from_numpy((rs.randn(cout, cin, ks, ks) / np.sqrt(cout * ks * ks)).astype(np.float32))
    ref = torch.nn.grad.conv2d_input((N, cin, H, W), w, dy[..., :cout].permute(0, 3, 1, 2).contiguous(), padding=ks // 2)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    if cpad != cout:  # channel-padded dY as used for convPb (65 -> 68 readable channels, zero pads)
        w = torch.cat([w, torch.zeros(cpad - cout, cin, ks, ks)], 0).contiguous()
    out = L.op_conv(dy.to(dev), w.to(dev), None, ks, 0, None, None, None, transpose_flip=True)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) < 2e-4


@pytest.mark.parametrize("N,H,W,cin,cout,ks,mode", [(2, 16, 64, 64, 64, 3, 1), (2, 30, 40, 128, 128, 3, 1),
                                                      (2, 8, 32, 64, 128, 3, 2), (1, 12, 24, 128, 256, 3, 0),
                                                      (2, 30, 40, 256, 64, 1, 1), (3, 5, 8, 256, 256, 1, 0),
                                                      (2, 30, 40, 256, 133, 1, 1), (1, 3, 5, 256, 65, 1, 1), (2, 15, 20, 256, 256, 1, 1),
                                                      (1, 9, 11, 64, 64, 3, 1),     # odd map (direct kernel)
                                                      (2, 10, 12, 48, 70, 3, 0),    # even map, ragged tiles, tails
                                                      (1, 36, 64, 64, 64, 3, 1),    # several block tiles per image
                                                      (2, 24, 128, 64, 64, 3, 1),   # 4x32 tiles with interior ones
                                                      (1, 24, 128, 128, 64, 3, 0),  # (scalar tile origin, no masks)
                                                      (1, 56, 40, 64, 128, 3, 1)])  # 16x8 tiles with interior ones
def test_conv_wgrad(N, H, W, cin, cout, ks, mode, conv_algo):
    from semantic_superpoint_amd import lib as L
    dev = _dev()
    rs = np.random.RandomState(11 + cin + cout + mode)
    mul = 2 if mode == 2 else 1
    x = torch.from_numpy(rs.randn(N, H * mul, W * mul, cin).astype(np.float32))
    dy = torch.from_numpy(rs.randn(N, H, W, cout).astype(np.float32))
    sc = torch.from_numpy(rs.uniform(-1.5, 1.5, cin).astype(np.float32))
    sh = torch.from_numpy(rs.uniform(-0.5, 0.5, cin).astype(np.float32))
    xin = _ref_input(x, mode, sc, sh)
    ref = torch.nn.grad.conv2d_weight(xin, (cout, cin, ks, ks), dy.permute(0, 3, 1, 2).contiguous(), padding=ks // 2)
    out = L.op_conv_wgrad(x.to(dev), dy.to(dev), ks, mode, sc.to(dev), sh.to(dev))
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) < 2e-4


def test_labels_bit_exact(golden_dir):
    """labels2Dto3D / getMasks: bit-exact against the reference's arrays (G2) for binary labels,
    1e-7 for gaussian-valued labels (64-term fp32 sum in a different order)."""
    from semantic_superpoint_amd import lib as L
    from tests import golden_util as G
    dev = _dev()
    g = G.load("g2_labels.npz")
    tgt, cm = L.op_labels(torch.from_numpy(g["labels_bin"]).to(dev), torch.from_numpy(g["mask"]).to(dev))
    torch.cuda.synchronize()
    assert torch.equal(tgt.cpu(), torch.from_numpy(g["labels3D_bin"]))
    assert torch.equal(cm.cpu(), torch.from_numpy(g["mask3D"]))
    tgt, _ = L.op_labels(torch.from_numpy(g["labels_gauss"]).to(dev), None)
    ref = torch.from_numpy(g["labels3D_gauss"])
    assert (tgt.cpu() - ref).abs().max() < 2e-7
    assert torch.equal(tgt.cpu() == 0, ref == 0)  # indexing: identical support


@pytest.mark.parametrize("N,H,W,C,relu,pool", [(2, 8, 12, 256, True, False), (2, 16, 24, 64, True, True),
                                               (2, 8, 12, 65, False, False), (1, 30, 40, 128, True, False),
                                               (2, 12, 16, 128, True, True)])
def test_bn_relu_pool_backward(N, H, W, C, relu, pool):
    """BatchNorm2d(train)+ReLU(+MaxPool2d) backward vs torch autograd (CPU fp32)."""
    from semantic_superpoint_amd import lib as L
    dev = _dev()
    rs = np.random.RandomState(C + H)
    y = torch.from_numpy(rs.randn(N, C, H, W).astype(np.float32) * 1.7 + 0.3).requires_grad_(True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).requires_grad_(True)
    beta = torch.from_numpy(rs.uniform(-0.3, 0.3, C).astype(np.float32)).requires_grad_(True)
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    a = F.relu(z) if relu else z
    if pool:
        a = F.max_pool2d(a, 2)
    g = torch.from_numpy(rs.randn(*a.shape).astype(np.float32))
    (a * g).sum().backward()
    with torch.no_grad():
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        scale = gamma * invstd
        shift = beta - mean * scale
    Cp = (C + 3) // 4 * 4
    def nhwc(t, c_pad=Cp):
        o = torch.zeros(t.shape[0], t.shape[2], t.shape[3], c_pad)
        o[..., :t.shape[1]] = t.detach().permute(0, 2, 3, 1)
        return o.contiguous().to(dev)
    dy, dg, db, dbias = L.op_bn_bwd(nhwc(y), nhwc(g), gamma.detach().to(dev), scale.to(dev), shift.to(dev), mean.to(dev),
                                    invstd.to(dev), relu, pool)
    torch.cuda.synchronize()
    ref = y.grad.permute(0, 2, 3, 1)
    assert _rel(dy.cpu()[..., :C], ref) < 1e-4  # C = 65 rides in 68-float pixels like the detector head
    assert _rel(dg.cpu(), gamma.grad) < 1e-4 and _rel(db.cpu(), beta.grad) < 1e-4


def test_pair_construction_kernels_golden(golden_dir):
    """Row a15: device warps vs the reference's arrays (G7: inv_warp_image_batch, compute_valid_mask with erosion 0,
    warpLabels) and vs the oracle's erosion."""
    from semantic_superpoint_amd import lib as L
    from oracle import cpu_ref as C
    from tests import golden_util as G
    dev = _dev()
    g = G.load("g7_warps.npz")
    Hs = torch.from_numpy(g["H"])
    inv = torch.inverse(Hs).contiguous()
    img = torch.from_numpy(g["img"]).to(dev)
    w = L.op_warp_image(img, inv)
    assert (w.cpu() - torch.from_numpy(g["warped"])).abs().max() < 1e-5
    ones = torch.ones_like(img)
    m = L.op_warp_image(ones, inv, nearest=True)
    ref_m = torch.from_numpy(g["mask"]).view_as(m.cpu())
    assert float((m.cpu() != ref_m).float().mean()) < 1e-3  # nearest ties at .5 may flip
    er = L.op_erode(m, 3).cpu()
    er_ref = torch.stack([C.erode_ellipse(m.cpu()[i, 0], 3) for i in range(4)]).unsqueeze(1)
    assert torch.equal(er, er_ref)
    Hh, Ww = img.shape[2:]
    for i in range(4):
        lab = torch.zeros(1, 1, Hh, Ww)
        pts = torch.from_numpy(g["pts%d" % i].astype(np.int64))
        lab[0, 0, pts[:, 1], pts[:, 0]] = 1
        out = L.op_warp_labels(lab.to(dev), Hs[i:i + 1]).cpu()  # host-scaled pixel homography: the reference's rounding
        ref = torch.from_numpy(g["wlabels%d" % i]).view(1, 1, Hh, Ww)
        assert torch.equal(out, ref), i
        out = L.op_warp_labels(lab.to(dev), Hs[i:i + 1], exact=False).cpu()
        assert float((out != ref).float().sum()) <= 2, i  # rounding ties of the analytic T^-1 H T



# ------------------------------------------------------------------------------------------------
# segmentation loss as an operator (ssp_op_sem_loss): both lane layouts against autograd of F.interpolate + cross_entropy in fp64
# ------------------------------------------------------------------------------------------------
def _sem_ref(logits, labels, C):
    x = logits.double().requires_grad_(True)
    up = F.interpolate(x, scale_factor=8, mode="bilinear", align_corners=False)
    lab = labels.clone()
    lab[(lab < 0) | (lab > C)] = C     # the kernels ignore every value outside [0, C)
    loss = F.cross_entropy(up, lab, ignore_index=C)
    loss.backward()
    return float(loss.detach()), x.grad.float()


def _sem_labels(B, H, W, C, kind, gen):
    if kind == "noise":      # every pixel its own class: worst case of the label histogram
        lab = torch.randint(0, C, (B, H, W), generator=gen)
    else:                    # segments: a few classes per 8x8 tile, like a segmentation map
        coarse = torch.randint(0, C, (B, 1, (H + 23) // 24, (W + 23) // 24), generator=gen).float()
        lab = F.interpolate(coarse, size=(H, W), mode="nearest")[:, 0].long()
    lab[0, :9, :] = C                    # ignored rows: the first tile row of image 0 has no counted pixel at all
    lab[-1, H // 2:H // 2 + 3, 5:40] = C
    lab[-1, -1, -1] = 255                # values a dataset may hold for "void": ignored like C
    lab[-1, -2, -1] = -1
    return lab


SEM_CASES = [
    # B, Hc, Wc, C, scale of the logits, labels
    (2, 8, 12, 133, 3.0, "segments"),
    (2, 8, 12, 133, 3.0, "noise"),
    (1, 5, 7, 133, 40.0, "segments"),     # sharp softmax: the shift bound matters
    (1, 4, 4, 133, 25.0, "noise"),
    (3, 3, 9, 21, 2.0, "segments"),       # two class blocks
    (1, 6, 5, 144, 2.0, "noise"),         # every class slot of the (x, class) form in use
    (1, 6, 5, 1, 2.0, "segments"),
    (1, 30, 40, 133, 3.0, "segments"),    # one image of the training shape: 8 row groups, the column range split over workgroups
]


@pytest.mark.parametrize("algo", [1, 2], ids=["pixels_then_classes", "x_class_lanes"])
@pytest.mark.parametrize("B,Hc,Wc,C,scale,kind", SEM_CASES)
def test_sem_loss_operator_vs_autograd(B, Hc, Wc, C, scale, kind, algo):
    from semantic_superpoint_amd import lib as L
    gen = torch.Generator().manual_seed(1000 * C + Hc)
    logits = torch.randn(B, C, Hc, Wc, generator=gen) * scale
    labels = _sem_labels(B, 8 * Hc, 8 * Wc, C, kind, gen)
    ref_loss, ref_grad = _sem_ref(logits, labels, C)
    loss, grad = L.op_sem_loss(logits.to(_dev()), labels, algo=algo)
    assert abs(loss - ref_loss) < 2e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    assert (grad.cpu() - ref_grad).abs().max() < 1e-7 + 2e-5 * float(ref_grad.abs().max())
    fwd_only, none = L.op_sem_loss(logits.to(_dev()), labels, grad=False, algo=algo)
    assert none is None and abs(fwd_only - loss) < 1e-6 * max(1.0, abs(loss))


def test_sem_loss_operator_dispatch_and_limits():
    """algo 0 is what the step runs: the (x, class) form up to 144 classes, the other form above; 2 refuses what it cannot do."""
    from semantic_superpoint_amd import lib as L
    gen = torch.Generator().manual_seed(7)
    for C in (133, 150):
        logits = torch.randn(1, C, 4, 6, generator=gen) * 2
        labels = _sem_labels(1, 32, 48, C, "segments", gen)
        ref_loss, ref_grad = _sem_ref(logits, labels, C)
        loss, grad = L.op_sem_loss(logits.to(_dev()), labels, algo=0)
        assert abs(loss - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
        assert (grad.cpu() - ref_grad).abs().max() < 1e-7 + 2e-5 * float(ref_grad.abs().max())
    with pytest.raises(RuntimeError):
        L.op_sem_loss(torch.randn(1, 150, 4, 6).to(_dev()), torch.zeros(1, 32, 48, dtype=torch.long), algo=2)
