"""The encoder's stem in-tree (csrc/stem.hip: peclr_stem_conv7x7_s2 + peclr_stem_pack): torchvision ResNet `conv1` = Conv2d(3, 64,
7, stride 2, padding 3, bias=False) behind /root/reference/src/models/resnet_model.py:15, on fp32 NHWC images.  fp32: held to the
error class of an fp32 kernel against float64 (next to MIOpen on the same data); bf16 / fp16 (autocast): one rounding of the
float64 convolution of the ROUNDED operands; fused BatchNorm statistics against a pass over the stored output; bit-repeatable."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _data(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, 3, h, w, generator=g) * 1.1 + 0.2).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(64, 3, 7, 7, generator=g) * 0.08).to(DEV)
    return x, wt


def _ref64(x, wt):
    return torch.nn.functional.conv2d(x.double(), wt.double(), None, 2, 3)


@pytest.mark.parametrize("n,h,w", [(4, 224, 224), (2, 64, 64), (3, 30, 46), (2, 225, 131), (1, 448, 448), (5, 9, 8)])
@pytest.mark.parametrize("weight_format", ["contiguous", "channels_last"])
def test_stem_fp32_is_an_fp32_convolution(n, h, w, weight_format):
    from peclr_amd import _capi as capi

    x, wt = _data(n, h, w, seed=h + w)
    if weight_format == "channels_last":
        wt = wt.contiguous(memory_format=torch.channels_last)
    planes = capi.StemPlanes(wt).pack()
    y = capi.stem_conv(x, planes)
    ref = _ref64(x, wt)
    assert y.shape == ref.shape and y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max()) / scale
    stock = float((torch.nn.functional.conv2d(x, wt, None, 2, 3).double() - ref).abs().max()) / scale
    assert err <= max(4 * stock, 4e-6), (err, stock)
    for _ in range(5):
        assert torch.equal(capi.stem_conv(x, planes), y)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_non_finite_pixel_reaches_exactly_the_outputs_a_7x7_window_connects_it_to(dtype):
    """K is padded with a zero-weighted eighth tap per filter row that points at a real neighbouring pixel; that operand is zeroed
    as well, so an inf there does not turn into a NaN (0 * inf) in an output a true 7x7 convolution keeps finite."""
    from peclr_amd import _capi as capi

    x, wt = _data(2, 40, 72, seed=11)
    bad = [(0, 1, 17, 33), (1, 0, 0, 0), (1, 2, 39, 71), (0, 0, 20, 8)]
    for (n, c, h, w) in bad:
        x[n, c, h, w] = float("inf")
    y = capi.stem_conv(x, capi.StemPlanes(wt, dtype).pack()).float()
    touched = torch.zeros(2, 1, 20, 36, dtype=torch.bool, device=DEV)
    for (n, c, h, w) in bad:
        for oh in range(20):
            for ow in range(36):
                if 0 <= h - (2 * oh - 3) < 7 and 0 <= w - (2 * ow - 3) < 7:
                    touched[n, 0, oh, ow] = True
    finite = torch.isfinite(y).all(dim=1, keepdim=True)
    assert bool((finite == ~touched).all()), (int((~finite & ~touched).sum()), int((finite & touched).sum()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w", [(4, 224, 224), (3, 30, 46), (1, 131, 225)])
def test_stem_16_bit_is_one_rounding_of_the_product_of_the_rounded_operands(dtype, n, h, w):
    from peclr_amd import _capi as capi

    x, wt = _data(n, h, w, seed=h + 2 * w)
    planes = capi.StemPlanes(wt, dtype).pack()
    y = capi.stem_conv(x, planes)
    assert y.dtype == dtype
    ref = _ref64(x.to(dtype), wt.to(dtype))
    # fp32 accumulation of exact products, then one rounding to the 16-bit format: half an ulp of the output -- at most 2^-8
    # (bf16: 8 significand bits) / 2^-11 (fp16) of its magnitude -- plus the fp32 sums' own round-off
    half_ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    bound = ref.abs() * half_ulp * 1.01 + 2e-5 * float(ref.abs().max())
    assert bool(((y.double() - ref).abs() <= bound).all()), float(((y.double() - ref).abs() - bound).max())
    for _ in range(3):
        assert torch.equal(capi.stem_conv(x, planes), y)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,h,w", [(8, 224, 224), (3, 30, 46)])
def test_stem_statistics_from_the_epilogue_equal_a_pass_over_the_output(dtype, n, h, w):
    """The partial table goes through peclr_bn2d_finalize_f32 (bn2d_pool_fwd with `pre`) and must give the mean / invstd a
    separate statistics pass over the stored tensor gives -- tile rows and pixels outside the image excluded."""
    from peclr_amd import _capi as capi

    x, wt = _data(n, h, w, seed=7)
    planes = capi.StemPlanes(wt, dtype).pack()
    g = torch.Generator().manual_seed(1)
    shift = (torch.randn(64, generator=g) * 0.1).to(DEV)
    y, partial, ns = capi.stem_conv(x, planes, stat_shift=shift)
    assert torch.equal(y, capi.stem_conv(x, planes)) and partial.shape == (2 * ns + 1, 64)
    gamma, beta = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)

    def stats(pre):
        rm, rv, nbt = torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
        _, _, _, save, ss = capi.bn2d_pool_fwd(y, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, pre=pre)
        return save, rm, rv

    (s1, rm1, rv1), (s0, rm0, rv0) = stats((partial, ns, shift)), stats(None)
    assert torch.allclose(s1[0], s0[0], rtol=1e-5, atol=2e-6) and torch.allclose(s1[1], s0[1], rtol=2e-5)
    assert torch.allclose(rm1, rm0, rtol=1e-5, atol=1e-6) and torch.allclose(rv1, rv0, rtol=2e-5)
    y64 = y.double()
    assert torch.allclose(s1[0].double(), y64.mean(dim=(0, 2, 3)), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_encoder_routes_the_stem_in_tree_with_its_statistics_and_trains(precision, request):
    """Through the module surface: `ResNetModel.features` hands the stem its BatchNorm, the stem runs in-tree (event log) with
    the statistics fused (no bn2d_stats launch at all in the forward), forward and gradients equal the MIOpen-stem arm's."""
    import copy

    from peclr_amd import _capi as capi
    from peclr_amd import bn2d as B
    from peclr_amd.config import Config
    from peclr_amd.encoder import get_wrapper_model

    # Both arms repeat themselves bit for bit: every convolution the in-tree kernels accept runs on them (`force`: fixed-order weight
    # gradients) and what stays on MIOpen must pick a deterministic algorithm -- left alone, MIOpen's atomically accumulated weight
    # gradients at these small shapes made this comparison a TWO-OUTCOME test (round 6: 1 run in 6 -- and 2 in 3 of the round-5
    # arithmetic -- landed on the same second outcome, median 8.19e-3: one early rectifier decision falling the other way).
    det = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled(), torch.backends.cudnn.deterministic
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.backends.cudnn.deterministic = True
    request.addfinalizer(lambda: (torch.use_deterministic_algorithms(det[0], warn_only=det[1]), setattr(torch.backends.cudnn, "deterministic", det[2])))
    torch.manual_seed(3)
    net = get_wrapper_model(Config({"resnet_size": "18"}), False).to(DEV).to(memory_format=torch.channels_last).train()
    for p in net.final_layer.parameters():
        p.requires_grad_(False)
    other = copy.deepcopy(net)
    B.enable_hip_batchnorm(net)
    B.enable_hip_batchnorm(other)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 3, 96, 96, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(8, 512, generator=g) / 512).to(DEV)
    cast = torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16")

    def run(model, stem):
        capi.EVENT_LOG = {}
        try:
            with B.routing(stem=stem, force=True):
                with cast:
                    y = model(x)
                y.float().backward(gy)
            torch.cuda.synchronize()
            tags = {k: len(v) for k, v in capi.EVENT_LOG.items()}
        finally:
            capi.EVENT_LOG = None
        return y.detach().float(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, tags

    y1, g1, t1 = run(net, True)
    y0, g0, t0 = run(other, False)
    assert t1.get("stem_fwd") == 1 and t1.get("stem_pack") == 1 and "stem_fwd" not in t0, (t1, t0)
    assert t0.get("bn2d_stats", 0) == t1.get("bn2d_stats", 0) + 1, (t1, t0)         # the stem BatchNorm's statistics pass is gone
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))   # noqa: E731
    assert set(g1) == set(g0)
    assert torch.allclose(net.features[1].running_mean, other.features[1].running_mean, rtol=1e-3, atol=2e-4 if precision == "fp32" else 2e-3)
    assert int(net.features[1].num_batches_tracked) == 1
    if precision == "fp32":
        assert rel(y1, y0) <= 2e-5, rel(y1, y0)
        # With both arms deterministic the comparison is a CONSTANT of the build: output 1.7e-6, gradients 2.4e-6 (median) / 5.1e-6
        # (worst) norm-wise at this seed (no rectifier decision falls differently in the two arms; seed 4 has one early in the
        # network: 7.8e-3 on most parameters -- what round 5's bars, 2e-4 / 2e-3 / 1e-1, had been widened for).  A wrong stem output
        # or a missing statistics hand-over is O(0.1 - 1).
        rels = sorted(rel(g1[n], g0[n]) for n in g0)
        print(f"stem arms, fp32: output {rel(y1, y0):.2e}, gradients median {rels[len(rels) // 2]:.2e} worst {rels[-1]:.2e}")
        assert rels[len(rels) // 2] <= 3e-5 and rels[-1] <= 1e-4, (rels[len(rels) // 2], rels[-1])
        return
    # bf16: eight images through twenty train-mode BatchNorm layers amplify one-ulp differences of the stem's output (the two
    # arms round differently: one rounding of the exact product here, MIOpen's kernel there) into O(1) differences of single
    # gradients -- so both arms are held against the fp32 run of the same network, and the in-tree arm must be no further from
    # it than the MIOpen arm is
    ref_net = copy.deepcopy(other)
    for m in (ref_net,):
        m.zero_grad(set_to_none=True)
    B.enable_hip_batchnorm(ref_net)
    capi.EVENT_LOG = None
    with B.routing(stem=True):
        yr = ref_net(x)
    yr.backward(gy)
    gr = {n: p.grad.detach().clone() for n, p in ref_net.named_parameters() if p.grad is not None}
    e1, e0 = rel(y1, yr), rel(y0, yr)
    assert e1 <= 1.5 * e0 + 1e-3, (e1, e0)
    w1, w0 = max(rel(g1[n], gr[n]) for n in gr), max(rel(g0[n], gr[n]) for n in gr)
    m1, m0 = sum(rel(g1[n], gr[n]) for n in gr) / len(gr), sum(rel(g0[n], gr[n]) for n in gr) / len(gr)
    print(f"bf16 vs fp32: in-tree stem worst {w1:.3f} mean {m1:.3f} | MIOpen stem worst {w0:.3f} mean {m0:.3f}")
    assert m1 <= 1.5 * m0 + 1e-2 and w1 <= 2.0 * w0 + 5e-2, (w1, w0, m1, m0)


@pytest.mark.parametrize("n,h,w", [(4, 224, 224), (2, 64, 64), (3, 30, 46), (2, 225, 131), (1, 448, 448), (5, 9, 8), (37, 32, 32)])
def test_stem_weight_gradient_fp32_is_an_fp32_weight_gradient(n, h, w):
    """peclr_stem_wgrad, fp32: against float64 next to MIOpen's fp32 weight gradient on the same data; every tap, image borders
    and ragged tiles (widths that are not multiples of 64 output pixels) included; bit-identical when repeated (fixed-order slabs;
    MIOpen's kernel accumulates with atomics)."""
    from peclr_amd import _capi as capi

    x, wt = _data(n, h, w, seed=3 * h + w)
    g = torch.Generator().manual_seed(h)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    gy = torch.randn(n, 64, ho, wo, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    dw = capi.stem_wgrad(gy, x)
    assert dw.shape == (64, 3, 7, 7) and dw.dtype == torch.float32
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), wt.double(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    stock = torch.ops.aten.convolution_backward(gy, x, wt, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    scale = float(ref.abs().max())
    err, err_stock = float((dw.double() - ref).abs().max()) / scale, float((stock.double() - ref).abs().max()) / scale
    assert err <= max(4 * err_stock, 4e-6), (err, err_stock)
    for _ in range(4):
        assert torch.equal(capi.stem_wgrad(gy, x), dw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w", [(4, 224, 224), (3, 30, 46), (2, 131, 225)])
def test_stem_weight_gradient_16_bit_is_the_fp32_sum_of_the_rounded_operands_products(dtype, n, h, w):
    """16-bit gradient, images rounded to the format: exact products accumulated in fp32 -> fp32 gradient of the master weight."""
    from peclr_amd import _capi as capi

    x, wt = _data(n, h, w, seed=h + 5 * w)
    g = torch.Generator().manual_seed(w)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    gy = torch.randn(n, 64, ho, wo, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    dw = capi.stem_wgrad(gy, x)
    ref = torch.ops.aten.convolution_backward(gy.double(), x.to(dtype).double(), wt.double(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    err = float((dw.double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 3e-6 * max(1.0, (n * ho * wo / 4096) ** 0.5), err
    for _ in range(3):
        assert torch.equal(capi.stem_wgrad(gy, x), dw)


def test_no_miopen_convolution_is_left_in_the_fp32_encoder_step():
    """With the stem's forward and weight gradient in-tree every convolution launch of a ResNet-50 fp32 forward + backward is an
    in-tree kernel: the event log (one entry per in-tree launch) accounts for 53 forward convolutions, and the stem's weight
    gradient equals MIOpen's to fp32 round-off."""
    import copy

    from peclr_amd import _capi as capi
    from peclr_amd import bn2d as B
    from peclr_amd.config import Config
    from peclr_amd.encoder import get_wrapper_model

    torch.manual_seed(2)
    net = get_wrapper_model(Config({"resnet_size": "50"}), False).to(DEV).to(memory_format=torch.channels_last).train()
    for p in net.final_layer.parameters():
        p.requires_grad_(False)
    other = copy.deepcopy(net)
    B.enable_hip_batchnorm(net)
    B.enable_hip_batchnorm(other)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(8, 3, 224, 224, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(8, 2048, generator=g) / 2048).to(DEV)

    def run(model, on):
        capi.EVENT_LOG = {}
        try:
            with B.routing(stem_wgrad=on, force=True):
                model(x).backward(gy)
            torch.cuda.synchronize()
            return {k: len(v) for k, v in capi.EVENT_LOG.items()}
        finally:
            capi.EVENT_LOG = None

    t1, t0 = run(net, True), run(other, False)
    fwd = sum(t1.get(k, 0) + t1.get(k + "~hbm", 0) for k in ("stem_fwd", "conv1x1_fwd", "conv3x3_fwd", "conv_s2_fwd"))
    assert fwd == 53, t1
    assert t1.get("stem_wgrad") == 1 and "stem_wgrad" not in t0
    a, b = net.features[0].weight.grad, other.features[0].weight.grad
    assert a.stride() == b.stride()
    assert float((a - b).norm() / b.norm()) <= 1e-4
