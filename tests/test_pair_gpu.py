""""Pair" arithmetic of the packed-weight GEMMs (round 6; include/peclr_hip.h `peclr_x6_pair`): both fp32 operands of the residual
blocks' convolutions (torchvision Bottleneck behind /root/reference/src/models/resnet_model.py:15) multiplied by a per-tensor power
of two and split into two fp16 numbers, three products on v_mfma_f32_32x32x16_f16.  Held against float64 next to the six-product
kernel and the v_mfma_f32 kernel on the same data; bit-repeatable; independent of the tile height; graceful on wide dynamic ranges."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _absmax(t):
    return t.detach().abs().max().reshape(1).float()


def _pair_planes(capi, specs):
    return capi.X6Planes(specs, pair=True).pack()


def _err(c, ref):
    return float((c.double() - ref).abs().max()) / float(ref.abs().max())


@pytest.mark.parametrize("m,n,k", [(8192, 512, 1024), (6272, 256, 64), (3136, 512, 128), (1568, 1024, 256), (392, 2048, 512),
                                   (12544 + 37, 256, 64), (1000, 384, 32), (300, 128, 2048)])
def test_pair_gemm_is_an_fp32_gemm(m, n, k):
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(DEV)
    bt = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
    ref = a.double() @ bt.double().t()
    six = capi.gemm_x6p(a, capi.X6Planes([(bt, False)]).pack().planes[0], n)
    pp = _pair_planes(capi, [(bt, False)])
    am = _absmax(a)
    out = capi.gemm_x6p(a, pp.planes[0], n, pair=(am, pp.scale(0)))
    e_pair, e_six, e_f32 = _err(out, ref), _err(six, ref), _err(capi.gemm(capi.GEMM_NT, a, bt), ref)
    print(f"M={m} N={n} K={k}: err/scale pair {e_pair:.2e}  six-product {e_six:.2e}  v_mfma_f32 {e_f32:.2e}")
    # the error class of an fp32 kernel: no worse than the k-ordered fp32 chain of v_mfma_f32 on the same data (+ 25 %: the two are
    # different roundings of the same sums), and within 2.5 x the six-product kernel's
    assert e_pair <= max(1.25 * e_f32, 4e-7), (e_pair, e_f32)
    assert e_pair <= max(2.5 * e_six, 4e-7), (e_pair, e_six)
    for tile_rows in (128, 256):                                  # the result does not depend on the tile height, nor on the run
        for _ in range(2):
            assert torch.equal(capi.gemm_x6p(a, pp.planes[0], n, tile_rows=tile_rows, pair=(am, pp.scale(0))), out), tile_rows


def test_pair_weight_pack_finds_the_power_of_two():
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(5)
    w1 = (torch.randn(256, 64, generator=g) * 0.03).to(DEV)
    w2 = (torch.randn(64, 128, generator=g) * 700.0).to(DEV)        # packed transposed: B_t = w2^T [128, 64]
    pp = _pair_planes(capi, [(w1, False), (w2, True)])
    torch.cuda.synchronize()
    for i, w in enumerate((w1, w2)):
        mx = float(w.abs().max())
        s = float(pp.scales[i])
        assert s > 0 and torch.frexp(torch.tensor(s))[0] == 0.5, s              # a power of two
        assert 2.0 ** 14 <= mx * s < 2.0 ** 15, (mx, s)
        assert float(pp.absmax[i]) == mx


def test_pair_gemm_wide_dynamic_range_and_special_values():
    """Rows whose magnitudes differ by 2^24: the small rows lose low bits of `lo` gradually -- the absolute error stays below
    2^-22 of (row sum of |a b|) + 2^-38 of max |A| max |B| K, i.e. far below one fp32 ulp of anything the large rows produce."""
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(11)
    m, n, k = 2048, 256, 512
    a = torch.randn(m, k, generator=g) * torch.exp2(-torch.randint(0, 25, (m, 1), generator=g).float())
    a = a.to(DEV)
    bt = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
    pp = _pair_planes(capi, [(bt, False)])
    out = capi.gemm_x6p(a, pp.planes[0], n, pair=(_absmax(a), pp.scale(0)))
    ref = a.double() @ bt.double().t()
    bound = (a.double().abs() @ bt.double().abs().t()) * 2.0 ** -21 + float(a.abs().max()) * float(bt.abs().max()) * k * 2.0 ** -38
    assert bool(((out.double() - ref).abs() <= bound).all()), float(((out.double() - ref).abs() / bound).max())
    z = torch.zeros(256, k, device=DEV)
    assert float(capi.gemm_x6p(z, pp.planes[0], n, pair=(_absmax(z), pp.scale(0))).abs().max()) == 0.0          # absmax = 0
    a2 = a[:256].clone()
    a2[3, 5] = float("inf")
    o2 = capi.gemm_x6p(a2, pp.planes[0], n, pair=(_absmax(a2), pp.scale(0)))
    assert not torch.isfinite(o2[3]).any() and torch.isfinite(o2[:3]).all() and torch.isfinite(o2[4:]).all()   # as an fp32 GEMM would


@pytest.mark.parametrize("with_stats", [False, True])
def test_pair_gemm_epilogues_see_the_unscaled_sums(with_stats):
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(3)
    m, n, k = 6272, 256, 64
    a = torch.randn(m, k, generator=g).to(DEV) * 3.0
    bt = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
    pp = _pair_planes(capi, [(bt, False)])
    pair = (_absmax(a), pp.scale(0))
    plain = capi.gemm_x6p(a, pp.planes[0], n, pair=pair)
    addend = torch.randn(m, n, generator=g).to(DEV)
    assert torch.equal(capi.gemm_x6p(a, pp.planes[0], n, addend=addend, pair=pair), plain + addend)
    if with_stats:
        shift = (torch.randn(n, generator=g) * 0.1).to(DEV)
        out, partial, ns = capi.gemm_x6p(a, pp.planes[0], n, stat_shift=shift, pair=pair)
        assert torch.equal(out, plain)
        d = plain.double() - shift.double()
        sums = partial[:2 * ns].view(ns, 2, n).double().sum(0)
        assert torch.allclose(sums[0], d.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(sums[1], (d * d).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("nb,c,hw", [(8, 64, 56), (8, 128, 28), (16, 256, 14), (32, 512, 7), (3, 128, 9)])
@pytest.mark.parametrize("flip", [False, True])
def test_pair_conv3x3_matches_float64_in_both_forms(nb, c, hw, flip):
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(c + hw)
    x = torch.randn(nb, c, hw, hw, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, generator=g) * 0.03).to(DEV).contiguous(memory_format=torch.channels_last)
    wmat = w.permute(0, 2, 3, 1).reshape(c, 9 * c) if not flip else w.permute(0, 2, 3, 1).reshape(c * 9, c)
    spec = [(wmat.contiguous(), 9 if flip else False)]
    pp = _pair_planes(capi, spec)
    pair = (_absmax(x), pp.scale(0))
    ref = (torch.nn.functional.conv2d(x.double(), w.double(), padding=1) if not flip
           else torch.nn.functional.conv_transpose2d(x.double(), w.double(), padding=1))
    six = capi.conv3x3_x6p(x, capi.X6Planes(spec).pack().planes[0], c, flip=flip)
    outs = [capi.conv3x3_x6p(x, pp.planes[0], c, flip=flip, variant=v, tile_rows=t, pair=pair) for v in (1, 0) for t in (256, 128)]
    e_pair, e_six = max(_err(outs[0], ref), _err(outs[2], ref)), _err(six, ref)
    stock = _err(torch.nn.functional.conv2d(x, w, padding=1) if not flip else torch.nn.functional.conv_transpose2d(x, w, padding=1), ref)
    print(f"3x3 {nb}x{c}x{hw}x{hw} flip={flip}: err/scale pair {e_pair:.2e}  six-product {e_six:.2e}  MIOpen {stock:.2e}")
    assert e_pair <= max(2.5 * e_six, 4 * stock, 6e-7), (e_pair, e_six, stock)
    # each form (halo patch: K ordered (chunk, tap); per tap: (tap, chunk)) gives the same bits at both tile heights and on every run
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[3], outs[2])
    assert torch.equal(capi.conv3x3_x6p(x, pp.planes[0], c, flip=flip, variant=1, tile_rows=256, pair=pair), outs[0])


@pytest.mark.parametrize("cin,cout,taps,hw,nb", [(256, 128, 9, 56, 4), (256, 512, 1, 56, 4), (512, 1024, 1, 28, 4), (128, 128, 9, 28, 8)])
def test_pair_stride_2_forward_matches_float64(cin, cout, taps, hw, nb):
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(cin + cout + taps)
    x = torch.randn(nb, cin, hw, hw, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    ks = 3 if taps == 9 else 1
    w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.03).to(DEV).contiguous(memory_format=torch.channels_last)
    spec = [(w.permute(0, 2, 3, 1).reshape(cout, taps * cin).contiguous(), False)]
    pp = _pair_planes(capi, spec)
    out = capi.conv_s2_x6p(x, pp.planes[0], cout, taps, pair=(_absmax(x), pp.scale(0)))
    six = capi.conv_s2_x6p(x, capi.X6Planes(spec).pack().planes[0], cout, taps)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=ks // 2)
    e_pair, e_six = _err(out, ref), _err(six, ref)
    print(f"stride 2 {cin}->{cout} taps {taps}: err/scale pair {e_pair:.2e}  six-product {e_six:.2e}")
    assert e_pair <= max(2.5 * e_six, 6e-7), (e_pair, e_six)


@pytest.mark.parametrize("cin,cout,ho,nb", [(128, 128, 28, 4), (256, 256, 14, 8)])
def test_pair_stride_2_input_gradient_matches_float64(cin, cout, ho, nb):
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(cin + ho)
    gy = torch.randn(nb, cout, ho, ho, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(DEV).contiguous(memory_format=torch.channels_last)
    spec = [(w.permute(0, 2, 3, 1).reshape(cout * 9, cin).contiguous(), 9)]
    pp = _pair_planes(capi, spec)
    out = capi.conv3x3_s2_dgrad_x6p(gy, pp.planes[0], cin, pair=(_absmax(gy), pp.scale(0)))
    six = capi.conv3x3_s2_dgrad_x6p(gy, capi.X6Planes(spec).pack().planes[0], cin)
    ref = torch.nn.functional.conv_transpose2d(gy.double(), w.double(), stride=2, padding=1, output_padding=1)
    e_pair, e_six = _err(out, ref), _err(six, ref)
    print(f"stride-2 dgrad {cout}->{cin}: err/scale pair {e_pair:.2e}  six-product {e_six:.2e}")
    assert e_pair <= max(2.5 * e_six, 6e-7), (e_pair, e_six)


def _tiny_step(pair_on, seed=3):
    """One training-mode forward + backward of a ResNet-50 on the fused backbone; returns loss, gradients and the kernels that ran."""
    from peclr_amd import _capi as capi
    from peclr_amd import bn2d as B
    from peclr_amd.resnet import resnet50

    torch.manual_seed(seed)
    net = resnet50().to(DEV).to(memory_format=torch.channels_last).train()
    B.enable_hip_batchnorm(net)
    x = torch.randn(16, 3, 96, 96, device=DEV).contiguous(memory_format=torch.channels_last)
    capi.EVENT_LOG = {}
    kernels = {}
    try:
        with B.routing(force=True, x6_pair=pair_on):
            y = net(x)
            loss = y.square().mean() + y.mean()
            loss.backward()
        torch.cuda.synchronize()
        kernels = {k: len(v) for k, v in capi.EVENT_LOG.items()}
    finally:
        capi.EVENT_LOG = None
    B.end_backward()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters()}, kernels, None


def test_the_step_takes_the_pair_kernels_and_agrees_with_the_six_product_step():
    """Every forward and input-gradient GEMM of the residual blocks finds its operand's maximum (left by the BatchNorm pass that
    wrote it) and runs in pair arithmetic; loss and gradients agree with the six-product step to fp32 round-off of a 50-layer net."""
    from peclr_amd import _capi as capi

    launched = []
    orig = capi._pair_arg

    def spy(pair, who):
        launched.append((who, pair is not None))
        return orig(pair, who)

    capi._pair_arg = spy
    try:
        l2, g2, _, _ = _tiny_step(True)
        n_pair = sum(1 for _, on in launched if on)
        n_all = len(launched)
        del launched[:]
        l6, g6, _, _ = _tiny_step(False)
        assert not any(on for _, on in launched)
    finally:
        capi._pair_arg = orig
    # 16 bottlenecks x (3 forward + 3 input-gradient products) + 4 downsample forwards + 4 downsample gradients; the stem's
    # output feeds layer1 through the pooled pass (which leaves its maximum too)
    assert n_all >= 100 and n_pair == n_all, (n_pair, n_all)
    assert abs(l2 - l6) <= 2e-6 * max(1.0, abs(l6)), (l2, l6)
    # gradients: the two arithmetics differ in the last bits of every product; layer4 normalises over 16 x 3 x 3 = 144 rows here, which
    # amplifies that (and flips rectifier decisions at ties) -- norm-wise agreement per tensor; the float64 comparisons of whole
    # networks (tests/test_round4_gpu.py) hold the default (pair) path to its absolute floors
    worst, where = 0.0, None
    for n, a in g2.items():
        b = g6[n]
        rel = float((a - b).norm()) / max(1e-20, float(b.norm()))
        if rel > worst:
            worst, where = rel, n
    print(f"pair vs six-product step: loss {l2} / {l6}, worst norm-wise gradient difference {worst:.2e} ({where})")
    assert worst <= 6e-2, (worst, where)                           # (measured 2.6e-2, a BatchNorm weight of layer2: a sum of cancelling terms)


def test_training_on_the_pair_kernels_tracks_training_on_the_six_product_kernels():
    """Whole steps (Hybrid2Model: encoder, head, alignment, NT-Xent, LARS / Adam) in the two fp32 arithmetics from the same weights
    and batch: the first two steps' losses agree to 2e-6, the third to 5e-3 (differences of a few 1e-7 in the gradients, amplified by
    the optimiser's normalisations and by batch statistics over 144 rows); both runs learn.  (From about the fifth step on the two trajectories separate: this learning
    rate makes the loss non-monotonic, and the curves are two samples of the same chaotic dynamics -- not compared.)"""
    import copy
    import warnings

    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
    from peclr_amd import bn2d as B
    from peclr_amd.bn2d import enable_hip_batchnorm

    warnings.simplefilter("ignore")
    torch.manual_seed(51)
    n = 16
    cfg = hybrid2_config(resnet_size="50", projection_head_input_dim=2048, augmentation=["crop", "rotate"],
                         batch_size=n, num_samples=64, warmup_epochs=1, lr=1e-3, pretrained=False)
    base = Hybrid2Model(cfg).to(DEV).train()
    base.encoder = base.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(base.encoder)
    g = torch.Generator().manual_seed(52)
    batch = {"transformed_image1": torch.randn(n, 3, 96, 96, generator=g), "transformed_image2": torch.randn(n, 3, 96, 96, generator=g),
             "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
             "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
             "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    batch = {k: v.to(DEV) for k, v in batch.items()}
    for k in ("transformed_image1", "transformed_image2"):
        batch[k] = batch[k].contiguous(memory_format=torch.channels_last)
    curves = {}
    for pair in (True, False):
        model = copy.deepcopy(base)
        enable_hip_batchnorm(model.encoder)
        tr = Trainer(max_epochs=10, precision="fp32").attach(model)
        tr.zero_grad()
        with B.routing(force=True, x6_pair=pair):
            curves[pair] = [float(tr.training_micro_step(batch, i)["loss"]) for i in range(5)]
    print("pair", curves[True], "six-product", curves[False])
    assert abs(curves[True][0] - curves[False][0]) <= 2e-6 * curves[False][0]
    assert abs(curves[True][1] - curves[False][1]) <= 2e-6 * curves[False][1]      # (warm-up: the first update is tiny)
    assert abs(curves[True][2] - curves[False][2]) <= 5e-3 * curves[False][2], (curves[True], curves[False])    # (measured: 1.4e-3 ... 2.1e-3)
    # (from the fourth step on the curves are two samples of a chaotic trajectory: 2.91 / 3.06 against 2.92 in two builds of the pair path)
    assert curves[True][4] < curves[True][0] - 0.3 and curves[False][4] < curves[False][0] - 0.3
