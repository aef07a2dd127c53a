"""peclr_gemm_x6s_f32 (csrc/gemm_x6s.hip): the entry-gradient GEMM of a residual block -- dY1 . W1 + the shortcut's gradient (+ the
backward reduction of the BatchNorm the result arrives at) = autograd's "MIOpen input gradient + elementwise add" at a
torchvision Bottleneck's entry (/root/reference/src/models/resnet_model.py:15) -- on the streaming kernel for HBM-bound shapes.
Held to: the SAME output as peclr_gemm_x6p_f32 bit for bit in every addend mode (same split, same products, same order), fp32
accuracy against float64, BatchNorm sums equal to a separate reduction pass, bit-repeatable, and routed by the module where it
applies."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

SHAPES = [(25088, 64, 256), (6272, 128, 512), (4 * 14 * 14 * 8 + 32, 128, 256), (1000, 64, 128), (33, 64, 256)]


def _operands(m, k, n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.randn(m, k, device=DEV, generator=g)
    bt = torch.randn(n, k, device=DEV, generator=g) * 0.05
    d = torch.randn(m, n, device=DEV, generator=g)
    return a, bt, d, g


@pytest.mark.parametrize("m,k,n", SHAPES)
def test_streaming_kernel_equals_the_tiled_kernel_bit_for_bit(m, k, n):
    from peclr_amd import _capi as capi

    a, bt, d, g = _operands(m, k, n, m + k + n)
    pk = capi.X6Planes([(bt, False)]).pack().planes[0]
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (m, n // 32), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)
    assert capi.gemm_x6s_ok(m, n, k)
    plain = capi.gemm_x6s(a, pk, n)
    assert torch.equal(plain, capi.gemm_x6p(a, pk, n))
    ref = a.double() @ bt.double().t()
    assert float((plain.double() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    assert torch.equal(capi.gemm_x6s(a, pk, n, d), capi.gemm_x6p(a, pk, n, d))
    assert torch.equal(capi.gemm_x6s(a, pk, n, d, addend_mask=mask), capi.gemm_x6p(a, pk, n, d, addend_mask=mask))
    for _ in range(6):
        assert torch.equal(capi.gemm_x6s(a, pk, n, d, addend_mask=mask), capi.gemm_x6p(a, pk, n, d, addend_mask=mask))


@pytest.mark.parametrize("imgs,h,w,k,n", [(8, 28, 28, 128, 256), (3, 6, 10, 64, 128), (16, 56, 56, 128, 256)])
def test_streaming_kernel_adds_the_compact_stride_2_gradient_at_the_even_pixels(imgs, h, w, k, n):
    from peclr_amd import _capi as capi

    m = imgs * h * w
    a, bt, _, g = _operands(m, k, n, m + 1)
    half = torch.randn(m // 4, n, device=DEV, generator=g)
    pk = capi.X6Planes([(bt, False)]).pack().planes[0]
    got = capi.gemm_x6s(a, pk, n, half, addend_s2=(h, w))
    assert torch.equal(got, capi.gemm_x6p(a, pk, n, half, addend_s2=(h, w)))
    dense = torch.zeros(imgs, h, w, n, device=DEV)
    dense[:, ::2, ::2] = half.view(imgs, h // 2, w // 2, n)
    assert torch.equal(got, capi.gemm_x6s(a, pk, n, dense.view(m, n)))


@pytest.mark.parametrize("m,k,n", [(25088, 64, 256), (6272, 128, 512), (1000, 64, 128)])
@pytest.mark.parametrize("mode", ["relu_from_x", "relu_mask", "no_relu"])
def test_streaming_kernel_reduces_the_batchnorm_backward_like_a_separate_pass(m, k, n, mode):
    """dgamma / dbeta / the dx coefficients from the kernel's per-wave partial table against peclr_bn2d_bwd_reduce over the
    stored gradient (and against the tiled kernel's epilogue): same sums, another order."""
    from peclr_amd import _capi as capi

    a, bt, d, g = _operands(m, k, n, m + 7)
    pk = capi.X6Planes([(bt, False)]).pack().planes[0]
    x = torch.randn(m, n, device=DEV, generator=g) * 0.7 + 0.2
    x4 = x.view(1, m, 1, n).permute(0, 3, 1, 2)                                  # NHWC view: m rows of n channels
    gamma, beta = (torch.rand(n, device=DEV, generator=g) + 0.5), torch.randn(n, device=DEV, generator=g) * 0.3
    rm, rv, nbt = torch.zeros(n, device=DEV), torch.ones(n, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
    relu = mode != "no_relu"
    res = torch.randn(m, n, device=DEV, generator=g).view(1, m, 1, n).permute(0, 3, 1, 2) if mode == "relu_mask" else None
    y, save, ss, mask = capi.bn2d_fwd(x4, res, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, relu, want_mask=mode == "relu_mask")
    assert (mask is not None) == (mode == "relu_mask")
    link = [x4, save, ss, mask, relu]
    dy_s, part_s, ns_s = capi.gemm_x6s(a, pk, n, d, bn_bwd=link)
    dy_p, part_p, ns_p = capi.gemm_x6p(a, pk, n, d, bn_bwd=link)
    assert torch.equal(dy_s, dy_p) and part_s.shape == (2 * ns_s, n) and ns_s == capi.lib().peclr_gemm_x6s_waves(m)
    dy4 = dy_s.view(1, m, 1, n).permute(0, 3, 1, 2)
    outs = [capi.bn2d_bwd(dy4, x4, None, mask, save, ss, True, relu, False, pre=pre) for pre in ((part_s, ns_s), (part_p, ns_p), None)]
    for got in outs[:2]:
        for q in range(3):                                       # dx, dgamma, dbeta
            want = outs[2][q]
            assert float((got[q] - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6, (mode, q)
    for _ in range(4):                                           # fixed assignment of row blocks to waves: the same bits every launch
        again = capi.gemm_x6s(a, pk, n, d, bn_bwd=link)
        assert torch.equal(again[0], dy_s) and torch.equal(again[1], part_s)


def test_bottleneck_entry_gradient_routes_to_the_streaming_kernel_and_changes_nothing():
    """Through the module: two bottlenecks at a layer1-like shape; with `gemm_x6s` on the fused entry gradient runs on the
    streaming kernel (kernel family in the event log), and every output and gradient equals the `gemm_x6s=False` arm -- bit for
    bit for the activations' gradients, to round-off for what depends on the BatchNorm sums' order."""
    from peclr_amd import _capi
    from peclr_amd import bn2d as B
    from peclr_amd import resnet

    res, fam = {}, {}
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(16, 256, 56, 56, generator=g) * 0.7 + 0.3).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(16, 256, 56, 56, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    for on in (False, True):
        torch.manual_seed(9)
        net = torch.nn.Sequential(resnet.Bottleneck(256, 64, norm_layer=B.FusedBatchNormAct2d),
                                  resnet.Bottleneck(256, 64, norm_layer=B.FusedBatchNormAct2d))
        net = net.to(DEV).to(memory_format=torch.channels_last).train()
        B.enable_hip_batchnorm(net)
        _capi.EVENT_LOG = {}
        try:
            with B.routing(gemm_x6s=on, force=True):
                x = x0.clone().requires_grad_()
                y = net(x)
                y.backward(gy)
            torch.cuda.synchronize()
            fam[on] = sorted({e[4] for k, v in _capi.EVENT_LOG.items() if k.startswith("conv1x1_dgrad_add_x6") for e in v})
        finally:
            _capi.EVENT_LOG = None
        assert B.last_backward_leftovers == 0
        res[on] = (y.detach(), x.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
    assert fam[True] == ["gemm_x6s_kernel"] and fam[False] == ["gemm_x6p_kernel"], fam
    assert torch.equal(res[True][0], res[False][0])
    a, b = res[True][1], res[False][1]
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    for n in res[False][2]:
        a, b = res[True][2][n], res[False][2][n]
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-7, n
