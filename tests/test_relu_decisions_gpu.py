"""One ReLU decision per activation (torchvision Bottleneck / BasicBlock via /root/reference/src/models/resnet_model.py:15:
`conv -> bn -> relu`, autograd's backward uses the forward's own output): wherever the in-tree backward RECOMPUTES the
rectifier from the BatchNorm input instead of reading a mask -- `peclr_bn2d_bwd_*` MASK mode 1 (csrc/bn2d.hip), the
`bb_*` epilogues of the input-gradient GEMMs (csrc/gemm_x6p.hip, csrc/conv_h.hip) -- it evaluates the forward's own
expression, `fmaf(x, scale, shift) > 0`, on the forward's own rounded constants (the saved scale_shift table).  Held here
with EXACT integer arithmetic: with a gradient of ones, d(beta) of a channel is the number of its active elements, which
must equal the number of positive forward outputs -- also for inputs placed within a few representable numbers of the
threshold, where an expression that rounds differently (x * scale + shift, (x - mean) * invstd * gamma + beta) would decide
the other way.  VERDICT round 4, weak #1 / next #1.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _nhwc(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def _forward_decisions(x, y, ss):
    """The forward's rectifier decisions, two ways that must agree: the sign of fmaf(x, scale, shift) evaluated exactly (the
    product of two fp32 numbers is exact in float64 and the sum is correctly rounded, so its sign is the fused
    multiply-add's) and y > 0.  fp16 outputs only: a positive value under half the smallest subnormal (3e-8) is stored as
    0 -- the kernel's decision (and the backward's) is still "on"."""
    c = x.shape[1]
    pre = x.double() * ss[0].view(1, c, 1, 1).double() + ss[1].view(1, c, 1, 1).double()
    on = pre > 0
    differ = (y.float() > 0) != on
    if y.dtype == torch.float16:
        differ &= ~((y == 0) & on & (pre < 6e-8))
    assert not bool(differ.any()), "the forward output's sign pattern is not that of fmaf(x, scale, shift)"
    return on, pre


def _threshold_inputs(ss, shape, dtype, seed):
    """x [N, C, H, W] whose elements sit within +-6 representable `dtype` numbers of the root of fmaf(x, scale, shift) = 0 of
    their channel (half of them; the rest ordinary data)."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(seed)
    root = (-ss[1].double() / ss[0].double()).to(dtype)                                  # [C]
    ints = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float16: torch.int16}[dtype]
    base = root.view(ints).view(1, c, 1, 1).expand(n, c, h, w)
    k = torch.randint(-6, 7, shape, generator=g).to(DEV).to(ints)
    near = (base + k).contiguous().view(dtype)                                            # neighbours in the number line of `dtype`
    far = (torch.randn(shape, generator=g) * 0.7 + 0.3).to(DEV).to(dtype)
    pick = (torch.rand(shape, generator=g) < 0.5).to(DEV)
    return _nhwc(torch.where(pick, near, far))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c", [64, 256])
def test_batchnorm_backward_rectifies_exactly_where_the_forward_did_at_the_threshold(dtype, c):
    """Evaluation-mode BatchNorm (scale / shift do not depend on x, so inputs can be PLACED at the threshold): forward with
    ReLU, backward with the mask recomputed from x (no residual -> no bit mask exists), gradient of ones."""
    from peclr_amd import _capi as capi

    g = torch.Generator().manual_seed(c)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(c, generator=g) * 0.3).to(DEV)
    rm = (torch.randn(c, generator=g) * 0.2).to(DEV)
    rv = (torch.rand(c, generator=g) + 0.5).to(DEV)
    shape = (8, c, 24, 24)
    probe = _nhwc(torch.zeros(shape, dtype=dtype))
    _, save, ss, _ = capi.bn2d_fwd(probe, None, gamma, beta, rm, rv, None, False, 1e-5, 0.1, True)
    x = _threshold_inputs(ss, shape, dtype, seed=c + 1)
    y, save, ss, mask = capi.bn2d_fwd(x, None, gamma, beta, rm, rv, None, False, 1e-5, 0.1, True)
    assert mask is None
    on, pre = _forward_decisions(x, y, ss)
    close = (pre.abs() < 1e-5 if dtype == torch.float32 else pre.abs() < 0.05).sum()
    assert int(close) > x.numel() // 4, "the inputs are not at the threshold"
    assert 0.2 < float(on.float().mean()) < 0.8
    dy = _nhwc(torch.ones(shape, dtype=dtype))
    dx, dgamma, dbeta, _ = capi.bn2d_bwd(dy, x, None, None, save, ss, False, True, False)
    torch.cuda.synchronize()
    assert torch.equal(dbeta, on.sum(dim=(0, 2, 3)).float()), "the backward's recomputed ReLU decision differs from the forward's"
    assert torch.equal(dx.float() != 0, on)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_backward_counts_the_forwards_active_elements_in_training_mode(dtype):
    """Training mode at a routed size (32 768 rows): statistics of the batch, natural data; standalone reduction."""
    from peclr_amd import _capi as capi

    c, shape = 128, (32, 128, 32, 32)
    g = torch.Generator().manual_seed(7)
    x = _nhwc((torch.randn(shape, generator=g) * 0.7 + 0.3).to(dtype))
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.3).to(DEV)
    rm, rv, nbt = torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
    y, save, ss, _ = capi.bn2d_fwd(x, None, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, True)
    on, _ = _forward_decisions(x, y, ss)
    dy = _nhwc(torch.ones(shape, dtype=dtype))
    _, _, dbeta, _ = capi.bn2d_bwd(dy, x, None, None, save, ss, True, True, False)
    assert torch.equal(dbeta, on.sum(dim=(0, 2, 3)).float())


def _identity_planes(c, dtype):
    from peclr_amd import _capi as capi

    eye = torch.eye(c, device=DEV)
    if dtype == torch.float32:
        return capi.X6Planes([(eye, True)]).pack().planes[0], eye
    return capi.HPlanes([(eye, True)], dtype).pack().planes[0], eye


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("placed", [False, True])
def test_gemm_epilogue_reduction_rectifies_exactly_where_the_forward_did(dtype, placed):
    """The BatchNorm backward reduction inside an input-gradient GEMM's epilogue (no bit mask: the layer has no residual).
    The GEMM multiplies a gradient of ones by the identity, so the gradient arriving at the layer is exactly ones and the
    epilogue's sum of (dy * on) per channel is a count.  placed=True: evaluation-mode constants and inputs at the threshold."""
    from peclr_amd import _capi as capi

    c, shape = 128, (32, 128, 32, 32)
    rows = shape[0] * shape[2] * shape[3]
    g = torch.Generator().manual_seed(23)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.3).to(DEV)
    rm, rv = (torch.randn(c, generator=g) * 0.2).to(DEV), (torch.rand(c, generator=g) + 0.5).to(DEV)
    nbt = torch.zeros((), device=DEV, dtype=torch.int64)
    if placed:
        _, _, ss0, _ = capi.bn2d_fwd(_nhwc(torch.zeros(shape, dtype=dtype)), None, gamma, beta, rm, rv, None, False, 1e-5, 0.1, True)
        x = _threshold_inputs(ss0, shape, dtype, seed=29)
        y, save, ss, _ = capi.bn2d_fwd(x, None, gamma, beta, rm, rv, None, False, 1e-5, 0.1, True)
    else:
        x = _nhwc((torch.randn(shape, generator=g) * 0.7 + 0.3).to(dtype))
        y, save, ss, _ = capi.bn2d_fwd(x, None, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, True)
    on, _ = _forward_decisions(x, y, ss)
    planes, _ = _identity_planes(c, dtype)
    gy = torch.ones(rows, c, device=DEV, dtype=dtype)
    fn = capi.gemm_x6p if dtype == torch.float32 else capi.gemm_h
    dy, partial, ns = fn(gy, planes, c, bn_bwd=[x, save, ss, None, True])
    assert torch.equal(dy, gy)                                  # 1 * I: exact in both arithmetics
    # partial: peclr_bn2d_bwd_reduce's layout [2 * n_split, C]; the sums of (dy * on) are the rows the finalize adds up for d(beta)
    dy4 = dy.view(shape[0], shape[2], shape[3], c).permute(0, 3, 1, 2)
    _, _, dbeta, _ = capi.bn2d_bwd(dy4, x, None, None, save, ss, not placed, True, False, pre=(partial, ns))
    assert torch.equal(dbeta, on.sum(dim=(0, 2, 3)).float()), "the GEMM epilogue's ReLU decision differs from the forward's"
