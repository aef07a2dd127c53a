"""Round-4 parity hardening of the in-tree fp32 backbone (all `-m gpu`):

  * whole Bottleneck / BasicBlock NETWORKS on the in-tree kernels against the SAME weights on stock ops in float64 (the
    round-3 checks of the compact shortcut gradient, the (dY, mask) hand-over, the parity-class input gradient and the
    BatchNorm reductions fused into GEMM epilogues were arm-vs-arm, both arms in-tree);
  * the two silent-wrong-gradient scenarios the advisor found (a BatchNorm output with two consumers; a hipGraph captured
    right after a pass without optimiser step);
  * bit-repeatability of every kernel that synchronises with hand-counted waits and raw barriers, at ResNet-50's full shapes.
Reference semantics: one Lightning step on one device, /root/reference/src/models/unsupervised/hybrid2_model.py:27-90 over
torchvision's ResNet (resnet_model.py:15).
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _nhwc(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def _f64_copy(net):
    """The same module tree on stock ops (`hip=False`: F.batch_norm / add / relu, torch's convolutions) in float64."""
    from peclr_amd import bn2d as B

    ref = copy.deepcopy(net)
    B.enable_hip_batchnorm(ref, False)
    return ref.double()


def _run(net, x, gy):
    x = x.clone().requires_grad_()
    y = net(x)
    y.backward(gy.to(y.dtype))
    torch.cuda.synchronize()
    return y.detach(), x.grad.detach(), {n: p.grad.detach() for n, p in net.named_parameters() if p.grad is not None}, \
        {n: b.detach().clone() for n, b in net.named_buffers() if b.dtype.is_floating_point}


def _rel(a, b):
    return float((a.double() - b).norm() / (b.norm() + 1e-30))


def _errors(got, want):
    (y, dx, gp, bufs), (yr, dxr, gpr, bufr) = got, want
    assert set(gp) == set(gpr)
    e = {"y": float((y.double() - yr).abs().max() / yr.abs().max()), "dx": _rel(dx, dxr),
         "grad": max(_rel(gp[n], gpr[n]) for n in gpr),
         "stat": max(float((bufs[n].double() - bufr[n]).abs().max() / (bufr[n].abs().max() + 1e-30)) for n in bufr)}
    e["grad_at"] = max(gpr, key=lambda n: _rel(gp[n], gpr[n]))
    return e


def _check(got, want, stock, floors, what):
    """In-tree fp32 against float64, CALIBRATED by stock fp32 ops (MIOpen + ATen) against the same float64 run: train-mode
    BatchNorm over few rows amplifies fp32 round-off (and a ReLU decision within round-off of zero flips a whole gradient
    element) by an amount that depends on depth and batch, so the bar is 6 x what the stock fp32 path shows on the same
    data, with a floor.  A dropped term -- a shortcut gradient missing from a reduction, a parity class, stale weight
    planes -- is O(0.1 ... 1) and two to three orders above either."""
    mine, theirs = _errors(got, want), _errors(stock, want)
    for k, floor in floors.items():
        assert mine[k] <= max(6 * theirs[k], floor), (what, k, mine, theirs)
    return mine, theirs


def _f32_stock_copy(net):
    from peclr_amd import bn2d as B

    ref = copy.deepcopy(net)
    B.enable_hip_batchnorm(ref, False)
    return ref


def _rectifier_margins(ref64, x64):
    """Run the float64 stock network on x64 and return {module name: pre-activation tensor} for every fused BatchNorm that is
    followed by a ReLU (the value the rectifier decides on: bn(x) (+ residual)), recomputed from the module's inputs."""
    from peclr_amd import bn2d as B

    names = {m: n for n, m in ref64.named_modules()}
    pre, hooks = {}, []

    def hook(mod, args, out):
        xin = args[0]
        res = args[1] if len(args) > 1 else None
        relu = args[2] if len(args) > 2 else mod.default_relu
        if relu:
            a = torch.nn.functional.batch_norm(xin, None, None, mod.weight, mod.bias, True, 0.0, mod.eps)
            pre[names[mod]] = (a + res if res is not None else a).detach()

    for m in ref64.modules():
        if isinstance(m, B.FusedBatchNormAct2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        ref64(x64)
    for h in hooks:
        h.remove()
    return pre


def _in_tree_activations(net, x):
    """{module name: post-ReLU output} of every fused BatchNorm of the in-tree network (forward only, routing forced)."""
    from peclr_amd import bn2d as B

    names = {m: n for n, m in net.named_modules()}
    outs, hooks = {}, []
    for m in net.modules():
        if isinstance(m, B.FusedBatchNormAct2d):
            hooks.append(m.register_forward_hook(lambda mod, args, out: outs.__setitem__(names[mod], out.detach())))
    with torch.no_grad(), B.routing(force=True):
        net(x)
    for h in hooks:
        h.remove()
    return outs


def _rectifier_ties(pre64, act32, tol=1e-5):
    """(decisions that differ from float64's although the float64 pre-activation is further than tol * scale from zero,
    decisions that differ within it, elements within it) summed over the layers."""
    wrong = ties = near = 0
    for n, a in pre64.items():
        if act32[n].shape != a.shape:        # stem (BatchNorm + ReLU + max-pool in one pass) / tail (+ average pool): no activation to compare
            continue
        scale = float(a.abs().max())
        differ = (act32[n].double() > 0) != (a > 0)
        close = a.abs() <= tol * scale
        wrong += int((differ & ~close).sum())
        ties += int((differ & close).sum())
        near += int(close.sum())
    return wrong, ties, near


def _basicblock_arm():
    """Three BasicBlocks at routed sizes (identity, identity, stride 2 + downsample; 32 768 rows), fresh from fixed seeds."""
    from peclr_amd import bn2d as B
    from peclr_amd import resnet

    torch.manual_seed(11)
    ds = torch.nn.Sequential(resnet.conv1x1(64, 128, 2), B.FusedBatchNormAct2d(128))
    net = torch.nn.Sequential(resnet.BasicBlock(64, 64, norm_layer=B.FusedBatchNormAct2d),
                              resnet.BasicBlock(64, 64, norm_layer=B.FusedBatchNormAct2d),
                              resnet.BasicBlock(64, 128, 2, ds, norm_layer=B.FusedBatchNormAct2d))
    net = net.to(DEV).to(memory_format=torch.channels_last).train()
    g = torch.Generator().manual_seed(3)
    x = _nhwc(torch.randn(32, 64, 32, 32, generator=g) * 0.7 + 0.3)
    gy = _nhwc(torch.randn(32, 128, 16, 16, generator=g))
    return net, x, gy


def test_basicblock_input_with_two_consumers_gets_the_whole_batchnorm_reduction():
    """ResNet-18/34 topology: a block's input feeds conv1 (3x3) AND the shortcut.  conv1's input-gradient GEMM used to
    register its BatchNorm backward sums for the buffer autograd then adds the shortcut's gradient INTO: the previous
    block's bn2 popped sums that missed the shortcut term (advisor, round 3).  fp32 in-tree against float64 stock ops."""
    from peclr_amd import _capi
    from peclr_amd import bn2d as B

    net, x, gy = _basicblock_arm()
    ref, stock = _f64_copy(net), _f32_stock_copy(net)
    B.enable_hip_batchnorm(net)
    _capi.EVENT_LOG = {}
    try:
        with B.routing(force=True):
            got = _run(net, x, gy)
        tags = {k: len(v) for k, v in _capi.EVENT_LOG.items()}
    finally:
        _capi.EVENT_LOG = None
    assert tags.get("conv3x3_dgrad", 0) >= 4 and tags.get("conv3x3_fwd", 0) >= 5, tags     # the in-tree kernels did run
    assert tags.get("conv_s2_dgrad", 0) >= 1, tags      # the shortcut's input gradient too (not MIOpen's atomically added one)
    assert B.last_backward_leftovers == 0 and B.end_backward() == 0
    want = _run(ref, x.double(), gy.double())
    # every rectifier decision of the in-tree forward is float64's, except where float64's own pre-activation is within
    # 1e-5 (relative) of zero -- a handful of the 1.1e7 decisions, each of which moves one element's gradient (2e-4 .. 5e-4
    # norm-wise here, for ANY fp32 evaluation: the stock arm has its own)
    fresh, _, _ = _basicblock_arm()
    B.enable_hip_batchnorm(fresh)
    wrong, ties, near = _rectifier_ties(_rectifier_margins(_f64_copy(fresh), x.double()), _in_tree_activations(fresh, x))
    print(f"rectifier decisions that differ from float64's: {wrong} away from zero, {ties} of the {near} within 1e-5 of it")
    assert wrong == 0 and ties <= 8
    # sums that miss the shortcut's contribution -- what this test is for -- are off by tens of per cent
    print(_check(got, want, _run(stock, x, gy), {"y": 2e-5, "dx": 1e-4, "grad": 2e-4, "stat": 1e-5}, "BasicBlock x 3"))


def test_in_tree_basicblock_arm_gives_the_same_bits_every_time():
    """Round 4 saw a second outcome of the test above (dx 2.3e-4, 1.bn2.bias 4.9e-4 against float64) about one run in ten
    behind other tests.  Root cause (tools/exp/two_outcome.py: every forward tensor equal, the first difference is the
    gradient arriving at block 1's bn2, and none under torch's deterministic mode): the ONE launch of the arm that was not
    in-tree -- MIOpen's fp32 input gradient of the 1x1 / stride-2 shortcut -- accumulates with float atomics.  That gradient
    is now the in-tree GEMM over the output pixels scattered into zeros, so the arm consists of bit-repeatable kernels only:
    fresh networks from the same seeds give the same bits, with other kernels and other allocator states in between."""
    from peclr_amd import _capi as capi
    from peclr_amd import bn2d as B

    def arm():
        net, x, gy = _basicblock_arm()
        B.enable_hip_batchnorm(net)
        with B.routing(force=True):
            out = _run(net, x, gy)
        assert B.last_backward_leftovers == 0 and B.end_backward() == 0
        return out

    first = arm()
    for i in range(5):
        junk = [torch.randn(257 * (i + 1), 64 * (j + 1), device=DEV) for j in range(3)]       # another allocator state,
        capi.gemm_x6t(torch.randn(4096, 128, device=DEV), torch.randn(4096, 256, device=DEV))  # other kernels in between
        del junk
        y, dx, gp, bufs = arm()
        assert torch.equal(y, first[0]) and torch.equal(dx, first[1]), i
        for n in first[2]:
            assert torch.equal(gp[n], first[2][n]), (i, n)
        for n in first[3]:
            assert torch.equal(bufs[n], first[3][n]), (i, n)


def _small_basicblock_arm(seed):
    from peclr_amd import bn2d as B
    from peclr_amd import resnet

    torch.manual_seed(seed)
    ds = torch.nn.Sequential(resnet.conv1x1(64, 128, 2), B.FusedBatchNormAct2d(128))
    net = torch.nn.Sequential(resnet.BasicBlock(64, 64, norm_layer=B.FusedBatchNormAct2d),
                              resnet.BasicBlock(64, 128, 2, ds, norm_layer=B.FusedBatchNormAct2d))
    net = net.to(DEV).to(memory_format=torch.channels_last).train()
    g = torch.Generator().manual_seed(seed + 1)
    x = _nhwc(torch.randn(4, 64, 16, 16, generator=g) * 0.7 + 0.3)
    gy = _nhwc(torch.randn(4, 128, 8, 8, generator=g))
    return net, x, gy


TIE_FREE_SEED = 12      # chosen with tools/exp/tie_free_seed.py: no float64 pre-activation within 1e-5 of zero


def test_two_consumer_blocks_equal_float64_where_no_rectifier_decision_is_a_tie():
    """The same composition (block input with two consumers, stride-2 shortcut scattered in-tree, BatchNorm reductions in the
    GEMM epilogues) on data whose float64 pre-activations all lie further than 5e-6 from zero (fp32 round-off there: < 1e-6) -- asserted here --, so
    that every rectifier decision of the fp32 arm IS float64's (asserted too) and the comparison needs no calibration by the
    stock fp32 path and no allowance for flipped decisions: absolute bars, an order of magnitude under the ones above."""
    from peclr_amd import bn2d as B

    net, x, gy = _small_basicblock_arm(TIE_FREE_SEED)
    ref = _f64_copy(net)
    B.enable_hip_batchnorm(net)
    pre = _rectifier_margins(copy.deepcopy(ref), x.double())     # (a copy: a training-mode forward moves the running statistics)
    assert len(pre) == 4
    margin = min(float(a.abs().min()) for a in pre.values())     # (BatchNorm outputs: unit scale)
    assert margin > 5e-6, f"test data: a float64 pre-activation lies {margin:.1e} from zero -- pick another TIE_FREE_SEED"
    act = _in_tree_activations(copy.deepcopy(net), x)       # (a copy: this forward moves the running statistics)
    assert _rectifier_ties(pre, act)[:2] == (0, 0)
    with B.routing(force=True):
        got = _run(net, x, gy)
    assert B.last_backward_leftovers == 0 and B.end_backward() == 0
    e = _errors(got, _run(ref, x.double(), gy.double()))
    print(e)
    assert e["y"] <= 5e-6 and e["dx"] <= 5e-6 and e["grad"] <= 1e-5 and e["stat"] <= 2e-6, e


@pytest.mark.parametrize("arch,n,size", [("resnet50", 8, 224), ("resnet18", 16, 128)])
def test_whole_network_in_tree_equals_float64_stock(arch, n, size):
    """The whole encoder with EVERY in-tree kernel routed (`routing(force=True)`: the composition of the full-size step --
    compact stride-2 shortcut gradient + s2add, (dY, mask) hand-over + maskadd, parity-class 3x3 / stride-2 input
    gradient, statistics and backward reductions in the GEMM epilogues, lazy NaN views) against the same weights on stock
    ops in float64: features, input gradient, every parameter gradient (norm-wise), running statistics."""
    from peclr_amd import _capi
    from peclr_amd import bn2d as B
    from peclr_amd.config import Config
    from peclr_amd.encoder import get_wrapper_model

    torch.manual_seed(5)
    net = get_wrapper_model(Config({"resnet_size": arch[len("resnet"):]}), False).to(DEV).to(memory_format=torch.channels_last).train()
    for p in net.final_layer.parameters():          # never receives a gradient (resnet_model.py:27-29)
        p.requires_grad_(False)
    ref, stock = _f64_copy(net), _f32_stock_copy(net)
    B.enable_hip_batchnorm(net)
    g = torch.Generator().manual_seed(size)
    x = _nhwc(torch.randn(n, 3, size, size, generator=g))
    din = 2048 if arch == "resnet50" else 512
    gy = (torch.randn(n, din, generator=g) / din).to(DEV)
    _capi.EVENT_LOG = {}
    try:
        with B.routing(force=True):
            got = _run(net, x, gy)
            left = B.last_backward_leftovers + B.end_backward()    # (the engine drains the side channels when the backward pass ends)
        tags = {k: len(v) for k, v in _capi.EVENT_LOG.items()}
    finally:
        _capi.EVENT_LOG = None
    assert left == 0, f"{left} gradient hand-overs were never consumed"
    if arch == "resnet50":
        for t in ("conv1x1_fwd", "conv1x1_dgrad", "conv1x1_dgrad_add_x6", "conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad",
                  "conv1x1_wgrad", "conv_s2_fwd", "conv_s2_dgrad", "conv3x3_s2_dgrad"):
            assert tags.get(t, 0) > 0, (t, tags)
        assert tags.get("bn2d_bwd_reduce", 0) <= 6, tags
    fresh = copy.deepcopy(ref).float()
    B.enable_hip_batchnorm(fresh)
    wrong, ties, near = _rectifier_ties(_rectifier_margins(copy.deepcopy(ref), x.double()), _in_tree_activations(fresh, x), tol=1e-4)
    print(f"{arch}: rectifier decisions that differ from float64's: {wrong} away from zero, {ties} of the {near} within 1e-4 of it")
    want = _run(ref, x.double(), gy.double())
    # fp32 round-off through 53 (20) train-mode BatchNorm layers over few rows; a dropped term (shortcut gradient, one parity
    # class, a stale reduction) is O(0.1 - 1) in the gradients of the layers below it
    # (floors of 2e-2 on the gradients: ONE ReLU decision within round-off of zero that falls the other way than in float64 drops
    # or adds one element's gradient -- 3e-3 ... 9e-3 norm-wise at these sizes, whichever fp32 path runs: tools/exp/rn18_err_ab.py)
    worst = _check(got, want, _run(stock, x, gy), {"y": 2e-5, "dx": 2e-2, "grad": 2e-2, "stat": 1e-5}, arch)
    print(f"{arch}: in-tree {worst[0]} | stock fp32 {worst[1]}")


def test_c5_layer1_shape_w112_routes_in_tree_and_matches_float64():
    """C5's layer1 (112 x 112 pixels per image: 2 x 64 views @448): wider rows than the halo-patch 3x3 variant holds in LDS
    at 256-row tiles -- whatever variant the library picks there, two chained bottlenecks against float64 stock ops."""
    from peclr_amd import _capi
    from peclr_amd import bn2d as B
    from peclr_amd import resnet

    torch.manual_seed(13)
    net = torch.nn.Sequential(resnet.Bottleneck(256, 64, norm_layer=B.FusedBatchNormAct2d),
                              resnet.Bottleneck(256, 64, norm_layer=B.FusedBatchNormAct2d))
    net = net.to(DEV).to(memory_format=torch.channels_last).train()
    ref, stock = _f64_copy(net), _f32_stock_copy(net)
    B.enable_hip_batchnorm(net)
    g = torch.Generator().manual_seed(112)
    x = _nhwc(torch.randn(6, 256, 112, 112, generator=g) * 0.7 + 0.3)
    gy = _nhwc(torch.randn(6, 256, 112, 112, generator=g))
    _capi.EVENT_LOG = {}
    try:
        with B.routing(force=True):
            got = _run(net, x, gy)
        tags = {k.split("~")[0] for k in _capi.EVENT_LOG}
    finally:
        _capi.EVENT_LOG = None
    assert {"conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "conv1x1_dgrad_add_x6"} <= tags, tags
    want = _run(ref, x.double(), gy.double())
    print(_check(got, want, _run(stock, x, gy), {"y": 2e-5, "dx": 1e-4, "grad": 2e-4, "stat": 1e-5}, "Bottleneck x 2 @112"))


def test_graph_captured_after_a_pass_without_optimiser_step_repacks_the_weight_planes():
    """Advisor, round 3: with `accumulate_grad_batches > 1` the eager micro-step before the capture takes no optimiser step,
    so the pack group's stamps are fresh at capture time; the first member (a shape that is not routed) never asks for
    planes, and no pack launch used to be recorded -- replays then ran on planes split from stale weights.  Capture a
    forward + backward right after an eager pass, change the weights in place, replay, compare with eager."""
    from peclr_amd import _capi
    from peclr_amd import bn2d as B
    from peclr_amd import resnet

    torch.manual_seed(17)
    net = torch.nn.Sequential(resnet.Bottleneck(512, 128, norm_layer=B.FusedBatchNormAct2d),
                              resnet.Bottleneck(512, 128, norm_layer=B.FusedBatchNormAct2d))
    net = net.to(DEV).to(memory_format=torch.channels_last).train()
    B.enable_hip_batchnorm(net)
    group = net[0].conv1.x6_group
    assert group is not None and group.convs[0] is net[0].conv1
    g = torch.Generator().manual_seed(2)
    x = _nhwc(torch.randn(16, 512, 28, 28, generator=g) * 0.7 + 0.3).requires_grad_()
    gy = _nhwc(torch.randn(16, 512, 28, 28, generator=g))
    rows = 16 * 28 * 28
    assert not B._x6_pays(rows, 128, 512) and B._x6_pays(rows, 512, 128)   # member 0 does not ask for planes, later ones do

    def step():
        for p in net.parameters():
            p.grad = None
        x.grad = None
        y = net(x)
        y.backward(gy)
        return y

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                   # eager pass, NO optimiser step: the group's stamps are fresh
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        _capi.EVENT_LOG = {}
        try:
            with torch.cuda.graph(graph, stream=side):
                y_static = step()
            packs = len(_capi.EVENT_LOG.get("x6_pack", []))
        finally:
            _capi.EVENT_LOG = None
        assert packs == 1, f"{packs} pack launches recorded in the capture"
        with torch.no_grad():                    # what an optimiser step does: new values, same storage
            for p in net.parameters():
                if p.dim() == 4:
                    p.mul_(1.25)
        _capi.WEIGHTS_EPOCH += 1
        graph.replay()
        torch.cuda.synchronize()
        y_replay, dx_replay = y_static.clone(), x.grad.clone()
        gw_replay = [p.grad.clone() for p in net.parameters()]
        y_eager = step().detach().clone()
        torch.cuda.synchronize()
    # (the running means moved once more between the two passes; they are the centre the GEMM epilogues subtract before
    # summing the statistics, so the two passes round differently in the last bits -- stale planes would be off by 25 %)
    close = lambda a, b: float((a.detach() - b.detach()).norm()) <= 5e-3 * float(b.detach().norm()) + 1e-7   # noqa: E731
    assert close(y_replay, y_eager) and close(dx_replay, x.grad)
    for a, p in zip(gw_replay, net.parameters()):
        assert close(a, p.grad)


@pytest.mark.parametrize("rows,k,n", [(256 * 28 * 28, 512, 128), (256 * 14 * 14, 256, 1024), (256 * 56 * 56, 64, 256),
                                      (256 * 7 * 7, 2048, 512)])
def test_1x1_kernels_repeat_themselves_bit_for_bit(rows, k, n):
    """gemm_x6p (forward shape, with statistics; input-gradient shape with a dense addend) and gemm_x6t (weight gradient) at
    ResNet-50's full 1x1 shapes: forty / twelve launches, every result equal to the first bit for bit -- the guard for
    hand-counted vmcnt / lgkmcnt waits and raw s_barriers (a missing wait shows as one wrong tile in a few thousand)."""
    from peclr_amd import _capi as capi

    g = torch.Generator(device=DEV).manual_seed(rows % 1000 + k + n)
    a = torch.randn(rows, k, device=DEV, generator=g)
    w = torch.randn(n, k, device=DEV, generator=g) * 0.05
    pk = capi.X6Planes([(w, False), (w, True)]).pack()
    shift = torch.randn(n, device=DEV, generator=g) * 0.1
    y0, p0, _ = capi.gemm_x6p(a, pk.planes[0], n, stat_shift=shift)
    for _ in range(40):
        y, p, _ = capi.gemm_x6p(a, pk.planes[0], n, stat_shift=shift)
        assert torch.equal(y, y0) and torch.equal(p, p0)
    add = torch.randn(rows, k, device=DEV, generator=g)
    d0 = capi.gemm_x6p(y0, pk.planes[1], k, add)
    for _ in range(20):
        assert torch.equal(capi.gemm_x6p(y0, pk.planes[1], k, add), d0)
    w0 = capi.gemm_x6t(y0, a)
    for _ in range(12):
        assert torch.equal(capi.gemm_x6t(y0, a), w0)


@pytest.mark.parametrize("nb,cin,cout,hw", [(256, 128, 128, 28), (256, 256, 256, 14), (256, 64, 64, 56)])
def test_3x3_weight_gradient_and_stride_2_kernels_repeat_themselves_bit_for_bit(nb, cin, cout, hw):
    """gemm_x6w (3x3 weight gradient, nine taps per workgroup), the stride-2 forward (3x3 and 1x1) and its weight gradients
    at ResNet-50's full shapes, launched repeatedly: bit-identical results."""
    from peclr_amd import _capi as capi

    g = torch.Generator(device=DEV).manual_seed(nb + cin + hw)
    x = torch.randn(nb, cin, hw, hw, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(nb, cout, hw, hw, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    x2, gy2 = x.permute(0, 2, 3, 1).reshape(-1, cin), gy.permute(0, 2, 3, 1).reshape(-1, cout)
    w0 = capi.gemm_x6t(gy2, x2, taps=9, hw=(hw, hw))
    for _ in range(12):
        assert torch.equal(capi.gemm_x6t(gy2, x2, taps=9, hw=(hw, hw)), w0)
    w3 = (torch.randn(cout, cin, 3, 3, device=DEV, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    w1 = torch.randn(cout, cin, device=DEV, generator=g) * 0.05
    pk = capi.X6Planes([(w3.permute(0, 2, 3, 1).reshape(cout, 9 * cin), False), (w1, False)]).pack()
    shift = torch.zeros(cout, device=DEV)
    for taps, planes in ((9, pk.planes[0]), (1, pk.planes[1])):
        y0, p0, _ = capi.conv_s2_x6p(x, planes, cout, taps, stat_shift=shift)
        for _ in range(12):
            y, p, _ = capi.conv_s2_x6p(x, planes, cout, taps, stat_shift=shift)
            assert torch.equal(y, y0) and torch.equal(p, p0), taps
        gys = y0
        ys2 = gys.permute(0, 2, 3, 1).reshape(-1, cout)
        d0 = capi.gemm_x6t(ys2, x2, taps=taps, hw=(hw // 2, hw // 2), stride=2)
        for _ in range(6):
            assert torch.equal(capi.gemm_x6t(ys2, x2, taps=taps, hw=(hw // 2, hw // 2), stride=2), d0), taps
