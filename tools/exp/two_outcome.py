"""Root-cause probe for the two-outcome BasicBlock test.

In ONE process: run the in-tree arm of `test_basicblock_input_with_two_consumers...` first, then repeatedly
[predecessor tests -> arm again], and compare every forward intermediate, output, gradient and running statistic of each
arm with the first one bit for bit.  Prints the first tensor that differs (in forward order, then backward results).

  python tools/exp/two_outcome.py [rounds] [--poison] [--hooks]
    --poison   torch.empty returns NaN-filled memory (read-before-write shows up as NaN / a changed result)
    --hooks    also record the gradient arriving at every module output (tensor hooks; may perturb in-place accumulation)
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
DEV = "cuda:0"


def digest(t):
    t = t.detach()
    if t.dim() == 4 and any(s == 0 for s in t.stride()):
        return "lazy-view"
    return hashlib.sha1(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]


def arm(hooks=False, keep=None):
    from peclr_amd import _capi
    from peclr_amd import bn2d as B
    from peclr_amd import resnet

    torch.manual_seed(11)
    ds = torch.nn.Sequential(resnet.conv1x1(64, 128, 2), B.FusedBatchNormAct2d(128))
    net = torch.nn.Sequential(resnet.BasicBlock(64, 64, norm_layer=B.FusedBatchNormAct2d),
                              resnet.BasicBlock(64, 64, norm_layer=B.FusedBatchNormAct2d),
                              resnet.BasicBlock(64, 128, 2, ds, norm_layer=B.FusedBatchNormAct2d))
    net = net.to(DEV).to(memory_format=torch.channels_last).train()
    B.enable_hip_batchnorm(net)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(32, 64, 32, 32, generator=g) * 0.7 + 0.3).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(32, 128, 16, 16, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    rec = {}
    names = {m: n for n, m in net.named_modules()}

    def fwd_hook(mod, inp, out):
        if isinstance(out, torch.Tensor):
            n = names[mod]
            rec[f"fwd:{n}"] = digest(out)
            if keep is not None:
                keep[f"fwd:{n}"] = out.detach().clone()
            if hooks and out.requires_grad:
                def gh(grad, n=n):
                    rec[f"bwd:{n}"] = digest(grad)
                    if keep is not None and not any(s == 0 for s in grad.stride()):
                        keep[f"bwd:{n}"] = grad.detach().clone()
                out.register_hook(gh)

    hs = [m.register_forward_hook(fwd_hook) for m in net.modules() if not list(m.children())]
    x = x.clone().requires_grad_()
    _capi.EVENT_LOG = {}
    _capi.LAUNCH_ORDER = []
    try:
        with B.routing(force=True):
            y = net(x)
            y.backward(gy)
        torch.cuda.synchronize()
        order = list(_capi.LAUNCH_ORDER)
    finally:
        _capi.EVENT_LOG = None
        _capi.LAUNCH_ORDER = None
    B.end_backward()
    for h in hs:
        h.remove()
    rec["out:y"] = digest(y)
    rec["out:dx"] = digest(x.grad)
    for n, p in net.named_parameters():
        if p.grad is not None:
            rec[f"grad:{n}"] = digest(p.grad)
            if keep is not None:
                keep[f"grad:{n}"] = p.grad.detach().clone()
    for n, b in net.named_buffers():
        if b.dtype.is_floating_point:
            rec[f"buf:{n}"] = digest(b)
    if keep is not None:
        keep["out:dx"] = x.grad.detach().clone()
    return rec, order


def predecessors():
    import pytest

    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return pytest.main([os.path.join(here, "tests/test_hip_parity.py"), "-x", "-q", "-m", "gpu", "-k", "x6_tn or x6t_stride",
                        "-p", "no:cacheprovider", "--no-header", "-q"])


def miopen_probe():
    """The suspect: MIOpen's fp32 input gradient of the 1x1 / stride-2 shortcut (64 -> 128 channels, 32 x 32 x 32 pixels)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(32, 64, 32, 32, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(32, 128, 16, 16, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 64, 1, 1, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last)
    run = lambda a, b, c: torch.ops.aten.convolution_backward(a, b, c, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]  # noqa: E731
    ref = run(gy.double(), x.double(), w.double())
    outs = {}
    worst = 0.0
    for i in range(40):
        junk = torch.randn(1 + 997 * i, 33, device=DEV)
        d = run(gy, x, w)
        torch.cuda.synchronize()
        del junk
        e = float((d.double() - ref).norm() / ref.norm())
        worst = max(worst, e)
        outs[digest(d)] = outs.get(digest(d), 0) + 1
    print(f"MIOpen 1x1/s2 dgrad, 40 launches: {len(outs)} distinct results, worst norm-wise error vs float64 {worst:.3e}", flush=True)


def main():
    if "--miopen" in sys.argv:
        return miopen_probe()
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
    poison = "--poison" in sys.argv
    hooks = "--hooks" in sys.argv
    if poison:
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    keep0 = {}
    first, order0 = arm(hooks, keep0)
    print(f"arm 0: {len(first)} tensors, {len(order0)} launches", flush=True)
    nan_keys = [k for k, v in keep0.items() if not torch.isfinite(v).all()]
    print("non-finite in arm 0:", nan_keys[:10], flush=True)
    outcomes = {}
    for r in range(rounds):
        if "--nopred" not in sys.argv:
            predecessors()
        keep = {}
        rec, order = arm(hooks, keep)
        diff = [k for k in first if rec.get(k) != first[k]]
        sig = hashlib.sha1(repr(sorted(rec.items())).encode()).hexdigest()[:8]
        outcomes[sig] = outcomes.get(sig, 0) + 1
        if order != order0:
            print(f"round {r}: LAUNCH ORDER differs: {[(i, a, b) for i, (a, b) in enumerate(zip(order0, order)) if a != b][:5]}", flush=True)
        if diff:
            print(f"round {r}: {len(diff)} tensors differ; first in recording order: {diff[:6]}", flush=True)
            for k in diff[:12]:
                if k in keep and k in keep0:
                    a, b = keep0[k].float(), keep[k].float()
                    d = (a - b).abs()
                    nz = int((d > 0).sum())
                    at = torch.nonzero(d > 0)[:4].tolist()
                    print(f"   {k}: {nz} elements differ of {d.numel()}, max |d| {float(d.max()):.3e} (scale {float(a.abs().max()):.3e}) at {at}", flush=True)
        else:
            print(f"round {r}: identical", flush=True)
    print("outcomes:", outcomes, flush=True)


if __name__ == "__main__":
    main()
