"""Round 6, gated side experiment of the review's item 1: would TWO fp16 planes (hi, lo) with an exact power-of-two per-tensor
scale and THREE products (hh, hl, lh) carry the fp32 GEMMs at the six-product kernels' accuracy?  (4 bytes per element like fp32,
half the matrix-core work.)  CPU emulation on REAL layer tensors: a ResNet-50 forward + backward (random init, train-mode
BatchNorm, 2 x 4 views @224) gives every 1x1 convolution's input X, weight W and output gradient dY; each of its three products
(forward X W^T, input gradient dY W, weight gradient dY^T X) is evaluated
  ref  : float64
  x6   : exact three-way bf16 split of both operands, six products, each an fp32 matmul, added smallest first in fp32
  f16x3: hi = fp16(s x), lo = fp16(s x - hi), s = 2^k per tensor with max |s x| in [2^14, 2^15); three fp32 matmuls; unscaled
and compared by max |err| / max |ref| (bench.py's fp32_gemm_check metric) and, per output element, |err| / sum_k |a_k b_k|
(the component-wise bound an fp32 dot product satisfies).   python tools/exp/fp16_pair_probe.py"""
import math
import sys

import torch

sys.path.insert(0, ".")
from peclr_amd.resnet import resnet50  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(16)


def bf16_split(x):
    h = x.bfloat16().float()
    r = x - h
    m = r.bfloat16().float()
    return h, m, r - m


def mm_x6(a, b):                     # a [M, K], b [K, N]
    ah, am, al = bf16_split(a)
    bh, bm, bl = bf16_split(b)
    acc = al @ bh
    for p, q in ((ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
        acc = acc + p @ q
    return acc


def f16_pair(x):
    mx = float(x.abs().max())
    k = 14 - math.floor(math.log2(mx)) if mx > 0 else 0
    s = 2.0 ** k
    xs = x * s                                       # exact (power of two, no overflow / underflow in fp32 here)
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return hi.float(), lo.float(), s


def mm_f16x3(a, b):
    ah, al, sa = f16_pair(a)
    bh, bl, sb = f16_pair(b)
    acc = al @ bh
    acc = acc + ah @ bl
    acc = acc + ah @ bh
    return acc * (1.0 / (sa * sb))


def mm_f32_chain(a, b):
    """What a true fp32 kernel computes: a k-ordered chain of fp32 multiply-adds (v_mfma_f32 is bit-for-bit that, fused; here product
    and sum round separately -- the same error class).  Rows sampled to keep the loop short."""
    acc = torch.zeros(a.shape[0], b.shape[1])
    for k in range(a.shape[1]):
        acc = acc + a[:, k:k + 1] * b[k:k + 1, :]
    return acc


def errs(c, ref, bound):
    e = (c.double() - ref).abs()
    return float(e.max() / ref.abs().max()), float((e / bound.clamp_min(1e-300)).max())


model = resnet50().train()
taps = {}
for name, m in model.named_modules():
    if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1):
        def fwd_hook(mod, inp, out, name=name):
            taps[name] = [inp[0].detach(), mod.weight.detach(), None]
            out.register_hook(lambda g, name=name: taps[name].__setitem__(2, g.detach()))
        m.register_forward_hook(fwd_hook)
x = torch.randn(8, 3, 224, 224)
y = model(x)
(y.square().mean() + y.mean()).backward()

print(f"{'layer':28s} {'product':8s} {'x6 err/scale':>13s} {'f16x3 err/scale':>16s} {'x6 comp.':>10s} {'f16x3 comp.':>12s} {'fp32 blas':>10s} {'fp32 chain':>11s}   range of dY: max / median")
worst = {}
for name in [n for n in taps if n.endswith(("0.conv1", "0.conv3", "1.conv1", "1.conv3"))]:
    X, W, dY = taps[name]
    cin, cout = X.shape[1], W.shape[0]
    X2 = X.permute(0, 2, 3, 1).reshape(-1, cin)
    dY2 = dY.permute(0, 2, 3, 1).reshape(-1, cout)
    W2 = W.reshape(cout, cin)
    rng = f"{float(dY2.abs().max()):.2e} / {float(dY2.abs().median()):.2e}"
    for what, a, b in (("fwd", X2, W2.t().contiguous()), ("dgrad", dY2, W2), ("wgrad", dY2.t().contiguous(), X2)):
        ref = a.double() @ b.double()
        bound = a.double().abs() @ b.double().abs()
        e6, c6 = errs(mm_x6(a, b), ref, bound)
        e3, c3 = errs(mm_f16x3(a, b), ref, bound)
        eb, _ = errs(a @ b, ref, bound)
        if what == "wgrad":                  # (a long contraction: the chain over a sample of output rows)
            sel = slice(0, min(a.shape[0], 32))
            ec = float((mm_f32_chain(a[sel], b).double() - ref[sel]).abs().max() / ref.abs().max())
        else:
            sel = torch.arange(0, a.shape[0], max(1, a.shape[0] // 2048))
            ec = float((mm_f32_chain(a[sel], b).double() - ref[sel]).abs().max() / ref.abs().max())
        print(f"{name:28s} {what:8s} {e6:13.2e} {e3:16.2e} {c6:10.2e} {c3:12.2e} {eb:10.2e} {ec:11.2e}   {rng if what == 'dgrad' else ''}")
        w = worst.setdefault(what, [0, 0, 0, 0, 0, 0])
        worst[what] = [max(w[0], e6), max(w[1], e3), max(w[2], c6), max(w[3], c3), max(w[4], eb), max(w[5], ec)]
print("worst over the layers:")
for what, w in worst.items():
    print(f"  {what:6s} err/scale x6 {w[0]:.2e}  f16x3 {w[1]:.2e}  fp32 blas {w[4]:.2e}  fp32 k-ordered chain {w[5]:.2e}   component-wise x6 {w[2]:.2e}  f16x3 {w[3]:.2e}")
