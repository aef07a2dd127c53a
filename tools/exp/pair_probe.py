"""Round 6: the fp16-pair kernels (three products) against the six-product kernels at the shapes of the C2 step -- time per
launch (cold caches) and error against float64 -- first on synthetic operands, then on REAL layer tensors: the inputs X, weights W
and output gradients dY of ResNet-50's 1x1 and 3x3 convolutions captured from one training step (2 x 16 views @224, random
initialisation, batch statistics), including dY of layers 1 - 4.   python tools/exp/pair_probe.py [--real-only]
(The weight-gradient sections need the pair arms of the weight-gradient kernels: tools/exp/pair_wgrad.patch -- not adopted, see
DESIGN.md section 0; without them those sections are skipped.)"""
import inspect
import sys

import torch

sys.path.insert(0, ".")
from peclr_amd import _capi as capi  # noqa: E402

DEV = "cuda:0"
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def timed(fn, reps=7):
    ts = []
    for r in range(reps):
        junk.fill_(r)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


def err(c, ref):
    return float((c.double() - ref).abs().max()) / float(ref.abs().max())


def am(t):
    return t.abs().max().reshape(1).float()


WGRAD_PAIR = "absmax" in inspect.signature(capi.gemm_x6t).parameters


if "--real-only" not in sys.argv:
    print("== 1x1 products (rows, Cin -> Cout): us per launch six-product / pair, err/scale vs float64 (six / pair / v_mfma_f32)")
    for m, k, n in ((802816, 64, 256), (802816, 256, 64), (200704, 128, 512), (200704, 512, 128), (50176, 256, 1024), (50176, 1024, 256),
                    (12544, 512, 2048), (12544, 2048, 512)):
        a = torch.randn(m, k, device=DEV)
        bt = torch.randn(n, k, device=DEV) * 0.05
        p6 = capi.X6Planes([(bt, False)]).pack().planes[0]
        pp = capi.X6Planes([(bt, False)], pair=True).pack()
        pair = (am(a), pp.scale(0))
        row = []
        for tr in (128, 256):
            row.append((timed(lambda: capi.gemm_x6p(a, p6, n, tile_rows=tr)), timed(lambda: capi.gemm_x6p(a, pp.planes[0], n, tile_rows=tr, pair=pair))))
        sl = slice(0, min(m, 16384))
        ref = a[sl].double() @ bt.double().t()
        e6, e2 = err(capi.gemm_x6p(a, p6, n)[sl], ref), err(capi.gemm_x6p(a, pp.planes[0], n, pair=pair)[sl], ref)
        ef = err(capi.gemm(capi.GEMM_NT, a[sl].contiguous(), bt), ref)
        print(f"  {m:7d} {k:5d} -> {n:5d}: 128-row {row[0][0]:7.1f} / {row[0][1]:7.1f} ({row[0][1] / row[0][0] - 1:+.0%})   256-row {row[1][0]:7.1f} / {row[1][1]:7.1f} "
              f"({row[1][1] / row[1][0] - 1:+.0%})   err {e6:.2e} / {e2:.2e} / {ef:.2e}", flush=True)
        del a, bt
    print("== 3x3 / stride 1 (images, H = W, C): us six-product / pair (halo form, 256-row tiles), err/scale (six / pair / MIOpen)")
    for nb, hw, c in ((256, 56, 64), (256, 28, 128), (256, 14, 256), (256, 7, 512)):
        x = torch.randn(nb, c, hw, hw, device=DEV).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(c, c, 3, 3, device=DEV) * 0.03).contiguous(memory_format=torch.channels_last)
        spec = [(w.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous(), False)]
        p6 = capi.X6Planes(spec).pack().planes[0]
        pp = capi.X6Planes(spec, pair=True).pack()
        pair = (am(x), pp.scale(0))
        t6 = timed(lambda: capi.conv3x3_x6p(x, p6, c, tile_rows=256))
        t2 = timed(lambda: capi.conv3x3_x6p(x, pp.planes[0], c, tile_rows=256, pair=pair))
        t2b = timed(lambda: capi.conv3x3_x6p(x, pp.planes[0], c, tile_rows=128, pair=pair))
        xs = x[:16]
        ref = torch.nn.functional.conv2d(xs.double(), w.double(), padding=1)
        e6, e2 = err(capi.conv3x3_x6p(xs, p6, c), ref), err(capi.conv3x3_x6p(xs, pp.planes[0], c, pair=(am(xs), pp.scale(0))), ref)
        em = err(torch.nn.functional.conv2d(xs, w, padding=1), ref)
        print(f"  {nb} x {hw:2d} x {hw:2d} x {c:3d}: {t6:7.1f} / {t2:7.1f} ({t2 / t6 - 1:+.0%}; 128-row tiles {t2b:7.1f})   err {e6:.2e} / {e2:.2e} / {em:.2e}", flush=True)
        del x, w

if "--real-only" not in sys.argv and WGRAD_PAIR:
    print("== weight gradients (rows, Cout, Cin): us six-product / pair, component-wise err (six / pair)")
    for k, m, n in ((802816, 64, 256), (802816, 256, 64), (200704, 512, 128), (200704, 128, 512), (50176, 1024, 256), (50176, 256, 1024),
                    (12544, 2048, 512), (12544, 512, 2048)):
        a = torch.randn(k, m, device=DEV) * 1e-4
        b = torch.randn(k, n, device=DEV).clamp_min(0)
        ab = (am(a), am(b))
        t6, t2 = timed(lambda: capi.gemm_x6t(a, b)), timed(lambda: capi.gemm_x6t(a, b, absmax=ab))
        sl = slice(0, min(k, 32768))
        a_, b_ = a[sl].contiguous(), b[sl].contiguous()
        ref = a_.double().t() @ b_.double()
        bound = a_.double().abs().t() @ b_.double().abs()
        e6 = float(((capi.gemm_x6t(a_, b_).double() - ref).abs() / bound).max())
        e2 = float(((capi.gemm_x6t(a_, b_, absmax=(am(a_), am(b_))).double() - ref).abs() / bound).max())
        print(f"  1x1 {k:7d} {m:5d} x {n:5d}: {t6:7.1f} / {t2:7.1f} ({t2 / t6 - 1:+.0%})   err {e6:.2e} / {e2:.2e}", flush=True)
        del a, b
    for nb, hw, c in ((256, 56, 64), (256, 28, 128), (256, 14, 256), (256, 7, 512)):
        x = torch.randn(nb, c, hw, hw, device=DEV).clamp_min(0).contiguous(memory_format=torch.channels_last)
        gy = (torch.randn(nb, c, hw, hw, device=DEV) * 1e-4).contiguous(memory_format=torch.channels_last)
        ab = (am(gy), am(x))
        if capi.wgrad3_x6r_ok(gy, x):
            t6, t2 = timed(lambda: capi.wgrad3_x6r(gy, x)), timed(lambda: capi.wgrad3_x6r(gy, x, absmax=ab))
            print(f"  3x3 ring {nb} x {hw} x {hw} x {c}: {t6:7.1f} / {t2:7.1f} ({t2 / t6 - 1:+.0%})", flush=True)
        r = nb * hw * hw
        gy2, x2 = gy.permute(0, 2, 3, 1).reshape(r, c), x.permute(0, 2, 3, 1).reshape(r, c)
        t6, t2 = timed(lambda: capi.gemm_x6t(gy2, x2, taps=9, hw=(hw, hw))), timed(lambda: capi.gemm_x6t(gy2, x2, taps=9, hw=(hw, hw), absmax=ab))
        print(f"  3x3 nine-tap {nb} x {hw} x {hw} x {c}: {t6:7.1f} / {t2:7.1f} ({t2 / t6 - 1:+.0%})", flush=True)
        del x, gy

# ---- real layer tensors
print("== real layer tensors (ResNet-50 training step, 2 x 16 views @224): err/scale vs float64, six-product / pair / v_mfma_f32 (1x1) or MIOpen (3x3)")
from peclr_amd.resnet import resnet50  # noqa: E402

torch.manual_seed(0)
model = resnet50().to(DEV).train()
taps = {}
for name, mod in model.named_modules():
    if isinstance(mod, torch.nn.Conv2d) and mod.stride == (1, 1) and mod.kernel_size in ((1, 1), (3, 3)):
        def hook(m_, inp, out, name=name):
            taps[name] = [inp[0].detach(), m_.weight.detach(), None]
            out.register_hook(lambda g_, name=name: taps[name].__setitem__(2, g_.detach()))
        mod.register_forward_hook(hook)
xin = torch.randn(32, 3, 224, 224, device=DEV)
y = model(xin)
(y.square().mean() + y.mean()).backward()
worst = {}
for name, (X, W, dY) in taps.items():
    if not name.split(".")[1] in ("0", "1"):
        continue
    cout, cin = W.shape[:2]
    if W.shape[2] == 1:
        X2 = X.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
        dY2 = dY.permute(0, 2, 3, 1).reshape(-1, cout).contiguous()
        W2 = W.reshape(cout, cin).contiguous()
        for what, a, bt in (("fwd", X2, W2), ("dgrad", dY2, W2.t().contiguous())):
            if a.shape[1] % 16 or bt.shape[0] % 64:
                continue
            n = bt.shape[0]
            ref = a.double() @ bt.double().t()
            p6 = capi.X6Planes([(bt, False)]).pack().planes[0]
            pp = capi.X6Planes([(bt, False)], pair=True).pack()
            e6, e2 = err(capi.gemm_x6p(a, p6, n), ref), err(capi.gemm_x6p(a, pp.planes[0], n, pair=(am(a), pp.scale(0))), ref)
            ef = err(capi.gemm(capi.GEMM_NT, a, bt), ref)
            rng = f"max/median |A| {float(a.abs().max()):.1e} / {float(a.abs().median()):.1e}"
            print(f"  {name:22s} {what:6s} [{a.shape[0]} x {a.shape[1]}] . [{n}]: {e6:.2e} / {e2:.2e} / {ef:.2e}   {rng}", flush=True)
            w_ = worst.setdefault(what, [0.0, 0.0, 0.0])
            worst[what] = [max(w_[0], e6), max(w_[1], e2), max(w_[2], ef)]
        if WGRAD_PAIR and cout % 4 == 0 and cin % 4 == 0 and cout >= 64 and cin >= 64:
            ref = dY2.double().t() @ X2.double()
            e6 = err(capi.gemm_x6t(dY2, X2), ref)
            e2 = err(capi.gemm_x6t(dY2, X2, absmax=(am(dY2), am(X2))), ref)
            ef = err(torch.ops.aten.convolution_backward(dY.contiguous(memory_format=torch.channels_last), X.contiguous(memory_format=torch.channels_last),
                                                        W, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1].reshape(cout, cin), ref)
            print(f"  {name:22s} wgrad  [{dY2.shape[0]} x {cout}]^T . [{cin}]: {e6:.2e} / {e2:.2e} / {ef:.2e} (MIOpen)", flush=True)
            w_ = worst.setdefault("wgrad", [0.0, 0.0, 0.0])
            worst["wgrad"] = [max(w_[0], e6), max(w_[1], e2), max(w_[2], ef)]
    else:
        Xc, dYc = X.contiguous(memory_format=torch.channels_last), dY.contiguous(memory_format=torch.channels_last)
        Wc = W.contiguous(memory_format=torch.channels_last)
        for what, a, spec, flip in (("fwd3", Xc, [(Wc.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous(), False)], False),
                                    ("dgrad3", dYc, [(Wc.permute(0, 2, 3, 1).reshape(cout * 9, cin).contiguous(), 9)], True)):
            ref = (torch.nn.functional.conv2d(a.double(), W.double(), padding=1) if not flip
                   else torch.nn.functional.conv_transpose2d(a.double(), W.double(), padding=1))
            p6 = capi.X6Planes(spec).pack().planes[0]
            pp = capi.X6Planes(spec, pair=True).pack()
            co = cout if not flip else cin
            e6, e2 = err(capi.conv3x3_x6p(a, p6, co, flip=flip), ref), err(capi.conv3x3_x6p(a, pp.planes[0], co, flip=flip, pair=(am(a), pp.scale(0))), ref)
            em = err(torch.nn.functional.conv2d(a, W, padding=1) if not flip else torch.nn.functional.conv_transpose2d(a, W, padding=1), ref)
            rng = f"max/median |A| {float(a.abs().max()):.1e} / {float(a.abs().median()):.1e}"
            print(f"  {name:22s} {what:6s} {tuple(a.shape)}: {e6:.2e} / {e2:.2e} / {em:.2e}   {rng}", flush=True)
            w_ = worst.setdefault(what, [0.0, 0.0, 0.0])
            worst[what] = [max(w_[0], e6), max(w_[1], e2), max(w_[2], em)]
        if not WGRAD_PAIR:
            continue
        # the 3x3 weight gradient: nine-tap kernel and (where it takes the shape) the ring kernel
        wz = torch.zeros_like(Wc)
        args = (None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        ref = torch.ops.aten.convolution_backward(dYc.double(), Xc.double(), wz.double(), *args)[1]
        em = err(torch.ops.aten.convolution_backward(dYc, Xc, wz, *args)[1], ref)
        r_ = Xc.shape[0] * Xc.shape[2] * Xc.shape[3]
        gy2, x2 = dYc.permute(0, 2, 3, 1).reshape(r_, cout), Xc.permute(0, 2, 3, 1).reshape(r_, cin)
        as_w = lambda t: t.view(cout, 3, 3, cin).permute(0, 3, 1, 2)          # noqa: E731
        hw_ = (Xc.shape[2], Xc.shape[3])
        e6 = err(as_w(capi.gemm_x6t(gy2, x2, taps=9, hw=hw_)), ref)
        e2 = err(as_w(capi.gemm_x6t(gy2, x2, taps=9, hw=hw_, absmax=(am(gy2), am(x2)))), ref)
        line = f"  {name:22s} wgrad3 nine-tap: {e6:.2e} / {e2:.2e} / {em:.2e}"
        w_ = worst.setdefault("wgrad3", [0.0, 0.0, 0.0])
        worst["wgrad3"] = [max(w_[0], e6), max(w_[1], e2), max(w_[2], em)]
        if capi.wgrad3_x6r_ok(dYc, Xc):
            e6r = err(as_w(capi.wgrad3_x6r(dYc, Xc)), ref)
            e2r = err(as_w(capi.wgrad3_x6r(dYc, Xc, absmax=(am(dYc), am(Xc)))), ref)
            line += f"   ring: {e6r:.2e} / {e2r:.2e}"
            w_ = worst.setdefault("wgrad3r", [0.0, 0.0, 0.0])
            worst["wgrad3r"] = [max(w_[0], e6r), max(w_[1], e2r), max(w_[2], em)]
        print(line, flush=True)
print("worst over the layers (six-product / pair / fp32 reference kernel):")
for k_, v in worst.items():
    print(f"  {k_:7s} {v[0]:.2e} / {v[1]:.2e} / {v[2]:.2e}")
