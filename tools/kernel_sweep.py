#!/usr/bin/env python
"""Roofline sweep of the hand-written kernels at sizes where they are NOT launch-latency bound.

At the head's real sizes (M = 256) every kernel is a few microseconds of latency; this sweep shows
what each kernel reaches on its own roofline as the problem grows (gathered negative sets, larger
batches).  Timing: HIP events around `iters` back-to-back launches on the launch stream.
Usage: python tools/kernel_sweep.py [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from peclr_amd import _capi

DEV = torch.device("cuda:0")
HBM, MFMA = 8000.0, 157.3


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def row(name, shape, us, flops=0, nbytes=0):
    r = {"kernel": name, "shape": shape, "us": round(us, 2)}
    if flops:
        tf = flops / us / 1e6
        r.update(tflops=round(tf, 2), mfma_frac=round(tf / MFMA, 4))
    if nbytes:
        gbs = nbytes / us / 1e3
        r.update(gbs=round(gbs, 1), hbm_frac=round(gbs / HBM, 4))
    print(r, flush=True)
    return r


def main():
    out = []
    L = _capi.lib()
    st = lambda: torch.cuda.current_stream().cuda_stream
    # ---- NT-Xent forward / backward
    for m in (256, 1024, 2048, 4096, 8192, 16384):
        z = torch.nn.functional.normalize(torch.randn(m, 128, device=DEV))
        n = m // 2
        jf, jb = _capi.ntxent_jsplit(m, m, False), _capi.ntxent_jsplit(m, m, True)
        part = torch.empty(jf, m, device=DEV); pos = torch.empty(m, device=DEV)
        lse = torch.empty(m, device=DEV); o17 = torch.zeros(17, device=DEV)
        fwd = lambda: L.peclr_ntxent_fwd_f32(z.data_ptr(), m, 0, z.data_ptr(), m, 128, n, 2.0, None, part.data_ptr(),
                                             pos.data_ptr(), jf, st())
        us = timed(fwd)
        out.append(row("ntxent_fwd", f"M={m} jsplit={jf}", us, 2 * m * m * 128, 512 * 2 * m))
        L.peclr_ntxent_finalize_f32(part.data_ptr(), jf, pos.data_ptr(), m, 1.0 / m, lse.data_ptr(), None, 0,
                                    o17.data_ptr(), st())
        slabs = torch.empty(jb, m, 128, device=DEV); one = torch.ones(1, device=DEV)
        bwd = lambda: L.peclr_ntxent_bwd_f32(z.data_ptr(), m, 0, z.data_ptr(), m, 128, n, 2.0, lse.data_ptr(),
                                             one.data_ptr(), 1.0 / m, slabs.data_ptr(), jb, st())
        us = timed(bwd)
        out.append(row("ntxent_bwd", f"M={m} jsplit={jb}", us, 4 * m * m * 128, 512 * 2 * m + 512 * m * jb))
    # ---- NT-Xent at the 8-GPU config C3: 256 local rows against the 2048 gathered rows
    z = torch.nn.functional.normalize(torch.randn(2048, 128, device=DEV))
    zr = z[512:768].contiguous()
    jf, jb = _capi.ntxent_jsplit(256, 2048, False), _capi.ntxent_jsplit(256, 2048, True)
    part = torch.empty(jf, 256, device=DEV); pos = torch.empty(256, device=DEV); lse = torch.zeros(2048, device=DEV)
    one = torch.ones(1, device=DEV); slabs = torch.empty(jb, 256, 128, device=DEV)
    us = timed(lambda: L.peclr_ntxent_fwd_f32(zr.data_ptr(), 256, 512, z.data_ptr(), 2048, 128, 128, 2.0, None,
                                              part.data_ptr(), pos.data_ptr(), jf, st()))
    out.append(row("ntxent_fwd", f"Mr=256 x Mg=2048 (C3 per rank) jsplit={jf}", us, 2 * 256 * 2048 * 128, 512 * (256 + 2048)))
    us = timed(lambda: L.peclr_ntxent_bwd_f32(zr.data_ptr(), 256, 512, z.data_ptr(), 2048, 128, 128, 2.0, lse.data_ptr(),
                                              one.data_ptr(), 1.0 / 2048, slabs.data_ptr(), jb, st()))
    out.append(row("ntxent_bwd", f"Mr=256 x Mg=2048 (C3 per rank) jsplit={jb}", us, 4 * 256 * 2048 * 128,
                   512 * (256 + 2048) + 512 * 256 * jb))
    # ---- GEMM (NT: the K1 forward shape family, then square)
    for (m, n, k) in ((256, 512, 2048), (2048, 512, 2048), (4096, 4096, 4096), (8192, 2048, 2048)):
        a = torch.randn(m, k, device=DEV); b = torch.randn(n, k, device=DEV)
        s = _capi.pick_split_k(m, n, k)
        us = timed(lambda: _capi.gemm(_capi.GEMM_NT, a, b, split_k=s))
        out.append(row("gemm_nt", f"{m}x{n}x{k} split_k={s}", us, 2 * m * n * k, 4 * (m * k + n * k + s * m * n)))
        if m >= 2048:
            bt = torch.randn(k, n, device=DEV); at = torch.randn(k, m, device=DEV)
            us = timed(lambda: _capi.gemm(_capi.GEMM_NN, a, bt))
            out.append(row("gemm_nn", f"{m}x{n}x{k}", us, 2 * m * n * k, 4 * (m * k + n * k + m * n)))
            us = timed(lambda: _capi.gemm(_capi.GEMM_TN, at, bt))
            out.append(row("gemm_tn", f"{m}x{n}x{k}", us, 2 * m * n * k, 4 * (m * k + n * k + m * n)))
    # ---- BN + ReLU, align
    for m in (256, 4096, 65536):
        h = 512
        x = torch.randn(1, m, h, device=DEV); g = torch.ones(h, device=DEV); b = torch.zeros(h, device=DEV)
        if m <= 1024:   # register-resident kernel (what ops.head_align uses up to 1024 rows)
            us = timed(lambda: _capi.bn_relu_fwd(x, b, g, b, 1e-5, 0.1, True, None, None, None))
            out.append(row("bn_relu_fwd", f"M={m} H={h}", us, 0, 4 * m * h * 3))
        else:           # streaming stats/finalize/apply kernels (ops.head_align above 1024 rows)
            x4 = x[0].view(m, h, 1, 1)
            us = timed(lambda: _capi.bn2d_fwd(x4, None, g, b, None, None, None, True, 1e-5, 0.1, True))
            out.append(row("bn_relu_fwd (streaming: bn2d stats+finalize+apply)", f"M={m} H={h}", us, 0, 4 * m * h * 3))
        p = torch.randn(1, m, 128, device=DEV)
        n = m // 2
        jit = tuple(torch.randint(-14, 1, (n,), device=DEV) for _ in range(4))
        ang = tuple(torch.randint(-45, 46, (n,), device=DEV).double() for _ in range(2))
        us = timed(lambda: _capi.align_fwd(p, n, 3, jit, (224, 224), ang))
        out.append(row("align_fwd", f"M={m}", us, 0, m * (512 * 3 + 48)))
        pp, zz, nn_, _ = _capi.align_fwd(p, n, 3, jit, (224, 224), ang)
        dz = torch.randn(m, 128, device=DEV)
        us = timed(lambda: _capi.align_bwd(dz, pp, zz, nn_, n, 3, ang))
        out.append(row("align_bwd", f"M={m}", us, 0, m * (512 * 4 + 16)))
    # ---- LARS/Adam over one big tensor list (ResNet-50 sized: 23.5 M parameters)
    from peclr_amd.optim import LARSAdam
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in (2048 * 512 * 9, 1024 * 2048, 512 * 512 * 9, 64 * 147,
                                                                    2048 * 1024, 2359296, 2359296, 1048576, 1179648)]
    for p in ps:
        p.grad = torch.randn_like(p)
    opt = LARSAdam(ps, lr=1e-3, lars=True)
    opt.step()
    tot = sum(p.numel() for p in ps)
    _capi.EVENT_LOG = {}
    for _ in range(10):
        opt.step()
    torch.cuda.synchronize()
    log, _capi.EVENT_LOG = _capi.EVENT_LOG, None
    for name, bpp in (("lars_sumsq", 8), ("lars_adam_update", 28)):
        us = sum(r[0].elapsed_time(r[1]) for r in log[name]) * 1e3 / len(log[name])
        out.append(row(name, f"{tot/1e6:.1f}M params", us, 0, bpp * tot))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
