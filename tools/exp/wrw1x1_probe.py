"""Weight gradient of ResNet-50's 1x1 convolutions (2x128 views @224, NHWC fp32): MIOpen's wrw (as autograd calls
it, incl. its zero-fill for split-K kernels) vs the hand-written TN GEMM with deterministic split-K slabs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(sys.path[0], ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(sys.path[0], ".miopen", "cache"))
import torch
from peclr_amd import _capi

N = 256
shapes = [  # (H=W, Cin, Cout, count per step)
    (56, 64, 64, 1), (56, 256, 64, 2), (56, 64, 256, 4), (56, 256, 128, 1), (28, 512, 128, 3), (28, 128, 512, 4),
    (28, 512, 256, 1), (14, 1024, 256, 5), (14, 256, 1024, 6), (14, 1024, 512, 1), (7, 2048, 512, 2), (7, 512, 2048, 3),
]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out, tot_mi, tot_me = [], 0.0, 0.0
for hw, cin, cout, cnt in shapes:
    r = N * hw * hw
    x = torch.randn(N, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, cout, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    mi = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                            [False, True, False]))
    x2, g2 = x.permute(0, 2, 3, 1).reshape(r, cin), gy.permute(0, 2, 3, 1).reshape(r, cout)
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    best = None
    tiles = ((cout + 63) // 64) * ((cin + 63) // 64)
    for target in (512, 1024, 2048, 4096):
        s = max(1, min(target // tiles, r // 256))
        def mine(s=s):
            slabs = _capi.gemm(_capi.GEMM_TN, g2, x2, split_k=s)
            return _capi.slab_reduce(slabs) if s > 1 else slabs
        t = timeit(mine)
        if best is None or t < best[0]:
            best = (t, s)
    got = (_capi.slab_reduce(_capi.gemm(_capi.GEMM_TN, g2, x2, split_k=best[1])) if best[1] > 1 else _capi.gemm(_capi.GEMM_TN, g2, x2))
    err = float((got - ref.reshape(cout, cin)).norm() / ref.norm())
    flops = 2.0 * r * cin * cout
    out.append({"hw": hw, "cin": cin, "cout": cout, "count": cnt, "miopen_us": round(mi, 1), "mine_us": round(best[0], 1),
                "split": best[1], "miopen_tf": round(flops / mi / 1e6, 1), "mine_tf": round(flops / best[0] / 1e6, 1), "rel_err": err})
    tot_mi += cnt * mi
    tot_me += cnt * best[0]
    del x, gy
print(json.dumps({"per_step_us": {"miopen": round(tot_mi), "mine": round(tot_me)}, "shapes": out}, indent=1))
