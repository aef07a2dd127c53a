"""Weight gradient of ResNet-50's 1x1 stride-1 convolutions: MIOpen's wrw (incl. its zero-fill / cast launches) vs a
plain library GEMM dW = gy2d^T @ x2d (torch.mm -> hipBLASLt), fp32 and bf16, NHWC views (no copies)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(sys.path[0], ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(sys.path[0], ".miopen", "cache"))
import torch

N = 256
shapes = [(56, 64, 64, 1), (56, 256, 64, 2), (56, 64, 256, 4), (56, 256, 128, 1), (28, 512, 128, 3), (28, 128, 512, 4),
          (28, 512, 256, 1), (14, 1024, 256, 5), (14, 256, 1024, 6), (14, 1024, 512, 1), (7, 2048, 512, 2), (7, 512, 2048, 3)]


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


for dtype in (torch.float32, torch.bfloat16):
    out, tot_mi, tot_mm = [], 0.0, 0.0
    for hw, cin, cout, cnt in shapes:
        r = N * hw * hw
        x = torch.randn(N, cin, hw, hw, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(N, cout, hw, hw, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 1, 1, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        mi = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                                [False, True, False])[1].float())
        x2, g2 = x.permute(0, 2, 3, 1).reshape(r, cin), gy.permute(0, 2, 3, 1).reshape(r, cout)
        mm = timeit(lambda: torch.mm(g2.t(), x2).float())
        ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        got = torch.mm(g2.t(), x2)
        err = float((got.float() - ref.float().reshape(cout, cin)).norm() / ref.float().norm())
        out.append((hw, cin, cout, cnt, round(mi, 1), round(mm, 1), f"{err:.1e}"))
        tot_mi += cnt * mi
        tot_mm += cnt * mm
        del x, gy
    print(json.dumps({"dtype": str(dtype), "per_step_us": {"miopen": round(tot_mi), "mm": round(tot_mm)}, "shapes(hw,cin,cout,count,miopen_us,mm_us,rel)": out}))
