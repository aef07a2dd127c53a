"""Shader clocks per loop segment of the ring 3x3 kernel (workgroup 0, wave 0): needs the timing build, PECLR_HIP_LIB=tools/exp/ab/libtiming.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"; dt = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
L = capi.lib()
buf = (ctypes.c_ulonglong * 8)()
for hw, c, nb in ((7, 128, 256), (7, 512, 256), (14, 256, 256), (28, 128, 256), (56, 64, 256)):
    x = torch.randn(nb, c, hw, hw, device=DEV, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, 9 * c, device=DEV, generator=g) * 0.03
    pk = capi.HPlanes([(w, False)], dt).pack()
    for _ in range(3): capi.conv_h(x, pk.planes[0], c, tile_rows=1)
    torch.cuda.synchronize()
    L.peclr_debug_conv_h_timing(buf, 1)
    reps = 10
    for _ in range(reps): capi.conv_h(x, pk.planes[0], c, tile_rows=1)
    torch.cuda.synchronize()
    L.peclr_debug_conv_h_timing(buf, 1)
    steps = buf[4]
    print(f"{hw}x{hw} x {c}: per step clocks: wait {buf[0] / steps:7.0f} | barrier {buf[1] / steps:7.0f} | issue {buf[2] / steps:7.0f} | reads + products {buf[3] / steps:7.0f}   ({steps // reps} steps)")
