#!/usr/bin/env python
"""Launch the MFMA kernels of the hot path at the shapes the configs use, labelled, for rocprofv3 passes
(kernel trace or --pmc).  Writes a launch manifest (labels in launch order) next to the profile so that
tools/pmc_mfma.py can attribute every dispatch.

    PECLR_LAUNCH_MANIFEST=gpurun_out/x/manifest.json rocprofv3 --kernel-trace --pmc <counters> -d ... -- \
        python tools/mfma_probe.py

Shapes: projection head at C2 (M = 256 rows, Din 2048 -> 512 -> 128): K1 forward, dW1, dh, K2 forward,
dW2, da; NT-Xent forward / backward at C2 (256 x 256), at C3's per-rank shape (256 local rows x 2048
gathered rows) and C5's (128 x 1024); the fused conv1x1 input-gradient + residual GEMM at ResNet-50's four
bottleneck-entry shapes for 256 images @224 (fp32) and its bf16 twin.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from peclr_amd import _capi  # noqa: E402

REPS = int(os.environ.get("PECLR_PROBE_REPS", "5"))
dev = "cuda"
torch.manual_seed(0)


def unit(m, d=128):
    z = torch.randn(m, d, device=dev)
    return (z / z.norm(dim=1, keepdim=True)).contiguous()


def main():
    m, din, hid, d = 256, 2048, 512, 128
    h, w1, w2 = torch.randn(m, din, device=dev), torch.randn(hid, din, device=dev), torch.randn(d, hid, device=dev)
    a, dp, dapre = torch.randn(m, hid, device=dev), torch.randn(m, d, device=dev), torch.randn(m, hid, device=dev)
    calls = [
        ("gemm_k1_fwd", lambda: _capi.gemm(_capi.GEMM_NT, h, w1, split_k=_capi.pick_split_k(m, hid, din), tag="gemm_k1_fwd")),
        ("gemm_k2_fwd", lambda: _capi.gemm(_capi.GEMM_NT, a, w2, split_k=_capi.pick_split_k(m, d, hid), tag="gemm_k2_fwd")),
        ("gemm_dw2", lambda: _capi.gemm(_capi.GEMM_TN, dp, a, tag="gemm_dw2")),
        ("gemm_da", lambda: _capi.gemm(_capi.GEMM_NN, dp, w2, tag="gemm_da")),
        ("gemm_dw1", lambda: _capi.gemm(_capi.GEMM_TN, dapre, h, tag="gemm_dw1")),
        ("gemm_dh", lambda: _capi.gemm(_capi.GEMM_NN, dapre, w1, tag="gemm_dh")),
    ]
    one = torch.ones(1, device=dev)
    for tag, mr, mg, n_half in (("c2_256x256", 256, 256, 128), ("c3_256x2048", 256, 2048, 128), ("c5_128x1024", 128, 1024, 64)):
        zall = unit(mg)
        rows = zall[:mr].contiguous()
        holder = {}

        def fwd(rows=rows, zall=zall, n_half=n_half, mg=mg, holder=holder, tag=tag):
            # EVENT names: ntxent_fwd / ntxent_finalize (renamed below through the manifest prefix)
            holder["lse"] = _capi.ntxent_fwd(rows, 0, zall, n_half, 2.0, 1.0 / mg, None, 0, False)[1]

        def bwd(rows=rows, zall=zall, n_half=n_half, mg=mg, holder=holder):
            lse_all = holder["lse"] if mg == rows.shape[0] else torch.zeros(mg, device=dev) + float(holder["lse"].mean())
            _capi.ntxent_bwd(rows, 0, zall, n_half, 2.0, lse_all, one, 1.0 / mg)

        calls += [(f"ntxent_fwd@{tag}", fwd), (f"ntxent_bwd@{tag}", bwd)]
    for r, cmid, cin in ((802816, 64, 256), (200704, 128, 512), (50176, 256, 1024), (12544, 512, 2048)):
        ga, wb, gd = torch.randn(r, cmid, device=dev), torch.randn(cmid, cin, device=dev), torch.randn(r, cin, device=dev)
        calls.append((f"conv1x1_dgrad_add@{r}x{cmid}x{cin}",
                      lambda ga=ga, wb=wb, gd=gd: _capi.gemm_add(_capi.GEMM_NN, ga, wb, gd, tag="conv1x1_dgrad_add")))
        gab, wbt, gdb = ga.bfloat16(), wb.t().contiguous().bfloat16(), gd.bfloat16()
        calls.append((f"conv1x1_dgrad_add_bf16@{r}x{cmid}x{cin}",
                      lambda gab=gab, wbt=wbt, gdb=gdb: _capi.gemm_add_bf16(gab, wbt, gdb, tag="conv1x1_dgrad_add")))
    # round 3: the six-product GEMM family at ResNet-50's shapes -- 1x1 forward / input gradient (peclr_gemm_x6p_f32, with and
    # without the residual-gradient addend), 3x3 forward (peclr_conv3x3_x6p_f32), weight gradients (peclr_gemm_x6t_f32)
    for r, n, k in ((200704, 128, 512), (200704, 512, 128), (50176, 256, 1024), (50176, 1024, 256), (12544, 512, 2048), (12544, 2048, 512)):
        xa, wt = torch.randn(r, k, device=dev), torch.randn(n, k, device=dev) * 0.05
        pk = _capi.X6Planes([(wt, False)]).pack()
        calls.append((f"gemm_x6p@{r}x{n}x{k}", lambda xa=xa, pk=pk, n=n: _capi.gemm_x6p(xa, pk.planes[0], n, tag="gemm_x6p")))
        if n > k:
            gd = torch.randn(r, n, device=dev)
            calls.append((f"gemm_x6p_add@{r}x{n}x{k}", lambda xa=xa, pk=pk, n=n, gd=gd: _capi.gemm_x6p(xa, pk.planes[0], n, gd, tag="gemm_x6p")))
        ga, gb = torch.randn(r, n, device=dev), torch.randn(r, k, device=dev)
        calls.append((f"gemm_x6t@{r}x{n}x{k}", lambda ga=ga, gb=gb: _capi.gemm_x6t(ga, gb, tag="gemm_x6t")))
    for c, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
        x4 = torch.randn(256, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        w4 = (torch.randn(c, c, 3, 3, device=dev) * 0.03).contiguous(memory_format=torch.channels_last)
        pk = _capi.X6Planes([(w4.permute(0, 2, 3, 1).reshape(c, 9 * c), False)]).pack()
        calls.append((f"conv3x3_x6p@256x{c}x{hw}", lambda x4=x4, pk=pk, c=c: _capi.conv3x3_x6p(x4, pk.planes[0], c, tile_rows=256, tag="conv3x3_x6p")))
        if c >= 128:
            x2 = x4.permute(0, 2, 3, 1).reshape(-1, c)
            calls.append((f"conv3x3_wgrad@256x{c}x{hw}", lambda x2=x2, hw=hw: _capi.gemm_x6t(x2, x2, taps=9, hw=(hw, hw), tag="conv3x3_wgrad")))
    for _, fn in calls:            # untimed warm-up (module load, first-touch)
        fn()
    torch.cuda.synchronize()
    order = []
    _capi.EVENT_LOG = {}
    for label, fn in calls:
        for _ in range(REPS):
            _capi.LAUNCH_ORDER = []
            fn()
            order += [f"{label}::{name}" if "@" in label else name for name in _capi.LAUNCH_ORDER]
    _capi.LAUNCH_ORDER = None
    torch.cuda.synchronize()
    path = os.environ.get("PECLR_LAUNCH_MANIFEST")
    if path:
        with open(path, "w") as f:
            json.dump({"order": order, "reps": REPS, "argv": sys.argv[1:]}, f)
    print(f"probe: {len(order)} labelled launches")


if __name__ == "__main__":
    main()
