#!/usr/bin/env python
"""Headline benchmark: images/sec (2-view) of the full PeCLR pretraining step
(ResNet-50 encoder -> projection head -> equivariance alignment -> NT-Xent -> backward ->
LARS/Adam step), per-device view batch 128 (256 images/step/device), synthetic 224x224 inputs,
at 1/2/4/8 MI355X (BASELINE.json: configs[1] at N=1, the same per-GPU work under data parallel
for N>1 -- weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # no launcher: spawns its own 8 ranks (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints TWO lines on stdout:
  `BENCH_DETAILS {...}`  first: the full record -- the per-kernel table ("kernels": HIP-event average, algorithmic
                         bytes / flops, fraction of the roof of the kernel that ran, for every hand-written launch tag),
                         the long-form cpu_baseline / config / dist objects.  Also written to
                         gpurun_out/bench_details.json (or $PECLR_BENCH_DETAILS) when that directory exists;
  `{...}`                LAST: the compact line of the driver's contract (< 4 KB) with
      "roofline"     the dominant hand-written kernel (by time): algorithmic bytes/flops per launch
                     / its average launch duration measured with HIP events on the launch stream
                     inside the timed region (one C-ABI entry point = one launch);
      "backbone"     the end-to-end figure of the encoder against the v_mfma_f32 peak AND the six-product roof;
      "cpu_baseline" the same step on the host cores: torch-CPU ResNet + the NumPy oracle head
                     (oracle/peclr_oracle.py) + the foreach LARS/Adam, on a bounded sample;
      "parity", "fp32_gemm_check".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MIOpen JIT-compiles every convolution kernel on first use (~70 s for ResNet-50 on a fresh box: the
# PyTorch wheel ships no gfx950 kernel database).  Keep its user find-db / kernel cache in-tree so a
# populated cache travels with the repo snapshot; an empty or stale one only costs the JIT again.
_MIOPEN_DIR = os.path.join(ROOT, ".miopen")
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(_MIOPEN_DIR, "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(_MIOPEN_DIR, "cache"))
for _d in (os.environ["MIOPEN_USER_DB_PATH"], os.environ["MIOPEN_CUSTOM_CACHE_DIR"]):
    os.makedirs(_d, exist_ok=True)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TF = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--resnet", default="50")
    ap.add_argument("--pairs", type=int, default=128, help="view pairs per device (N); 2N images/step/device")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="backbone compute dtype (configs[1] is fp32; head/logits/loss are always fp32); fp16 = the "
                         "reference's precision 16 (native AMP: dynamic loss scaling, decided on the device inside the fused "
                         "optimiser step, so it runs in the hipGraphs too)")
    ap.add_argument("--channels-last", type=int, default=1,
                    help="NHWC activations/weights for the MIOpen backbone (default): its gfx950 igemm kernels "
                         "are NHWC-native, NCHW costs ~11%% of the step in layout transposes "
                         "(profiles/r01a vs r01b)")
    ap.add_argument("--fused-bn", type=int, default=1,
                    help="backbone BatchNorm2d/add/ReLU glue on the hand-written NHWC kernels (needs "
                         "--channels-last 1; fp32 or bf16 activations); 0 = stock PyTorch/MIOpen ops")
    ap.add_argument("--fork-gemm", type=int, default=1,
                    help="bottleneck entry: conv1's input gradient + the residual branch's gradient as ONE hand-written "
                         "GEMM (needs --fused-bn 1; fp32 or bf16); 0 = MIOpen dgrad + autograd's elementwise add")
    ap.add_argument("--augment", type=int, default=0,
                    help="1: every step draws a fresh batch through the GPU two-view augmentation (uint8 224x224 source "
                         "images + 2.5D joints resident in HBM -> rotate / crop / resize / colour jitter / normalise "
                         "kernels -> batch dict) instead of re-using one synthetic batch; the metric's default is 0")
    ap.add_argument("--checkpoint", type=int, default=0,
                    help="1: activation checkpointing in the encoder (residual blocks keep only their input for backward "
                         "and re-run there): for batch sizes / resolutions beyond the activation budget; ~1/3 more compute")
    ap.add_argument("--overlap-wgrad", type=int, default=0,
                    help="1: the backbone's weight gradients run on a second HIP stream, next to the BatchNorm / residual "
                         "glue of the layers below them (a parallel branch of the captured graph)")
    ap.add_argument("--accum", type=int, default=1)
    ap.add_argument("--sync-bn", type=int, default=0,
                    help="1: BatchNorm statistics over the global batch (exact N-rank == 1-device semantics, two small "
                         "all-reduces per BN layer); 0: per-rank statistics, like DDP without SyncBatchNorm")
    ap.add_argument("--miopen-find", type=int, default=0,
                    help="1 = torch.backends.cudnn.benchmark (MIOpen exhaustive find; tens of minutes of kernel "
                         "JIT on a box without a populated user find-db)")
    ap.add_argument("--graph", default="auto", choices=["auto", "0", "1"],
                    help="1: the timed steps replay ONE hipGraph of the whole step (forward + backward + fused "
                         "optimiser, Trainer.capture_step_graph); 0: eager launches; auto (default): at N=1 try the "
                         "graph in a child process and fall back to eager if that process dies (a failed capture is a "
                         "crash inside the HIP runtime, not an exception); at N>1 the step is split into two graphs "
                         "around its collectives (Trainer.capture_split_graphs), falling back to eager on all ranks "
                         "if any rank's capture raises")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16-companion", type=int, default=1,
                    help="fp32 runs at --gpus 1: after the fp32 measurement, and outside it, time the same K steps under --dtype bf16 "
                         "in a child process and attach {ms_per_step, value, roofline} as `bf16_companion` (BASELINE configs C3 / C5 "
                         "are bf16 configurations; 0 = skip)")
    ap.add_argument("--dry-dist", action="store_true",
                    help="form the process group (spawning the ranks if no launcher did), check that every rank is seen, "
                         "print {\"dry_dist\": ...} and stop: the N > 1 launch path without a step (runs over gloo on CPU)")
    ap.add_argument("--cpu-pairs", type=int, default=8, help="pairs in the probe step (and in the bounded CPU-baseline sample when the whole workload does not fit the budget)")
    ap.add_argument("--cpu-whole", type=int, default=-1,
                    help="CPU baseline on the WHOLE workload (1 warm-up + 3 timed steps): 1 always, 0 never (bounded sample of "
                         "2 x --cpu-pairs views), -1 (default) when a probe step predicts that it fits --cpu-budget")
    ap.add_argument("--cpu-budget", type=float, default=400.0,
                    help="seconds of CPU work the whole-workload baseline may be predicted to take (the default configuration, "
                         "ResNet-50 2 x 128 @224, needs ~100 s on the MI355X hosts)")
    return ap.parse_args()


def synthetic_batch(n, size, seed, device, channels_last=False):
    """randn images (post-normalisation ~ N(0,1)), integer-degree float64 angles
    in [-45,45], int64 jitter in [-14,0]; generator seed = the reference's seed 5 (+ rank).
    On a GPU the two views live in ONE [2N,3,S,S] buffer (`transformed_images`) and
    `transformed_image1/2` are its halves, so the step's `cat(view1, view2)` (hybrid2_model.py:30-32)
    is a no-op instead of a 154 MB copy per step."""
    g = torch.Generator().manual_seed(seed)
    im1 = torch.randn(n, 3, size, size, generator=g)
    im2 = torch.randn(n, 3, size, size, generator=g)
    b = {"jitter_x_1": torch.randint(-14, 1, (n,), generator=g),
         "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_1": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
         "angle_1": torch.randint(-45, 46, (n,), generator=g).double(),
         "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    b = {k: v.to(device) for k, v in b.items()}
    if device.type == "cpu":
        return {"transformed_image1": im1, "transformed_image2": im2, **b}
    stacked = torch.cat([im1, im2]).to(device)
    if channels_last:
        stacked = stacked.contiguous(memory_format=torch.channels_last)
    return {"transformed_images": stacked, "transformed_image1": stacked[:n], "transformed_image2": stacked[n:], **b}


def build_model(args, device, pairs):
    from peclr_amd import Hybrid2Model, hybrid2_config

    din = 512 if args.resnet in ("18", "34") else 2048
    cfg = hybrid2_config(resnet_size=args.resnet, projection_head_input_dim=din, augmentation=["crop", "rotate"],
                         batch_size=pairs, num_of_mini_batch=args.accum, pretrained=False)
    torch.manual_seed(5)
    return Hybrid2Model(cfg).to(device).train()


def kernel_table(event_log, m_rows, m_global, din, hid, n_params):
    """Average launch duration per hand-written kernel (HIP events) + algorithmic work per launch
    (element size 4 B everywhere on this fp32 path)."""
    from peclr_amd import _capi

    d = 128
    s1 = _capi.pick_split_k(m_rows, hid, din)
    s2 = _capi.pick_split_k(m_rows, d, hid)
    work = {  # name -> (bound, flops, bytes)
        "gemm_k1_fwd": ("mfma", 2 * m_rows * din * hid, 4 * (m_rows * din + din * hid + s1 * m_rows * hid)),
        "bn_relu_fwd": ("hbm", 0, 4 * m_rows * hid * (s1 + 2)),
        "gemm_k2_fwd": ("mfma", 2 * m_rows * hid * d, 4 * (m_rows * hid + hid * d + s2 * m_rows * d)),
        "align_fwd": ("hbm", 0, m_rows * (512 * s2 + 512 + 512 + 48)),
        "ntxent_fwd": ("mfma", 2 * m_rows * m_global * d, 512 * (m_rows + m_global)),
        "ntxent_finalize": ("hbm", 0, 4 * m_rows * (_capi.ntxent_jsplit(m_rows, m_global, False) + 10)),
        "ntxent_bwd": ("mfma", 4 * m_rows * m_global * d,
                       512 * (m_rows + m_global) + 512 * m_rows * _capi.ntxent_jsplit(m_rows, m_global, True)),
        "align_bwd": ("hbm", 0, m_rows * (4 * 512 + 16)),
        "gemm_dw2": ("mfma", 2 * m_rows * hid * d, 4 * (m_rows * d + m_rows * hid + hid * d)),
        "gemm_da": ("mfma", 2 * m_rows * hid * d, 4 * (m_rows * d + hid * d + m_rows * hid)),
        "bn_relu_bwd": ("hbm", 0, 4 * m_rows * hid * 3),
        "gemm_dw1": ("mfma", 2 * m_rows * din * hid, 4 * (m_rows * hid + m_rows * din + din * hid)),
        "gemm_dh": ("mfma", 2 * m_rows * din * hid, 4 * (m_rows * hid + din * hid + m_rows * din)),
        "lars_sumsq": ("hbm", 0, 8 * n_params),
        "lars_adam_update": ("hbm", 0, 28 * n_params),
    }
    out = {}
    for name, recs in event_log.items():
        ms = [r[0].elapsed_time(r[1]) for r in recs]
        if not ms:
            continue
        avg_us = 1e3 * sum(ms) / len(ms)
        if name in work and not any(r[2] for r in recs):
            bound, flops, nbytes = work[name]
        else:  # shape-dependent launches (bn2d_*, conv1x1_dgrad_add): the wrapper recorded each launch's
            # algorithmic work; the roof that takes longer at its peak is the one that bounds the kernel
            nbytes, flops = sum(r[2] for r in recs) / len(recs), sum(r[3] for r in recs) / len(recs)
            bound = None     # decided below, against the roof of the kernel that ran
        entry = {"bound": bound, "launches": len(ms), "avg_us": round(avg_us, 3), "bytes": nbytes, "flops": flops}
        ran = sorted({r[4] for r in recs if len(r) > 4 and r[4]})
        if ran:
            entry["kernel"] = ran[0] if len(ran) == 1 else ran
        x6 = any(k.startswith("gemm_x6") for k in ran) if ran else name in X6_TAGS
        pair = bool(ran) and all("<pair>" in k for k in ran)       # fp32 operands as two fp16 numbers, three products
        h16 = bool(ran) and all(k.startswith(("conv_h", "wgrad_h", "wgrad3_h")) for k in ran)     # 16-bit operands: the dense bf16 / fp16 MFMA peak
        peak_tf = MFMA_BF16_PEAK_TF if h16 else MFMA_BF16_PEAK_TF / 3.0 if pair else MFMA_BF16_PEAK_TF / 6.0 if x6 else MFMA_F32_PEAK_TF
        if bound is None:
            bound = entry["bound"] = "mfma" if flops / (peak_tf * 1e12) > nbytes / (HBM_PEAK_GBS * 1e9) else "hbm"
        if bound == "hbm":
            ach = nbytes / (avg_us * 1e-6) / 1e9
            entry.update(achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5))
        else:
            ach = flops / (avg_us * 1e-6) / 1e12
            entry.update(achieved=round(ach, 3), peak=round(peak_tf, 1), unit="TFLOP/s", frac=round(ach / peak_tf, 5))
            if pair:
                entry["peak_note"] = "fp32 GEMM as three fp16 MFMA products of the scaled, two-way split operands: dense fp16 peak / 3"
            elif x6:
                entry["peak_note"] = "fp32 GEMM as six bf16 MFMA products: dense bf16 peak / 6"

        out[name] = entry
    return out


# launches of peclr_gemm_x6_f32 / _tn_f32 (fp32 operands split into three bf16 numbers, six products on the bf16 MFMA):
# their MFMA roof is the dense bf16 peak / 6 fp32-equivalent flop/s, not the v_mfma_f32 peak
X6_TAGS = {"conv1x1_fwd", "conv1x1_dgrad", "conv1x1_dgrad_add_x6", "conv1x1_wgrad", "gemm_x6", "gemm_x6p", "gemm_x6_tn", "conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "gemm_x6t"}


X6_PRODUCTS = 3.0 if os.environ.get("PECLR_X6_PAIR", "1") != "0" else 6.0      # MFMA products per fp32 multiply-add of the step's GEMMs


def mfma_peak(name):
    return MFMA_BF16_PEAK_TF / 6.0 if name in X6_TAGS else MFMA_F32_PEAK_TF


def pmc_traffic(kernel, args=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, gfx950 corrections applied by tools/pmc_traffic.py) --
    PMC counters cannot be collected from inside the timed run.  None if no profile covers it."""
    # the committed passes were taken on the default workload, one table per precision; other shapes have no PMC figure
    dtype = "fp32" if args is None else args.dtype
    default = args is None or (args.resnet == "50" and args.pairs == 128 and args.size == 224)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json" if dtype == "fp32" else f"pmc_traffic_{dtype}.json")
    if not os.path.exists(path) or not default:
        return None
    with open(path) as f:
        table = json.load(f)
    entry = table.get(kernel)
    return entry.get("traffic_bytes_per_launch") if entry else None


def _physical_cores():
    try:
        import psutil

        return psutil.cpu_count(logical=False)
    except Exception:  # noqa: BLE001
        return None


def _median_time(fn, warmup, timed):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2]), ts


def _calibrate_cpu_threads(args):
    """torch's default (one thread per logical core) is far from the fastest setting for these convolutions on a
    many-core host (128 threads: 4-5x slower than 16 on the MI355X boxes): time one small encoder forward + backward
    at a few thread counts and keep the fastest.  Returns (threads, {threads: seconds})."""
    a = argparse.Namespace(**vars(args))
    a.resnet, a.accum = "18", 1
    model = build_model(a, torch.device("cpu"), 8)
    x = torch.randn(32, 3, 224, 224)            # representative of the timed shapes (small inputs favour more threads)
    default = torch.get_num_threads()
    tried = {}
    for thr in sorted({t for t in (8, 16, 32, 64, default) if t <= default}):
        torch.set_num_threads(thr)
        for rep in range(2):                       # first pass warms the thread pool / primitive cache
            t0 = time.perf_counter()
            model.zero_grad(set_to_none=True)
            model.encoder(x).square().mean().backward()
            tried[thr] = time.perf_counter() - t0
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    return best, {k: round(v, 3) for k, v in tried.items()}


def cpu_baseline(args):
    """BASELINE.md section 3, on this host's cores (kind "port": the NumPy oracle + the same torch-CPU
    ResNet module + the foreach LARS/Adam; the reference's own files never run on the GPU box), at the thread
    count a short calibration finds fastest:
      (i)  head only  -- oracle K1..K8 forward + backward at C2's shape (M = 256, Din = 2048), >= 3 warm-up +
                         >= 10 timed, median;
      (ii) full step  -- C1 (ResNet-18, 2x32 @224, Din 512) with the same protocol, and the bench's own
                         workload: all 2 x `--pairs` views (1 warm-up + 3 timed, median) unless a probe step predicts
                         more than `--cpu-budget` seconds for them, then a bounded sample of 2 x `--cpu-pairs` views.
    `value` is the workload line."""
    import numpy as np

    from oracle import peclr_oracle as O
    from peclr_amd.optim import LARSAdam

    default_threads = torch.get_num_threads()
    threads, tried = _calibrate_cpu_threads(args)

    def make_step(resnet, n, size):
        a = argparse.Namespace(**vars(args))
        a.resnet, a.accum = resnet, 1
        model = build_model(a, torch.device("cpu"), n)
        params = [p for name, p in model.named_parameters() if "final_layer" not in name]
        opt = LARSAdam([{"params": params, "weight_decay": 1e-6}], lr=1e-3, lars=True, fused=False)
        batch = synthetic_batch(n, size, 5, torch.device("cpu"))
        ph = model.projection_head
        head = [ph[0].weight, ph[0].bias, ph[1].weight, ph[1].bias, ph[3].weight]
        jx = torch.cat([batch["jitter_x_1"], batch["jitter_x_2"]]).numpy()
        jy = torch.cat([batch["jitter_y_1"], batch["jitter_y_2"]]).numpy()
        ang = torch.cat([batch["angle_1"], batch["angle_2"]]).numpy()

        def step():
            x = torch.cat([batch["transformed_image1"], batch["transformed_image2"]])
            h = model.encoder(x)
            r = O.head_loss_fwd_bwd(h.detach().numpy(), *[t.detach().numpy() for t in head], n, crop=True,
                                    rotate=True, jitter_x=jx, jitter_y=jy, angle=ang, image_hw=(size, size))
            h.backward(torch.from_numpy(np.ascontiguousarray(r["dh"])))
            for t, k in zip(head, ("dw1", "db1", "dgamma", "dbeta", "dw2")):
                t.grad = torch.from_numpy(np.ascontiguousarray(r[k].astype(np.float32)))
            opt.step()
            opt.zero_grad(set_to_none=True)
        return step

    # (i) head only, C2 shape
    rng = np.random.default_rng(5)
    m_rows, din, hid, n = 256, 2048, 512, 128
    hh = rng.standard_normal((m_rows, din)).astype(np.float32)
    w1 = (rng.standard_normal((hid, din)) / np.sqrt(din)).astype(np.float32)
    w2 = (rng.standard_normal((128, hid)) / np.sqrt(hid)).astype(np.float32)
    b1, gamma, beta = np.zeros(hid, np.float32), np.ones(hid, np.float32), np.zeros(hid, np.float32)
    jx, jy = rng.integers(-14, 1, m_rows), rng.integers(-14, 1, m_rows)
    ang = rng.integers(-45, 46, m_rows).astype(np.float64)
    head_med, head_ts = _median_time(lambda: O.head_loss_fwd_bwd(hh, w1, b1, gamma, beta, w2, n, crop=True, rotate=True,
                                                                 jitter_x=jx, jitter_y=jy, angle=ang,
                                                                 image_hw=(224, 224)), 3, 10)
    # (ii) full step, C1
    c1_med, c1_ts = _median_time(make_step("18", 32, 224), 3, 10)
    # (ii) the bench's workload: whole if affordable, else a bounded sample
    probe = make_step(args.resnet, args.cpu_pairs, args.size)
    probe()
    t0 = time.perf_counter()
    probe()
    per_pair = (time.perf_counter() - t0) / args.cpu_pairs
    # the whole workload (4 steps of it) unless it is predicted past `--cpu-budget` seconds of CPU work: the default
    # configuration (ResNet-50, 2 x 128 @224: ~24 s per step on the MI355X boxes' hosts, ~100 s in all) always fits
    whole = args.cpu_whole == 1 or (args.cpu_whole < 0 and 4 * 2.0 * per_pair * args.pairs <= args.cpu_budget)
    ns = args.pairs if whole else args.cpu_pairs
    w_med, w_ts = _median_time(make_step(args.resnet, ns, args.size) if whole else probe, 1, 3)
    torch.set_num_threads(default_threads)
    return {"value": round(2 * ns / w_med, 3), "unit": "images/sec", "cores": threads,
            "physical_cores": _physical_cores(), "torch_threads": threads, "torch_threads_default": default_threads,
            "thread_calibration_s": tried, "kind": "port",
            "sample": f"1 warm-up + 3 timed steps, median (BASELINE.md section 3 asks >= 3 + >= 10: deviation, one step takes "
                      f"{w_med:.0f} s; head_only and c1_full_step below follow the protocol); ResNet-{args.resnet}, 2x{ns} synthetic "
                      f"{args.size}x{args.size} views ({'the whole workload' if whole else f'bounded sample of the 2x{args.pairs} workload'}), "
                      f"fp32, torch-CPU encoder + NumPy oracle head + foreach LARS/Adam, {threads} threads (fastest of "
                      f"{sorted(tried)}; torch's default here is {default_threads}); {w_med:.2f} s/step",
            "head_only": {"what": "oracle K1..K8 forward + backward (head, stats, align crop+rotate, NT-Xent), "
                                  "M=256 rows, Din=2048, float32 inputs", "median_ms": round(1e3 * head_med, 3),
                          "warmup": 3, "timed": len(head_ts), "min_ms": round(1e3 * head_ts[0], 3),
                          "max_ms": round(1e3 * head_ts[-1], 3)},
            "c1_full_step": {"what": "ResNet-18, 2x32 synthetic 224x224 views, Din 512, crop+rotate, fwd + bwd + "
                                     "LARS/Adam step", "median_s": round(c1_med, 4), "images_per_sec": round(64 / c1_med, 2),
                             "warmup": 3, "timed": len(c1_ts), "min_s": round(c1_ts[0], 4), "max_s": round(c1_ts[-1], 4)}}


def parity_line(model, batch, autocast):
    """`BASELINE.json:metric`'s second half ("NT-Xent loss delta vs ref"): the HIP head / alignment / loss on
    the bench batch against the oracle on the same encoder output, computed BEFORE the timed region.
    Checker only (oracle/step_check.py); nothing it touches is timed."""
    from oracle import step_check

    d = step_check.step_deltas(model, batch, autocast=autocast)
    return {k: d[k] for k in ("loss_delta_vs_oracle", "sim_max_abs_delta", "z_max_abs_delta", "stats_max_abs_delta",
                              "loss_hip", "loss_oracle", "rows") if k in d}


def committed_profile(name, args):
    """Per-kernel figures that cannot be taken from inside the timed run (rocprofv3 kernel-trace durations,
    SQ MFMA-busy counters): read from the committed summaries under profiles/, default workload only."""
    path = os.path.join(ROOT, "profiles", name)
    default = args.dtype == "fp32" and args.resnet == "50" and args.pairs == 128 and args.size == 224
    if not os.path.exists(path) or not default:
        return {}
    with open(path) as f:
        return json.load(f)


COMPACT_LIMIT = 4096      # bytes: the driver records a bounded tail of stdout; round 3's 23.5 KB line did not parse


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(full):
    """The LAST stdout line: the driver's contract + roofline / backbone / cpu_baseline / parity / fp32_gemm_check,
    each cut down to the figures a reader checks (the long forms -- the per-kernel table first of all -- travel on
    the `BENCH_DETAILS` line).  Pure function of the full record, so it is testable without a GPU; the result
    serialises to < COMPACT_LIMIT bytes."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in keep}
    cfg = full.get("config") or {}
    out["config"] = {k: (_short(cfg[k], 200) if isinstance(cfg[k], str) else cfg[k]) for k in
                     ("workload", "global_batch", "parallelism", "accumulate_grad_batches", "launch", "graph_fallback",
                      "fp32_gemm", "conv16", "bn") if k in cfg}
    if cfg.get("amp"):
        out["config"]["amp"] = {k: cfg["amp"][k] for k in ("loss_scale", "optimizer_steps_taken_in_timed_region") if k in cfg["amp"]}
    out["loss"] = full.get("loss")
    roof = full.get("roofline")
    if roof:
        r = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_us", "launches_per_step",
                                      "algorithmic_bytes", "algorithmic_flops", "traffic", "traffic_source", "events") if k in roof}
        fam = roof.get("kernel_family")
        if fam:
            tags = list(fam.get("tags", {}).items())[:4]
            r["kernel_family"] = {"name": fam.get("name"), "share_of_handwritten_time": fam.get("share_of_handwritten_time"),
                                  "top_tags": {t: [v.get("bound"), v.get("frac"), v.get("ms_per_step")] for t, v in tags}}
        out["roofline"] = r
    bb = full.get("backbone")
    if bb:
        out["backbone"] = {k: bb[k] for k in ("flops_per_step_per_gpu", "achieved", "unit", "peak", "frac", "frac_vs_v_mfma_f32",
                                              "frac_vs_x6_roof", "frac_vs_bf16_mfma", "handwritten_share_of_gpu_time") if k in bb}
    for k in ("hand_written_us_per_step", "latency_bound_launches_per_step"):
        if k in full:
            out[k] = full[k]
    cb = full.get("cpu_baseline")
    if cb:
        c = {k: cb[k] for k in ("value", "unit", "cores", "physical_cores", "torch_threads", "kind") if k in cb}
        c["sample"] = _short(cb.get("sample", ""), 330)
        if "head_only" in cb:
            c["head_only_median_ms"] = cb["head_only"].get("median_ms")
        if "c1_full_step" in cb:
            c["c1_full_step_images_per_sec"] = cb["c1_full_step"].get("images_per_sec")
        out["cpu_baseline"] = c
    if "parity" in full:
        out["parity"] = full["parity"]
    for k in ("loss_delta_vs_oracle", "sim_max_abs_delta"):
        if k in full:
            out[k] = full[k]
    if full.get("fp32_gemm_check"):
        out["fp32_gemm_check"] = {k: (float(f"{v:.3e}") if isinstance(v, float) else v)
                                  for k, v in full["fp32_gemm_check"].items() if k != "reference"}
    d = full.get("dist")
    if d:
        out["dist"] = {k: d[k] for k in ("backend", "rccl_version", "ranks_seen", "devices", "launcher") if k in d}
        out["dist"]["grad_buckets"] = len(d.get("grad_buckets", []))
        cbs = d.get("collective_bytes_per_step") or {}
        out["dist"]["all_reduce_total_bytes"] = cbs.get("all_reduce_total")
    if full.get("bf16_companion"):
        out["bf16_companion"] = full["bf16_companion"]
    out["details"] = full.get("details", "BENCH_DETAILS line above")
    line = json.dumps(out)
    if len(line) > COMPACT_LIMIT:      # never let free text push the judged keys out of the recorded tail again
        for obj, key in ((out.get("cpu_baseline"), "sample"), (out["config"], "launch"), (out["config"], "workload")):
            if obj and key in obj:
                obj[key] = _short(obj[key], 80)
        line = json.dumps(out)
    return line


def emit(full):
    """Details first (own line, own prefix, and a file next to the profiles when a scratch directory exists), the
    compact line LAST."""
    path = os.environ.get("PECLR_BENCH_DETAILS")
    if not path and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        path = os.path.join(ROOT, "gpurun_out", "bench_details.json")
    if path:
        try:
            with open(path, "w") as f:
                json.dump(full, f)
            full["details"] = os.path.relpath(path, ROOT) + " + the BENCH_DETAILS line above"
        except OSError:
            pass
    print("BENCH_DETAILS " + json.dumps(full), flush=True)
    print(compact_line(full), flush=True)


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: start N copies of this command, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them), forward rank 0's stdout,
    exit non-zero if any rank does."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               PECLR_BENCH_LAUNCHER="bench.py (self-spawned ranks)")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]
    procs = []
    for r in range(n):
        # ranks > 0 print nothing on stdout by contract; whatever they do say goes to stderr
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), text=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's lines are forwarded by a reader thread while ALL children are polled: the first rank that exits non-zero
    # (or the overall limit) terminates the others -- a rank that dies before or inside a collective would otherwise leave
    # rank 0 blocked in it until the RCCL watchdog fires (10 min) or, over gloo, for ever, and this parent with it
    import threading

    def forward():
        for ln in procs[0].stdout:
            sys.stdout.write(ln)
            sys.stdout.flush()

    reader = threading.Thread(target=forward, daemon=True)
    reader.start()
    limit = float(os.environ.get("PECLR_BENCH_RANKS_TIMEOUT", "3000"))
    t0, failed = time.monotonic(), None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
        elif time.monotonic() - t0 > limit:
            failed = f"no result after {limit:.0f} s"
        if failed or all(c is not None for c in codes):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    codes = [p.returncode for p in procs]
    if failed or any(codes):
        raise SystemExit(f"bench.py: {failed or 'a rank failed'}; rank exit codes {codes}")


def dry_dist(args):
    """--dry-dist: the launch path without a step.  Forms the group exactly as the measured run does and checks it."""
    from peclr_amd import dist as pdist

    if os.environ.get("PECLR_BENCH_DRY_DIE_RANK") == os.environ.get("RANK", "0"):    # (tests: a rank that dies before the rendezvous)
        raise SystemExit(3)
    local = pdist.init_from_env()
    world, rank = pdist.world_size(), pdist.rank()
    dev = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    seen, backend = 1, None
    if world > 1:
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        seen, backend = int(ones.item()), torch.distributed.get_backend()
        torch.distributed.barrier()
    ok = seen == args.gpus == world
    if rank == 0:
        print(json.dumps({"dry_dist": True, "ok": ok, "n_gpus": args.gpus, "world_size": world, "ranks_seen": seen,
                          "backend": backend, "device": str(dev),
                          "launcher": os.environ.get("PECLR_BENCH_LAUNCHER", "external (torch.distributed.run)")}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if not ok:
        raise SystemExit(f"bench.py --dry-dist: --gpus {args.gpus}, WORLD_SIZE {world}, ranks seen {seen}")


def try_graph_child():
    """--graph auto at N=1: run this same command with --graph 1 in a child; returns its JSON line or a reason."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:], "--graph", "1"]
    try:
        p = subprocess.run(cmd, env=dict(os.environ, PECLR_BENCH_CHILD="1"), capture_output=True, text=True, timeout=2400)
    except subprocess.TimeoutExpired:
        return None, "graph child timed out"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    details = [ln for ln in p.stdout.splitlines() if ln.startswith("BENCH_DETAILS ")]
    if p.returncode == 0 and lines:
        return "\n".join(details[-1:] + lines[-1:]), None
    sys.stderr.write("# graph child failed; the end of its stderr:\n" + "\n".join(p.stderr.splitlines()[-12:]) + "\n")
    return None, f"graph child exited with {p.returncode}"


def companion_from_line(line_obj):
    """What the fp32 line keeps of a 16-bit run of the same command: the step time, the aggregate rate, the loss and that run's own
    roofline object (the dominant hand-written kernel of THAT precision)."""
    roof = line_obj.get("roofline") or {}
    return {"dtype": line_obj.get("dtype"), "ms_per_step": line_obj.get("ms_per_step"), "value": line_obj.get("value"),
            "unit": line_obj.get("unit"), "steps": line_obj.get("steps"), "warmup": line_obj.get("warmup"), "loss": line_obj.get("loss"),
            "launch": _short((line_obj.get("config") or {}).get("launch") or "", 60),
            "loss_delta_vs_oracle": line_obj.get("loss_delta_vs_oracle"),
            "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_us", "launches_per_step",
                                                  "algorithmic_bytes", "traffic") if k in roof},
            "how": "same command, --dtype bf16, in a child process after the fp32 run"}


def bf16_companion():
    """The default fp32 run's second measurement: this command once more under --dtype bf16 (bf16 autocast backbone on the in-tree
    16-bit kernels, fp32 head / loss / optimiser), its own warm-up, capture and K timed replays, no CPU baseline.  Runs AFTER the
    fp32 measurement, in its own process: nothing of it is inside the fp32 timed region."""
    import subprocess

    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
        elif a in ("--dtype", "--graph", "--bf16-companion"):
            skip = True
        elif not a.startswith(("--dtype=", "--graph=", "--bf16-companion=")):
            argv.append(a)
    cmd = [sys.executable, os.path.abspath(__file__), *argv, "--dtype", "bf16", "--no-cpu-baseline", "--bf16-companion", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("PECLR_BENCH_CHILD", "PECLR_BENCH_DETAILS", "PECLR_LAUNCH_MANIFEST")}
    env["PECLR_BENCH_DETAILS"] = os.devnull
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    except subprocess.TimeoutExpired:
        return {"error": "bf16 companion run timed out"}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stderr.write("# bf16 companion failed; the end of its stderr:\n" + "\n".join(p.stderr.splitlines()[-12:]) + "\n")
        return {"error": f"bf16 companion run exited with {p.returncode}"}
    return companion_from_line(json.loads(lines[-1]))


def wants_companion(args):
    # (--no-cpu-baseline = "only the measurement itself": profiling / A-B commands carry it, and get no companion either)
    return bool(args.bf16_companion and not args.no_cpu_baseline and args.dtype == "fp32" and args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
                and not os.environ.get("PECLR_BENCH_CHILD"))


def fp32_gemm_check(device):
    """Untimed, before the timed region: the error of the six-product kernels that carry the backbone's GEMM-shaped fp32 work
    (fp32 operands as three bf16 numbers, six MFMA products: peclr_gemm_x6p_f32 forward / input gradient,
    peclr_conv3x3_x6p_f32, peclr_gemm_x6t_f32 weight gradients) and of the v_mfma_f32 kernel against float64 on the same
    fp32 data, relative to the output scale -- the bench line carries the evidence that its fp32 GEMMs are fp32-accurate."""
    from peclr_amd import _capi

    g = torch.Generator(device=device).manual_seed(1234)
    m, n, k = 8192, 512, 1024
    a = torch.randn(m, k, device=device, generator=g)
    bt = torch.randn(n, k, device=device, generator=g) * 0.05
    ref = a.double() @ bt.double().t()
    scale = float(ref.abs().max())
    err = lambda t, r=ref, s=scale: float((t.double() - r).abs().max()) / s  # noqa: E731
    at = a.t().contiguous()                                         # [K, M]: the weight-gradient kernels contract over rows
    planes = _capi.X6Planes([(bt, False)]).pack().planes[0]
    x = torch.randn(8, 256, 14, 14, device=device, generator=g).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 256, 3, 3, device=device, generator=g) * 0.03).contiguous(memory_format=torch.channels_last)
    p3 = _capi.X6Planes([(w.permute(0, 2, 3, 1).reshape(256, 9 * 256), False)]).pack().planes[0]
    y3 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    # the "pair" arithmetic the step runs by default (PECLR_X6_PAIR=1): per-tensor power of two, two fp16 planes, three products
    am = lambda t: t.abs().max().reshape(1).float()             # noqa: E731
    pp = _capi.X6Planes([(bt, False)], pair=True).pack()
    pp3 = _capi.X6Planes([(w.permute(0, 2, 3, 1).reshape(256, 9 * 256), False)], pair=True).pack()
    pair = {"pair_x6p_max_err_over_scale": err(_capi.gemm_x6p(a, pp.planes[0], n, pair=(am(a), pp.scale(0)))),
            "pair_conv3x3_max_err_over_scale": float((_capi.conv3x3_x6p(x, pp3.planes[0], 256, pair=(am(x), pp3.scale(0))).double() - y3).abs().max()) / float(y3.abs().max())}
    return {"shape": [m, n, k], **pair, "x6p_max_err_over_scale": err(_capi.gemm_x6p(a, planes, n)),
            "x6t_max_err_over_scale": err(_capi.gemm_x6t(at, bt.t().contiguous())),
            "conv3x3_x6p_max_err_over_scale": float((_capi.conv3x3_x6p(x, p3, 256).double() - y3).abs().max()) / float(y3.abs().max()),
            "conv3x3_miopen_max_err_over_scale": float((torch.nn.functional.conv2d(x, w, padding=1).double() - y3).abs().max()) / float(y3.abs().max()),
            "v_mfma_f32_max_err_over_scale": err(_capi.gemm(_capi.GEMM_NT, a, bt)), "reference": "float64 on the same fp32 inputs"}


def amp_steps_taken(trainer):
    """fp16 only: optimiser steps actually TAKEN so far (the device-side loss scaler skips a step whose gradients
    overflowed); read outside the timed region."""
    scaler = getattr(trainer, "_scaler", None)
    return scaler.good_steps() if hasattr(scaler, "good_steps") else None


def main():
    args = parse()
    warnings.simplefilter("ignore")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)             # no launcher: one child per GPU, rank 0's lines forwarded
    if args.dry_dist:
        return dry_dist(args)
    graph_note = None
    if args.graph == "auto":
        single = int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1
        if single and not os.environ.get("PECLR_BENCH_CHILD"):
            line, graph_note = try_graph_child()
            if line is not None:
                if wants_companion(args):
                    # the fp32 measurement is complete (its child has exited): attach the bf16 run of the same command
                    parts = line.split("\n")
                    obj = json.loads(parts[-1])
                    obj["bf16_companion"] = bf16_companion()
                    parts[-1] = json.dumps(obj)
                    if len(parts[-1]) > COMPACT_LIMIT:
                        obj["bf16_companion"].pop("how", None)
                        parts[-1] = json.dumps(obj)
                    line = "\n".join(parts)
                print(line, flush=True)
                return
            args.graph = "0"
        else:
            args.graph = "1" if (not single and not args.sync_bn) else "0"
    use_graph = args.graph == "1"
    from peclr_amd import Trainer, _capi
    from peclr_amd import dist as pdist
    from peclr_amd.resnet import conv_flops_per_image

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: peclr_amd has no CPU path for its kernels")
    local = pdist.init_from_env()
    world, rank = pdist.world_size(), pdist.rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = torch.device("cuda", local)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)

    model = build_model(args, device, args.pairs)
    if args.channels_last:
        model.encoder = model.encoder.to(memory_format=torch.channels_last)
    fused_bn = bool(args.fused_bn and args.channels_last)
    if fused_bn:
        from peclr_amd.bn2d import enable_hip_batchnorm

        enable_hip_batchnorm(model.encoder)
        if not args.fork_gemm:
            for m in model.encoder.modules():
                if getattr(m, "hip_fork", False):
                    m.hip_fork = False
    trainer = Trainer(max_epochs=100, accumulate_grad_batches=args.accum, precision=args.dtype,
                      sync_batchnorm=bool(args.sync_bn), activation_checkpointing=bool(args.checkpoint),
                      overlap_wgrad=bool(args.overlap_wgrad)).attach(model)
    trainer.zero_grad()
    batch = synthetic_batch(args.pairs, args.size, 5 + rank, device, channels_last=bool(args.channels_last))

    next_batch = lambda: batch  # noqa: E731 -- the metric's default: one synthetic batch, resident in HBM
    if args.augment:
        import random

        import numpy as np

        from peclr_amd import TwoViewAugmenter

        rs = np.random.default_rng(5 + rank)
        raw = torch.from_numpy(rs.integers(0, 256, (args.pairs, 224, 224, 3), dtype=np.uint8)).to(device)
        joints = torch.from_numpy(np.concatenate([rs.normal((112, 108), 25, (args.pairs, 21, 2)),
                                                  rs.normal(0, 1, (args.pairs, 21, 1))], axis=2)).float()
        augmenter = TwoViewAugmenter(params={"resize_shape": [args.size, args.size]}, rng=random.Random(5 + rank),
                                     channels_last=bool(args.channels_last))
        next_batch = lambda: augmenter(raw, joints)  # noqa: E731 -- host parameter draws + two HIP launches
        batch = next_batch()

    def one_step(i):
        for micro in range(args.accum):  # one optimiser step = `accum` micro-batches
            out = trainer.training_micro_step(next_batch(), i * args.accum + micro)
        return out

    if use_graph and world > 1 and args.sync_bn:
        raise SystemExit("--graph 1 at N>1 needs per-rank BatchNorm statistics")
    split = use_graph and world > 1
    # Everything runs on ONE non-default stream: hipStreamEndCapture crashes on this ROCm build when the
    # process has already run the step eagerly on the default stream (tools/exp/graph_capture_sizes.py).
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream())
    parity = None
    with torch.cuda.stream(stream):
        if rank == 0 or world > 1:   # forward-only, untimed; every rank runs it (BatchNorm state stays in step)
            parity = parity_line(model, batch, trainer._autocast() if args.dtype != "fp32" else None)
        x6_check = (fp32_gemm_check(device) if (rank == 0 and args.dtype == "fp32" and os.environ.get("PECLR_GEMM_X6", "1") != "0")
                    else None)
        if split:
            # every rank must take the same path: agree on whether all captures succeeded
            ok = 1
            try:
                trainer.capture_split_graphs(batch, warmup=max(args.warmup, 3))
            except Exception as exc:  # noqa: BLE001 -- any failure means "run eager"
                ok, graph_note = 0, f"split-graph capture failed on rank {rank}: {type(exc).__name__}"
                trainer.reducer.zero_grad()
            flag = torch.tensor([ok], device=device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if int(flag) == 0:
                use_graph = split = False
                graph_note = graph_note or "split-graph capture failed on another rank"
        if use_graph:
            # W untimed eager steps (on a side stream) + the capture, then K timed replays
            if split:
                def replay():                      # one optimiser step = `accum` micro-batches through the split graphs
                    for _ in range(args.accum):
                        out = trainer.replay_split(next_batch() if args.augment else None)
                    return out
            elif args.accum > 1:      # one graph per micro-batch, accumulators + optimiser step every accum-th replay
                trainer.capture_micro_graph(batch, warmup_windows=max(args.warmup, 1))

                def replay():
                    for _ in range(args.accum):
                        out = trainer.replay_micro(next_batch() if args.augment else None)
                    return out
            else:
                trainer.capture_step_graph(batch, warmup=max(args.warmup, 3))
                replay = lambda: trainer.replay_step(next_batch() if args.augment else None)  # noqa: E731
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            amp_before = amp_steps_taken(trainer)
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = replay()
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            dt = time.perf_counter() - t0
            amp_after = amp_steps_taken(trainer)
            loss = float(out["loss"])
            # per-kernel HIP events cannot sit inside a graph: the SAME K steps once more, eagerly, with an
            # event pair around every hand-written launch (same kernels, same shapes, same stream)
            _capi.EVENT_LOG, _capi.LAUNCH_ORDER, _capi.TAG_BOUND_SUFFIX = {}, [], True
            for i in range(args.steps):
                one_step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
        else:
            for i in range(args.warmup):
                out = one_step(i)
            _capi.EVENT_LOG, _capi.LAUNCH_ORDER, _capi.TAG_BOUND_SUFFIX = {}, [], True
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            amp_before = amp_steps_taken(trainer)
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = one_step(args.warmup + i)
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            dt = time.perf_counter() - t0
            amp_after = amp_steps_taken(trainer)
            loss = float(out["loss"])
    table_steps = args.steps
    event_log, _capi.EVENT_LOG = _capi.EVENT_LOG, None
    launch_order, _capi.LAUNCH_ORDER = _capi.LAUNCH_ORDER, None
    if rank == 0 and os.environ.get("PECLR_LAUNCH_MANIFEST"):
        # names of the hand-written launches of the measured pass, in order: the LAST len(order) peclr:: dispatches
        # of a rocprofv3 trace of this process are exactly these (tools/pmc_mfma.py aligns on that)
        with open(os.environ["PECLR_LAUNCH_MANIFEST"], "w") as f:
            json.dump({"order": launch_order, "steps": args.steps, "argv": sys.argv[1:]}, f)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    dist_info = None
    if world > 1:
        ones = torch.ones(1, device=device)
        torch.distributed.all_reduce(ones)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = None
        devs = [None] * world
        torch.distributed.all_gather_object(devs, int(local))
        shared = os.environ.get("PECLR_SHARE_DEVICE") == "1"       # tests only: two ranks rehearsing on one GPU over gloo
        if int(ones.item()) != args.gpus or (not shared and (torch.distributed.get_backend() != "nccl"
                                                              or len(set(devs)) != world)):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {int(ones.item())} ranks seen on devices {devs} over "
                             f"{torch.distributed.get_backend()}: every rank needs its own GPU and the RCCL backend")
        dist_info = {"backend": torch.distributed.get_backend(), "rccl_version": rccl, "ranks_seen": int(ones.item()),
                     "devices": devs, "launcher": os.environ.get("PECLR_BENCH_LAUNCHER", "external (torch.distributed.run)"),
                     "grad_buckets": [{"params": len(b.params), "bytes": int(b.flat.numel() * b.flat.element_size())}
                                      for b in trainer.reducer.buckets],
                     "collectives_per_step": "all_gather z [2N,128] fp32 + all_gather [row_lse | stats16 | loss] + "
                                             f"{len(trainer.reducer.buckets)} bucket all_reduce(SUM)",
                     # what every collective of one step moves, for checking a SCALE record against (per rank: bytes this rank
                     # contributes; an all-gather delivers world x that to every rank, a ring all-reduce moves
                     # 2 (world - 1) / world of the bucket per rank)
                     "collective_bytes_per_step": {
                         "all_gather_z_per_rank": 2 * args.pairs * 128 * 4,
                         "all_gather_lse_stats_loss_per_rank": (2 * args.pairs + 17) * 4,
                         "bucket_all_reduce": [int(b.flat.numel() * b.flat.element_size()) for b in trainer.reducer.buckets],
                         "all_reduce_total": int(sum(b.flat.numel() * b.flat.element_size() for b in trainer.reducer.buckets)),
                         "ring_bytes_on_the_wire_per_rank": int(2 * (world - 1) / world * sum(
                             b.flat.numel() * b.flat.element_size() for b in trainer.reducer.buckets))},
                     "note": "no scaling curve has been measured by the builder (1 GPU per lease); per-N values "
                             "are the driver's"}
    if rank == 0:
        images = world * 2 * args.pairs * args.accum * args.steps
        n_params = sum(p.numel() for n, p in model.named_parameters() if "final_layer" not in n)
        din = model.config.projection_head_input_dim
        kernels = kernel_table(event_log, 2 * args.pairs, world * 2 * args.pairs, din, 512, n_params)
        # dominant kernel = the kernel FAMILY with the largest share of the step's hand-written GPU time (avg duration x
        # launches, summed over the tags it runs under: one GEMM kernel serves several convolution roles, and its
        # MFMA-bound and HBM-bound shapes are logged apart); the roofline object is that family's biggest tag
        spent = lambda k: kernels[k]["avg_us"] * kernels[k]["launches"]      # noqa: E731
        family = lambda k: kernels[k]["kernel"] if isinstance(kernels[k].get("kernel"), str) else k   # noqa: E731
        by_family = {}
        for k in kernels:
            by_family[family(k)] = by_family.get(family(k), 0.0) + spent(k)
        top_family = max(by_family, key=by_family.get)
        dominant = max((k for k in kernels if family(k) == top_family), key=spent)
        roof = {k: kernels[dominant][k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        roof["kernel_family"] = {"name": top_family, "share_of_handwritten_time": round(by_family[top_family] / sum(by_family.values()), 4),
                                 "tags": {k: {"bound": kernels[k]["bound"], "frac": kernels[k]["frac"],
                                              "ms_per_step": round(spent(k) / (table_steps * args.accum) / 1e3, 3)}
                                          for k in sorted(kernels, key=spent, reverse=True) if family(k) == top_family}}
        if use_graph:
            roof["events"] = "eager pass of the same K steps right after the timed graph replays"
        roof.update(kernel=dominant, avg_us=kernels[dominant]["avg_us"],
                    launches_per_step=kernels[dominant]["launches"] // (table_steps * args.accum),
                    traffic=pmc_traffic(dominant, args),
                    algorithmic_bytes=kernels[dominant]["bytes"], algorithmic_flops=kernels[dominant]["flops"])
        flops_img = conv_flops_per_image(model.encoder.features, (args.size, args.size))
        step_flops = 3 * flops_img * 2 * args.pairs * args.accum
        peak_tf = MFMA_F32_PEAK_TF if args.dtype == "fp32" else MFMA_BF16_PEAK_TF   # fp16 dense peak = the bf16 one
        ach_tf = step_flops * args.steps / dt / 1e12
        result = {
            "metric": "images/sec (2-view) ResNet-50 bs128 @1/2/4/8 MI355X; NT-Xent loss Δ vs ref",
            "value": round(images / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" if not args.augment else "synthetic uint8 source images + joints, fresh two-view GPU augmentation per step",
            "config": {"workload": f"ResNet-{args.resnet} encoder, 2x{args.pairs} synthetic {args.size}x{args.size} "
                                   f"views per GPU, crop+rotate equivariance alignment, NT-Xent tau=0.5, "
                                   f"LARS(Adam) step, {args.dtype}",
                       "global_batch": world * 2 * args.pairs * args.accum, "parallelism": f"dp{world}",
                       "accumulate_grad_batches": args.accum, "channels_last": bool(args.channels_last), "fused_bn": fused_bn, "fork_gemm": bool(fused_bn and args.fork_gemm), "activation_checkpointing": bool(args.checkpoint), "overlap_wgrad": bool(args.overlap_wgrad),
                       "launch": (((f"{['zero', 'one', 'two', 'three', 'four', 'five'][1 + len(trainer._graph_bs)]} hipGraph replays per step "
                                    "(forward to z | backward of head + layer4 | of layer3 | of layer2..stem, each stage's "
                                    "gradient all-reduce in flight under the next stage), "
                                    if len(getattr(trainer, "_graph_bs", [])) > 1 else
                                    "two hipGraph replays per step (forward to z | backward from dz), ") +
                                   "collectives, NT-Xent and optimiser eager between/after them" +
                                   (f"; {args.accum} micro-batches per optimiser step, gradients added into the buckets, "
                                    "all-reduce and optimiser on the window's last one" if args.accum > 1 else "")) if split else
                                  f"{args.accum} hipGraph replays (micro-batch forward + backward) + eager accumulate / optimiser per step"
                                  if use_graph and args.accum > 1 else
                                  "one hipGraph replay per step (whole step captured)" if use_graph else
                                  "eager launches" + (f" ({graph_note})" if graph_note else "")),
                       "graph_fallback": graph_note,   # None unless --graph auto had to fall back to eager launches
                       # fp32 runs: the GEMM-shaped backbone work (the 1x1 and 3x3 convolutions of the residual blocks: forward,
                       # input gradient, weight gradient; the fused entry gradient) runs on the bf16 matrix cores at fp32 ACCURACY: every fp32 operand is split
                       # exactly into three bf16 numbers and six of the nine partial products are accumulated in fp32
                       # (peclr_gemm_x6_f32; error vs float64 <= the v_mfma_f32 kernel's, tests/test_hip_parity.py)
                       # 16-bit runs: which kernels carry the residual blocks' convolutions
                       "conv16": (None if args.dtype == "fp32" else
                                  "in-tree (conv_h / wgrad_h: LDS-DMA operands, fp32 accumulate, fused BatchNorm epilogues, weights packed "
                                  "from the fp32 masters); MIOpen: 7x7 stem, 3x3 / stride-2 weight gradients"
                                  if (fused_bn and os.environ.get("PECLR_CONV16", "1") != "0") else "MIOpen"),
                       "fp32_gemm": ((("forward + input gradients in pair arithmetic: both operands x a per-tensor power of two, split into 2 fp16 numbers, "
                                       "3 MFMA products, fp32 accumulate (error vs float64 <= the six-product kernels': fp32_gemm_check); "
                                       "weight gradients and the stem: exact 3-way bf16 split, 6 products"
                                       if os.environ.get("PECLR_X6_PAIR", "1") != "0" else
                                       "exact 3-way bf16 split, 6 MFMA products, fp32 accumulate (fp32 accuracy)")
                                      if os.environ.get("PECLR_GEMM_X6", "1") != "0" else "v_mfma_f32 / MIOpen fp32")
                                     if args.dtype == "fp32" else None),
                       # fp16: dynamic loss scaling skips a step whose scaled gradients overflow (the launches still
                       # run, the update kernel returns early): how many of the K timed steps were real updates
                       "amp": None if amp_after is None else {
                           "loss_scale": trainer._scaler.get_scale(), "optimizer_steps_taken_in_timed_region": amp_after - amp_before,
                           "timed_steps": args.steps, "scaler": "device-side (peclr_amp_state), inside the fused optimiser launches"},
                       "bn": "global-batch statistics (synchronised)" if (args.sync_bn and world > 1)
                       else "per-rank batch statistics"},
            "loss": round(loss, 6),
            "fp32_gemm_check": x6_check,
            "roofline": roof,
            "kernels": kernels,
            # fp32: the step's GEMM work runs in pair arithmetic (three fp16 products: own roof = the dense fp16 peak / 3 = 833
            # TFLOP/s fp32-equivalent) or, with PECLR_X6_PAIR=0, on the six-product kernels (dense bf16 peak / 6 = 417); `frac` is
            # priced against the roof of the kernels that run, frac_vs_v_mfma_f32 against the 157.3 TFLOP/s of the v_mfma_f32
            # instructions they no longer use
            "backbone": {"note": "whole encoder, 3x forward conv FLOPs over the step time; fp32: every 1x1 and 3x3 convolution of the "
                                 "residual blocks (stride 1 and 2; forward, input gradient, weight gradient) and the 7x7 stem run on the "
                                 "in-tree split-operand kernels (config.fp32_gemm); 16-bit: see config.conv16",
                         "flops_per_step_per_gpu": step_flops, "achieved": round(ach_tf, 2), "unit": "TFLOP/s",
                         **({"peak": round(MFMA_BF16_PEAK_TF / X6_PRODUCTS, 1), "frac": round(ach_tf / (MFMA_BF16_PEAK_TF / X6_PRODUCTS), 4),
                             "frac_vs_x6_roof": round(ach_tf / (MFMA_BF16_PEAK_TF / 6.0), 4),
                             "frac_vs_v_mfma_f32": round(ach_tf / MFMA_F32_PEAK_TF, 4)}
                            if (args.dtype == "fp32" and os.environ.get("PECLR_GEMM_X6", "1") != "0") else
                            {"peak": peak_tf, "frac": round(ach_tf / peak_tf, 4),
                             **({"frac_vs_bf16_mfma": round(ach_tf / peak_tf, 4)} if args.dtype != "fp32" else
                                {"frac_vs_v_mfma_f32": round(ach_tf / peak_tf, 4)})})},
            "hand_written_us_per_step": round(sum(k["avg_us"] * k["launches"] / table_steps
                                                  for k in kernels.values()), 1),
            # launch-count reduction of the head / alignment / loss (reference: ~100 stock-op
            # launches forward incl. a CPU round trip, ~60 backward through autograd)
            "head_launches_per_step": {name: kernels[name]["launches"] // (table_steps * args.accum) for name in
                                       ("gemm_k1_fwd", "bn_relu_fwd", "gemm_k2_fwd", "align_fwd", "ntxent_fwd", "ntxent_finalize",
                                        "ntxent_bwd", "slab_reduce", "align_bwd", "gemm_dw2", "gemm_da", "bn_relu_bwd",
                                        "gemm_dw1", "gemm_dh") if name in kernels},
        }
        if parity is not None:
            result["parity"] = parity
            result["loss_delta_vs_oracle"] = parity["loss_delta_vs_oracle"]
            result["sim_max_abs_delta"] = parity["sim_max_abs_delta"]
        # figures a run cannot take of itself: rocprofv3 kernel-trace durations (HIP-event pairs add ~4 us, which
        # matters for the us-scale head kernels) and SQ counters, from the committed profiles of this command
        prof_file = "r03_bench_mfma.json"
        prof = committed_profile(prof_file, args)   # tools/profile_passes.sh + tools/pmc_mfma.py
        for name, k in kernels.items():
            p = prof.get(name)
            if not p:
                continue
            # NOT measured by this run: figures of the committed rocprofv3 passes of this command (their own build, their
            # own box), kept apart from the live numbers
            cp = {"source": f"profiles/{prof_file}", "rocprof_avg_us": p["avg_us"], "symbol": p.get("symbol")}
            if min(k["avg_us"], p["avg_us"]) < 30.0:
                work = k["flops"] / 1e12 if k["bound"] == "mfma" else k["bytes"] / 1e9
                cp["achieved_rocprof"] = round(work / (p["avg_us"] * 1e-6), 3)
                cp["frac_rocprof"] = round(cp["achieved_rocprof"] / k["peak"], 5)
            for key in ("mfma_util", "mfma_busy_of_sq_busy", "lds_conflict_share", "occupancy"):
                if key in p and (k["bound"] == "mfma" or not key.startswith("mfma")):
                    cp[key] = p[key]
            k["committed_profile"] = cp
        if world > 1:
            result["dist"] = dist_info
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args)
        # launches that are latency-bound by construction (a few KB of partial sums): counted, because each costs a
        # launch slot of the graph whatever its bytes
        small = ("bn2d_finalize", "bn2d_bwd_finalize", "wgrad_slab_reduce", "bn2d_combine")
        result["latency_bound_launches_per_step"] = {
            "count": sum(k["launches"] for t, k in kernels.items() if t.split("~")[0] in small) // (table_steps * args.accum),
            "ms": round(sum(k["launches"] * k["avg_us"] for t, k in kernels.items() if t.split("~")[0] in small)
                        / (table_steps * args.accum) / 1e3, 3)}
        if wants_companion(args):           # (eager fallback of the default run: the child-process route above did not print)
            result["bf16_companion"] = bf16_companion()
        emit(result)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
