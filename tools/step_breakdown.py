#!/usr/bin/env python
"""Where a step's GPU time goes: every dispatch between the first and the last hand-written launch of the measured
pass (located through the launch manifest, as in tools/pmc_mfma.py) grouped into categories, per step.

    python tools/step_breakdown.py <kernel-trace results.db> <manifest.json> [steps]
"""
import json
import sqlite3
import sys
from collections import OrderedDict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_mfma import EXPECT  # noqa: E402

CATS = OrderedDict([
    ("MIOpen / CK convolution forward", lambda n: ("igemm_fwd" in n or "conv_fwd" in n or "ConvFwd" in n or "grouped_conv_fwd" in n)),
    ("MIOpen / CK convolution input gradient", lambda n: ("igemm_bwd" in n or "conv_bwd_data" in n or "ConvBwdData" in n)),
    ("MIOpen / CK convolution weight gradient", lambda n: ("igemm_wrw" in n or "conv_bwd_weight" in n or "ConvBwdWeight" in n
                                                          or "kernel_batched_gemm_xdl" in n)),   # CK's wrw-as-batched-GEMM solver
    ("MIOpen zero-fill / cast for split-K weight gradients", lambda n: "SubTensorOp" in n or "fillBufferAligned" in n),
    ("hand-written: 16-bit convolutions (conv_h: forward, input gradient, fused entry gradient; wgrad_h / wgrad3_h: weight gradients; h_pack)",
     lambda n: "peclr" in n and any(k in n for k in ("conv_h_kernel", "wgrad_h_kernel", "wgrad3_h_kernel", "h_pack_kernel"))),
    ("hand-written: 7x7 stem forward + filter pack (stem_fwd_kernel, stem_pack_kernel)", lambda n: "peclr" in n and "stem_" in n),
    ("hand-written: BatchNorm2d glue (bn2d_*)", lambda n: "peclr" in n and "bn2d_" in n),
    ("hand-written: fused dgrad + residual GEMM (128x128)", lambda n: "peclr" in n and "gemm_f32_nn128" in n),
    ("hand-written: 3x3 convolutions on the bf16 matrix cores (forward, input gradient: gemm_x6p <.., 9>; weight gradient: gemm_x6w)",
     lambda n: "peclr" in n and ("gemm_x6w" in n or "wgrad_x6r" in n or ("gemm_x6p_kernel" in n and ", 9, " in n))),
    ("hand-written: fp32 GEMMs on the bf16 matrix cores (1x1 convolutions, fused dgrad, weight gradients)", lambda n: "peclr" in n and "gemm_x6" in n),
    ("hand-written: head GEMMs / bf16 GEMM", lambda n: "peclr" in n and "gemm_" in n),
    ("hand-written: BN1d+ReLU, align, NT-Xent", lambda n: "peclr" in n and any(k in n for k in ("bn_relu", "align_", "ntxent", "slab_reduce"))),
    ("hand-written: LARS / Adam", lambda n: "peclr" in n and ("sumsq" in n or "lars_adam" in n)),
    ("hand-written: other", lambda n: "peclr" in n),
    ("ATen / other", lambda n: True),
])


def main():
    db, manifest = sys.argv[1:3]
    order = json.load(open(manifest))
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else int(order.get("steps", 1))
    order = order["order"]
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, name, end - start from kernels order by dispatch_id").fetchall()
    ours = [r for r in rows if "peclr" in r[1] and "fold_partials" not in r[1]]   # (the fold launch in front of a finalize is not a manifest entry of its own)
    wants = [EXPECT.get(name.split("::")[-1]) for name in order]
    for off in range(len(ours) - len(order) + 1):
        lo = len(ours) - len(order) - off
        if all(w is None or (any(x in d[1] for x in w) if isinstance(w, tuple) else w in d[1]) for w, d in zip(wants, ours[lo:lo + len(order)])):
            first, last = ours[lo][0], ours[lo + len(order) - 1][0]
            break
    else:
        lo = len(ours) - len(order)
        bad = [(k, order[k], ours[lo + k][1][:90]) for k, w in enumerate(wants)
               if w is not None and not (any(x in ours[lo + k][1] for x in w) if isinstance(w, tuple) else w in ours[lo + k][1])][:5]
        raise SystemExit(f"measured pass not found: {len(ours)} peclr:: dispatches, manifest {len(order)}; first mismatches "
                         f"when aligned at the end: {bad}")
    tot = OrderedDict((k, [0.0, 0]) for k in CATS)
    for did, name, dur in rows:
        if first <= did <= last:
            for k, pred in CATS.items():
                if pred(name):
                    tot[k][0] += dur / 1e6
                    tot[k][1] += 1
                    break
    total = sum(v[0] for v in tot.values())
    print(f"# {db}: {steps} measured step(s), dispatches {first}..{last}; GPU time {total / steps:.2f} ms per step (sum of kernel durations)")
    print(f"{'ms/step':>9} {'share':>7} {'launches/step':>14}  category")
    for k, (ms, n) in tot.items():
        print(f"{ms / steps:9.3f} {100 * ms / total:6.1f}% {n / steps:14.1f}  {k}")


if __name__ == "__main__":
    main()
