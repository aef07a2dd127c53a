#!/usr/bin/env python
"""Attribute the dispatches of a rocprofv3 run (rocpd sqlite database) to the bench's kernel names and
summarise durations and SQ counters per name.

    python tools/pmc_mfma.py <results.db> <manifest.json> <out.json> [<more results.db of other --pmc passes> ...]

The manifest (bench.py / tools/mfma_probe.py with PECLR_LAUNCH_MANIFEST set) lists the names of the
hand-written launches of the measured pass in launch order; they are the LAST len(order) dispatches whose
symbol lives in namespace `peclr::` (nothing of ours is launched after the measured pass).  One C-ABI
entry point = one launch, so the alignment is one to one; it is verified name against symbol.

Counters (each pass: rocprofv3 --kernel-trace --pmc <<= 8 SQ counters> [GRBM_GUI_ACTIVE]); rocprofv3 reports
every counter SUMMED over its instances (SQ_*: 8 XCC x 4 shader engines, GRBM_GUI_ACTIVE: 8 XCC), and
`rocprofv3 -L` on this image defines MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM)
and OccupancyPercent = 400 * sum(SQ_WAVE_CYCLES) / max(GRBM_GUI_ACTIVE) / CU_NUM / 32:
  mfma_util             = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)    (the MfmaUtil formula;
                          cross-check: v_mfma_f32_32x32x2_f32 books 64 busy cycles, so the counter equals
                          64 * flops / 4096 for the fp32 GEMMs -- it does, to the digit)
  mfma_busy_of_sq_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 * 1024)          (same, over the time the shader
                          engines have any wave resident instead of the whole dispatch window)
  lds_conflict_share    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                       (extra cycles / LDS-array cycles)
  waves_per_cu          = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs)             (SQ_WAVE_CYCLES counts quad-cycles)
  occupancy             = waves_per_cu / 32 wave slots                                  (OccupancyPercent / 100)
  shader_clock_ghz      = (GRBM_GUI_ACTIVE/8) / duration                                 (profiled passes clock lower than
                          un-profiled runs: compare mfma_util with flops-based fractions at THIS clock)
"""
import json
import sqlite3
import sys

# bench name (or its stem after '::') -> substring the kernel symbol must contain
EXPECT = {
    "gemm_k1_fwd": "gemm_f32_", "gemm_k2_fwd": "gemm_f32_", "gemm_dw1": "gemm_f32_",
    "gemm_dw2": "gemm_f32_", "gemm_da": "gemm_f32_", "gemm_dh": "gemm_f32_",
    # (convolution tags: the fp32 six-product kernels, or -- 16-bit runs -- conv_h / wgrad_h kernels)
    "conv1x1_dgrad_add": ("gemm_", "conv_h_kernel"), "conv1x1_dgrad_add_x6": "gemm_x6", "conv1x1_fwd": ("gemm_x6", "conv_h_kernel"),
    "conv1x1_dgrad": ("gemm_x6", "conv_h_kernel"), "conv1x1_wgrad": ("gemm_x6", "wgrad_h_kernel"), "conv3x3_fwd": ("gemm_x6p", "conv_h_kernel"),
    "conv3x3_dgrad": ("gemm_x6p", "conv_h_kernel"), "conv3x3_wgrad": ("gemm_x6w", "wgrad3_h_kernel", "wgrad_x6r_kernel"), "gemm_x6p": "gemm_x6p",
    "gemm_x6t": "gemm_x6t", "conv3x3_x6p": "gemm_x6p", "conv_s2_fwd": ("gemm_x6p", "conv_h_kernel"), "conv_s2_dgrad": ("gemm_x6p", "conv_h_kernel"),
    "conv3x3_s2_dgrad": ("gemm_x6p", "conv_h_kernel"), "conv_s2_x6p": "gemm_x6p", "wgrad_slab_reduce": "slab_reduce", "x6_pack": ("x6_pack_kernel", "x6_pair_kernel"),
    "x6_absmax": ("x6_absmax_kernel", "x6_pair_kernel<false>"),
    "h_pack": "h_pack_kernel", "bn_relu_fwd": "bn_relu_fwd_kernel", "bn_relu_bwd": "bn_relu_bwd_kernel",
    "align_fwd": "align_fwd_kernel", "align_bwd": "align_bwd_kernel", "ntxent_fwd": "ntxent_kernel<false>",
    "ntxent_bwd": "ntxent_kernel<true>", "ntxent_finalize": "ntxent_finalize_kernel", "slab_reduce": "slab_reduce",
    "lars_sumsq": "sumsq_kernel", "lars_adam_update": "lars_adam_kernel", "bn2d_stats": "bn2d_stats_kernel",
    "bn2d_finalize": "finalize", "bn2d_apply": "bn2d_apply_kernel", "bn2d_bwd_reduce": "bn2d_bwd_reduce_kernel",
    "bn2d_bwd_finalize": "finalize", "bn2d_bwd_apply": "bn2d_bwd_apply_kernel", "bn2d_pool_apply": "bn2d_pool_apply_kernel",
    "bn2d_pool_bwd_reduce": "bn2d_pool_bwd_reduce_kernel", "bn2d_pool_bwd_apply": ("bn2d_pool_bwd_apply_kernel", "bn2d_pool_bwd_apply16_kernel"),
    "bn2d_apply_avgpool": "bn2d_apply_avgpool_kernel", "bn2d_bwd_reduce_avgpool": "bn2d_bwd_reduce_kernel",
    "bn2d_bwd_apply_avgpool": "bn2d_bwd_apply_kernel",
}
CUS, SIMD_NUM, XCCS, SES = 256, 1024, 8, 32


def dispatches(db):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, "
                     "accum_vgpr_count, lds_size from kernels order by dispatch_id").fetchall()
    ours = [r for r in rows if "peclr" in r[1] and "fold_partials" not in r[1]]   # (the fold launch in front of a finalize is not a manifest entry of its own)
    counters = {}
    try:
        for did, cname, val in c.execute("select dispatch_id, counter_name, value from counters_collection"):
            counters.setdefault(did, {})[cname] = val
    except sqlite3.OperationalError:
        pass
    return ours, counters


def align(db, order):
    ours, counters = dispatches(db)
    if len(ours) < len(order):
        raise SystemExit(f"{db}: {len(ours)} peclr:: dispatches < {len(order)} manifest entries")
    wants = [EXPECT.get(name.split("::")[-1].split("~")[0]) for name in order]     # ("~hbm": bench.py's suffix for HBM-bound GEMM launches)
    # the measured pass is the LAST run of dispatches whose symbols match the manifest entry by entry (a few
    # launches of ours may follow it, e.g. the FLOP counter's eval-mode forward in bench.py)
    for off in range(len(ours) - len(order) + 1):
        lo = len(ours) - len(order) - off
        if all(w is None or (any(x in d[1] for x in w) if isinstance(w, tuple) else w in d[1]) for w, d in zip(wants, ours[lo:lo + len(order)])):
            tail = ours[lo:lo + len(order)]
            break
    else:
        raise SystemExit(f"{db}: no run of {len(order)} consecutive peclr:: dispatches matches the manifest")
    per = {}
    for name, d in zip(order, tail):
        e = per.setdefault(name, {"symbol": d[1].split("(")[0][-110:], "us": [], "grid": [d[4] // max(d[7], 1), d[5], d[6]],
                                  "workgroup": d[7], "vgpr": d[8], "agpr": d[9], "lds_bytes": d[10], "counters": {}})
        e["us"].append((d[3] - d[2]) / 1e3)
        for cname, val in counters.get(d[0], {}).items():
            e["counters"].setdefault(cname, []).append(val)
    return per


def main():
    db, manifest, out = sys.argv[1:4]
    extra = sys.argv[4:]
    order = json.load(open(manifest))["order"]
    merged = None
    for path in [db] + extra:
        per = align(path, order)
        if merged is None:
            merged = per
        else:
            for name, e in per.items():
                merged[name]["counters"].update(e["counters"])
    res = {}
    for name, e in merged.items():
        us = sorted(e["us"])
        c = {k: sum(v) / len(v) for k, v in e["counters"].items()}
        r = {"symbol": e["symbol"], "launches": len(us), "avg_us": round(sum(us) / len(us), 3), "min_us": round(us[0], 3),
             "grid_workgroups": e["grid"], "workgroup": e["workgroup"], "vgpr": e["vgpr"], "agpr": e["agpr"],
             "lds_bytes": e["lds_bytes"], "counters_per_launch": {k: round(v, 1) for k, v in c.items()}}
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if c.get("SQ_BUSY_CYCLES"):
            r["mfma_busy_of_sq_busy"] = round(mfma / (c["SQ_BUSY_CYCLES"] / SES * SIMD_NUM), 4)
        if c.get("GRBM_GUI_ACTIVE"):
            gui = c["GRBM_GUI_ACTIVE"] / XCCS          # per-XCC mean, standing in for reduce(.., max)
            r["mfma_util"] = round(mfma / (gui * SIMD_NUM), 4)
            if r["avg_us"] >= 50.0:   # on us-scale kernels the GRBM window is mostly dispatch overhead
                r["shader_clock_ghz"] = round(gui / (r["avg_us"] * 1e3), 3)
            if "SQ_WAVE_CYCLES" in c:
                r["waves_per_cu"] = round(4 * c["SQ_WAVE_CYCLES"] / (gui * CUS), 2)
                r["occupancy"] = round(r["waves_per_cu"] / 32, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        res[name] = r
    res["_note"] = ("durations: rocprofv3 kernel trace (end - start); counters: per-launch means of separate --pmc passes "
                    "of the same command, attributed through the launch manifest; formulas in tools/pmc_mfma.py")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for name, r in res.items():
        if name != "_note":
            print(f"{name:48s} {r['avg_us']:9.2f} us  x{r['launches']:<4d} mfma/sq_busy={r.get('mfma_busy_of_sq_busy')} "
                  f"util={r.get('mfma_util')} clk={r.get('shader_clock_ghz')} lds_conf={r.get('lds_conflict_share')} waves/cu={r.get('waves_per_cu')}")


if __name__ == "__main__":
    main()
