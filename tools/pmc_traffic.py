#!/usr/bin/env python
"""HBM traffic per launch of one kernel from two rocprofv3 PMC passes (rocpd sqlite databases).

MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE come from the L2's fabric-side request
counters, are reported in KiB, need SEPARATE passes (TCC slots), and on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced streaming read -> doubled here; WRITE_SIZE is taken as is
(uncalibrated).  Usage:
    python tools/pmc_traffic.py <fetch_pass.db> <write_pass.db> <kernel-substring> [out.json]
"""
import json
import sqlite3
import sys


def per_launch(db, counter, needle):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, value from counters_collection").fetchall()
    vals = [v for n, cn, v in rows if needle in n and cn == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


# bench.py kernel name -> substring of the HIP kernel symbol
BENCH_KERNELS = {
    "gemm_k1_fwd": None, "bn_relu_fwd": "bn_relu_fwd_kernel", "align_fwd": "align_fwd_kernel",
    "ntxent_fwd": "ntxent_kernel<false>", "ntxent_finalize": "ntxent_finalize_kernel",
    "ntxent_bwd": "ntxent_kernel<true>", "slab_reduce": "slab_reduce_kernel", "align_bwd": "align_bwd_kernel",
    "bn_relu_bwd": "bn_relu_bwd_kernel", "lars_sumsq": "sumsq_kernel", "lars_adam_update": "lars_adam_kernel",
    "bn2d_stats": "bn2d_stats_kernel", "bn2d_finalize": "bn2d_stats_finalize_kernel", "bn2d_apply": "bn2d_apply_kernel",
    "bn2d_bwd_reduce": "bn2d_bwd_reduce_kernel", "bn2d_bwd_finalize": "bn2d_bwd_finalize_kernel",
    "bn2d_bwd_apply": "bn2d_bwd_apply_kernel",
    "bn2d_pool_apply": "bn2d_pool_apply_kernel", "bn2d_pool_bwd_reduce": "bn2d_pool_bwd_reduce_kernel",
    "bn2d_pool_bwd_apply": "bn2d_pool_bwd_apply_kernel", "bn2d_apply_avgpool": "bn2d_apply_avgpool_kernel",
    "conv1x1_dgrad_add": "gemm_f32_nn128_kernel",
}


def table(fetch_db, write_db, out_path):
    out = {}
    for bench_name, needle in BENCH_KERNELS.items():
        if needle is None:
            continue
        f, nf = per_launch(fetch_db, "FETCH_SIZE", needle)
        w, nw = per_launch(write_db, "WRITE_SIZE", needle)
        if f is None or w is None:
            continue
        out[bench_name] = {"symbol": needle, "launches_sampled": [nf, nw], "FETCH_SIZE_KiB_raw": round(f, 2),
                           "WRITE_SIZE_KiB_raw": round(w, 2),
                           "traffic_bytes_per_launch": round((2 * f + w) * 1024)}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; per-launch means; "
                    "FETCH_SIZE doubled (gfx950 tallies wide coalesced reads at half), WRITE_SIZE as reported "
                    "(MI355X_MICROARCH.md section HBM)")
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1] == "--table":
        return table(sys.argv[2], sys.argv[3], sys.argv[4])
    fetch_db, write_db, needle = sys.argv[1:4]
    fetch_kib, nf = per_launch(fetch_db, "FETCH_SIZE", needle)
    write_kib, nw = per_launch(write_db, "WRITE_SIZE", needle)
    out = {"kernel": needle, "launches_sampled": [nf, nw],
           "FETCH_SIZE_KiB_per_launch_raw": fetch_kib, "WRITE_SIZE_KiB_per_launch_raw": write_kib,
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads are tallied at half), WRITE_SIZE x1",
           "traffic_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024 if fetch_kib is not None and write_kib is not None else None}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
