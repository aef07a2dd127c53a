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


def main():
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
