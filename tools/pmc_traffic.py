#!/usr/bin/env python
"""HBM traffic per launch of one kernel from two rocprofv3 PMC passes (rocpd sqlite databases).

MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE come from the L2's fabric-side request
counters, are reported in KiB, need SEPARATE passes (TCC slots), and on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced streaming read -> doubled here; WRITE_SIZE is taken as is
(uncalibrated).  Usage:
    python tools/pmc_traffic.py <fetch_pass.db> <write_pass.db> <kernel-substring> [out.json]
    python tools/pmc_traffic.py --table <fetch_pass.db> <write_pass.db> <manifest.json> <out.json>     (every bench tag)
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


def table(fetch_db, write_db, manifest, out_path):
    """Per bench tag: the FETCH / WRITE counters of exactly the launches the bench timed under that tag (both passes are
    aligned with the launch manifest, as tools/pmc_mfma.py does), so `traffic` and the bench's `algorithmic_bytes`
    describe the same launches.  Keyed by tag; `symbol` is the kernel that ran."""
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from pmc_mfma import align

    order = json.load(open(manifest))["order"]
    f, w = align(fetch_db, order), align(write_db, order)
    out = {}
    for tag in f:
        fv, wv = f[tag]["counters"].get("FETCH_SIZE"), w.get(tag, {}).get("counters", {}).get("WRITE_SIZE")
        if not fv or not wv:
            continue
        fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
        out[tag] = {"symbol": f[tag]["symbol"], "launches_sampled": [len(fv), len(wv)], "FETCH_SIZE_KiB_raw": round(fk, 2),
                    "WRITE_SIZE_KiB_raw": round(wk, 2), "traffic_bytes_per_launch": round((2 * fk + wk) * 1024)}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --graph 0; per-launch means over "
                    "the launches of each bench tag (launch manifest); FETCH_SIZE doubled (gfx950 tallies wide coalesced reads "
                    "at half), WRITE_SIZE as reported (MI355X_MICROARCH.md section HBM)")
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1] == "--table":
        return table(*sys.argv[2:6])
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
