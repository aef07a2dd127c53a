#!/usr/bin/env python
"""Per-kernel summary (count / avg / min / max / share) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on this image)."""
import sqlite3
import sys


def main(path, top=60, flt=None):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                     "from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, total kernel time {tot/1e6:.3f} ms")
    print(f"{'share%':>7} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    for r in rows:
        if flt and flt not in r[0]:
            continue
        print(f"{r[5]/tot*100:7.2f} {r[1]:6d} {r[2]/1e3:10.2f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f}  {r[0][:150]}")
        top -= 1
        if top == 0:
            break


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60, sys.argv[3] if len(sys.argv) > 3 else None)
