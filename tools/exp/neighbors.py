"""Who launches the small kernels?  For every dispatch whose name contains <pattern>, count the (previous, next)
kernel names in stream order (rocpd database of a rocprofv3 --kernel-trace run)."""
import collections
import sqlite3
import sys


def short(n):
    for key in ("igemm_wrw", "igemm_fwd", "igemm_bwd", "batched_gemm", "grouped_conv_fwd", "grouped_conv_bwd_data", "grouped_conv_bwd_weight",
                "bn2d_", "gemm_h_nt128", "gemm_bf16", "SubTensorOpWithScalar", "SubTensorOpWithCast", "bfloat16_copy", "bfloat16tofloat32",
                "copyBuffer", "fillBuffer", "FillFunctor", "sumsq", "lars_adam", "transpose", "elementwise"):
        if key in n:
            return key
    return n[:50]


def main(path, pattern):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    names = [short(r[0]) for r in rows]
    pairs = collections.Counter()
    dur = collections.defaultdict(float)
    for i, r in enumerate(rows):
        if pattern in r[0]:
            key = (names[i - 1] if i else "-", names[i + 1] if i + 1 < len(rows) else "-")
            pairs[key] += 1
            dur[key] += (r[2] - r[1]) / 1e3
    for key, n in pairs.most_common(15):
        print(f"{n:6d}  {dur[key] / n:8.2f} us   prev={key[0]:28s} next={key[1]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
