"""The re-scheduled weight-gradient loops (gemm_x6w2_kernel, gemm_x6t2_kernel: round 6) against the first forms they replace.

Both compute the same products in the same order per accumulator, so the results must agree BIT FOR BIT -- on ragged shapes as
well (columns that do not exist, a last k-step past the tensor, images whose pixel count is not a multiple of 4, several
slabs).  The library reads its switches (PECLR_X6W2 / PECLR_X6T2) once per process, so the first forms run in a child process on
the same seeded inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

# (rows K, M, N) of 1x1 weight gradients: every tile shape of pick_tile, ragged M / N / K, one and many slabs
ONE = [(50176, 128, 512), (40000 + 36, 256, 64), (8192 + 4, 132, 260), (3000, 512, 128), (30000 + 4, 64, 256), (30000, 64, 64),
       (9000, 60, 132), (20000 + 12, 128, 64), (10000, 36, 300), (20000, 128, 128), (1000 + 8, 128, 256)]
# (images, channels out, channels in, H, W) of 3x3 / stride-1 weight gradients: both workgroup shapes, H W % 4 != 0, W != H
NINE = [(6, 128, 128, 9, 9), (4, 256, 256, 7, 7), (16, 256, 256, 14, 14), (5, 64, 64, 10, 10), (3, 36, 36, 11, 11), (2, 64, 64, 56, 56),
        (3, 132, 68, 7, 9), (7, 64, 128, 6, 13), (2, 192, 64, 28, 28)]
# (images, channels out, channels in, H out, W out) of 3x3 / stride-2 weight gradients (X has 2H x 2W pixels)
NINE_S2 = [(6, 128, 128, 9, 9), (3, 64, 64, 14, 14), (4, 256, 256, 7, 7), (3, 132, 68, 7, 6), (9, 64, 36, 10, 10), (2, 128, 128, 28, 28), (5, 512, 512, 7, 7)]


def _inputs(seed, k, m, n):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(k, m, generator=g).to(DEV), torch.randn(k, n, generator=g).to(DEV)


def _all_results():
    from peclr_amd import _capi

    out = {}
    for i, (k, m, n) in enumerate(ONE):
        a, b = _inputs(100 + i, k, m, n)
        out[f"one{i}"] = _capi.gemm_x6t(a, b).cpu().numpy()
    for i, (nb, co, ci, h, w) in enumerate(NINE):
        a, b = _inputs(200 + i, nb * h * w, co, ci)
        out[f"nine{i}"] = _capi.gemm_x6t(a, b, taps=9, hw=(h, w)).cpu().numpy()
    for i, (nb, co, ci, h, w) in enumerate(NINE_S2):
        g = torch.Generator().manual_seed(300 + i)
        a, b = torch.randn(nb * h * w, co, generator=g).to(DEV), torch.randn(4 * nb * h * w, ci, generator=g).to(DEV)
        out[f"nine_s2_{i}"] = _capi.gemm_x6t(a, b, taps=9, hw=(h, w), stride=2).cpu().numpy()
    return out


if __name__ == "__main__":            # the child: first forms (environment set by the parent), results to the file named in argv
    sys.path.insert(0, ROOT)
    np.savez(sys.argv[1], **_all_results())
    sys.exit(0)


def test_rescheduled_weight_gradient_loops_reproduce_the_first_forms_bit_for_bit(tmp_path):
    assert os.environ.get("PECLR_X6W2", "1") != "0" and os.environ.get("PECLR_X6T2", "1") != "0", "this process must run the re-scheduled loops"
    new = _all_results()
    path = str(tmp_path / "first_forms.npz")
    env = dict(os.environ, PECLR_X6W2="0", PECLR_X6T2="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env=env, cwd=ROOT, timeout=900)
    old = np.load(path)
    assert sorted(old.files) == sorted(new)
    for name in new:
        assert new[name].shape == old[name].shape and np.isfinite(new[name]).all(), name
        assert np.array_equal(new[name].view(np.uint32), old[name].view(np.uint32)), name


def test_rescheduled_loops_leave_no_trace_of_rows_they_must_not_read():
    """Rows before the tensor, past its end and outside an image are out-of-range buffer offsets that read as zeros: a tensor
    surrounded by NaNs in memory must give the same weight gradient as the tensor alone (K ragged, so that the last k-step hangs
    over the end; the 3x3 taps of the first and last image reach before / past the tensor)."""
    from peclr_amd import _capi

    nb, c, h, w = 3, 64, 7, 9
    k = nb * h * w
    g = torch.Generator().manual_seed(7)
    pad = 4096
    for taps, hw, stride in ((9, (h, w), 1), (1, None, 1), (9, (h, w), 2)):
        kb = k * stride * stride
        big_a = torch.full((k + 2 * pad, c), float("nan"), device=DEV)
        big_b = torch.full((kb + 2 * pad, c), float("nan"), device=DEV)
        a, b = torch.randn(k, c, generator=g).to(DEV), torch.randn(kb, c, generator=g).to(DEV)
        big_a[pad:pad + k] = a
        big_b[pad:pad + kb] = b
        inside = _capi.gemm_x6t(big_a[pad:pad + k], big_b[pad:pad + kb], taps=taps, hw=hw, stride=stride)
        alone = _capi.gemm_x6t(a, b, taps=taps, hw=hw, stride=stride)
        assert torch.isfinite(inside).all()
        assert torch.equal(inside, alone)
