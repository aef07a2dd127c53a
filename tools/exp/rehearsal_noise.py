"""Calibration of tests/test_dist_gpu_rehearsal.py::test_two_ranks_on_one_gpu_equal_one_rank: runs its measurement N times
per arm and prints the largest deviations seen (two ranks vs one rank, both arms deterministic), next to the test's bars.
usage: python tools/exp/rehearsal_noise.py [N=20] [frozen|sync|both]"""
import pathlib
import sys
import tempfile

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from tests import test_dist_gpu_rehearsal as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
arms = {"frozen": (False,), "sync": (True,), "both": (False, True)}[sys.argv[2] if len(sys.argv) > 2 else "both"]
for sync_bn in arms:
    top, fails = {}, 0
    for i in range(n):
        with tempfile.TemporaryDirectory() as d:
            d = pathlib.Path(d)
            (d / "worker.py").write_text(T.WORKER)
            try:
                seen = T.measure_two_ranks_against_one(d / "run", d / "worker.py", sync_bn)
            except AssertionError as e:
                fails += 1
                print(f"sync_bn={sync_bn} run {i}: EXACT check failed: {str(e)[:300]}", flush=True)
                continue
        for k, v in seen["worst"].items():
            top[k] = max(top.get(k, 0.0), v)
        print(f"sync_bn={sync_bn} run {i}: " + " ".join(f"{k}={v:.3e}" for k, v in seen["worst"].items()), flush=True)
    print(f"== sync_bn={sync_bn}: {n} runs, {fails} exact-check failures; max " + " ".join(f"{k}={v:.3e}" for k, v in top.items())
          + f" | bars {T.BARS[sync_bn]}", flush=True)
