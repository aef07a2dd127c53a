"""RCCL with a live communicator on ONE GPU (the most a single-GPU lease can host): the branches of the data-parallel
path that gloo never takes -- dist.all_gather_into_tensor on device tensors, asynchronous bucket all-reduces launched
between hipGraph replays on RCCL's own stream, `device_id=` initialisation -- executed for real through a one-rank
group with `peclr_amd.dist.FORCE_COLLECTIVES`.  The two-rank arithmetic is tests/test_dist_gloo.py (CPU) and
tests/test_dist_gpu_rehearsal.py (two processes on one GPU over gloo); the scaling run itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT
from tests.test_dist_gloo import free_port

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_split_graphs_with_live_rccl_collectives_match_eager_and_plain():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank.py"), str(free_port())],
                       capture_output=True, text=True, timeout=850, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    plain, eager, graph = rec["losses"]["plain"], rec["losses"]["eager_rccl"], rec["losses"]["graph_rccl"]
    # the collectives are identities on one rank: same trajectory as the run without a process group
    for a, b in zip(plain, eager):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (plain, eager)
    # capture ran 3 eager warm-up steps + the captured step's eager twin; replays continue the same trajectory
    assert len(graph) == 5 and all(g == g and abs(g) < 1e3 for g in graph)
    assert graph[0] <= plain[0] + 0.5 and graph[-1] < plain[0]          # it trains
    # what went over RCCL per replayed step: z all-gather + packed lse/stats/loss gather, one all-reduce per bucket
    assert rec["per_replay"]["all_gather_into_tensor"] == 2, rec
    assert rec["per_replay"]["all_reduce"] == rec["buckets"], rec
    assert rec["graphs"] >= 2 and rec["buckets"] >= 2
    assert rec["eager_counts"]["all_gather_into_tensor"] == 2 * len(eager)
