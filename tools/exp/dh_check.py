import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(sys.path[0], ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(sys.path[0], ".miopen", "cache"))
import torch
from oracle import step_check
from tests.test_config_sizes_gpu import build, synthetic_batch
warnings.simplefilter("ignore")
for seed in range(6):
    m = build("50", 128, seed=seed)
    b = synthetic_batch(128, 224, 100 + seed)
    d = step_check.step_deltas(m, b, backward=True)
    print(os.environ.get("PECLR_GEMM_TILE", "128"), seed, "dh_rel %.2e" % d["dh_rel"], "ties", d["relu_tie_count"], "loss d %.1e" % d["loss_delta_vs_oracle"])
