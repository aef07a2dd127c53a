# Experiment: where does whole-step hipGraph capture stop working?  Each configuration runs in its own
# process (a crash inside hipStreamEndCapture kills only that process).
#   python tools/exp/graph_capture_sizes.py            -> runs the matrix, prints one line per config
#   python tools/exp/graph_capture_sizes.py one <resnet> <size> <pairs> <what> <bn>   -> one config
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(resnet, size, pairs, what, bn):
    import faulthandler, warnings
    faulthandler.enable()
    warnings.simplefilter("ignore")
    os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))
    import torch
    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
    from peclr_amd.bn2d import enable_hip_batchnorm
    dev = "cuda:0"
    torch.manual_seed(0)
    din = 512 if resnet in ("18", "34") else 2048
    cfg = hybrid2_config(resnet_size=resnet, projection_head_input_dim=din, augmentation=["crop", "rotate"],
                         batch_size=pairs, num_samples=64 * pairs, pretrained=False)
    model = Hybrid2Model(cfg).to(dev).train()
    model.encoder = model.encoder.to(memory_format=torch.channels_last)
    if bn == "fused":
        enable_hip_batchnorm(model.encoder)
    tr = Trainer(max_epochs=10).attach(model)
    g = torch.Generator().manual_seed(1)
    n = pairs
    batch = {"transformed_image1": torch.randn(n, 3, size, size, generator=g), "transformed_image2": torch.randn(n, 3, size, size, generator=g),
             "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
             "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
             "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    batch = {k: v.to(dev) for k, v in batch.items()}
    for k in ("transformed_image1", "transformed_image2"):
        batch[k] = batch[k].contiguous(memory_format=torch.channels_last)
    params = [p for nme, p in model.named_parameters() if "final_layer" not in nme]

    def work():
        if what == "fwd":
            with torch.no_grad():
                return model.training_step(batch, 0)["loss"]
        loss = model.training_step(batch, 0)["loss"]
        loss.backward()
        if what == "step":
            tr.optimizer.step()
        return loss

    if what.startswith("trainer"):
        if "bf16" in what:
            tr.precision = "bf16"
        if "eager" in what:          # the bench's flow: eager steps on the default stream first
            for i in range(5): tr.training_micro_step(batch, i)
            torch.cuda.synchronize()
        if "events" in what:
            from peclr_amd import _capi
            _capi.EVENT_LOG = {}
            for i in range(2): tr.training_micro_step(batch, i)
            torch.cuda.synchronize(); _capi.EVENT_LOG = None
        tr.capture_step_graph(batch)
        print("capture ok", flush=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            out = tr.replay_step()
        torch.cuda.synchronize()
        print(f"replay ok loss {float(out['loss']):.5f} {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/replay", flush=True)
        return
    if what == "eager":
        for i in range(5): tr.training_micro_step(batch, i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): out = tr.training_micro_step(batch, i)
        torch.cuda.synchronize()
        print(f"replay ok loss {float(out['loss']):.5f} {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/replay", flush=True)
        return
    split = what == "split"
    if split:
        what = "bwd"
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            for p in params: p.grad = None
            work()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print("warm ok", flush=True)
    for p in params: p.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        loss = work()
    print("capture ok", flush=True)
    if split:
        for _ in range(3):
            gr.replay(); tr.optimizer.step(); tr.scheduler.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            gr.replay(); tr.optimizer.step(); tr.scheduler.step()
        torch.cuda.synchronize()
        print(f"replay ok loss {float(loss):.5f} {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/replay", flush=True)
        return
    t0 = time.perf_counter()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    print(f"replay ok loss {float(loss):.5f} {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/replay", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6])
        sys.exit(0)
    matrix = [l.split() for l in (sys.argv[1:] or [
        "50 224 128 trainer_eager fused", "50 224 128 trainer_eager_events fused", "50 224 128 trainer_bf16 fused", "50 224 128 trainer_eager_bf16 fused", "50 224 128 trainer_eager stock", "152 224 128 trainer_eager fused"])]
    for cfg in matrix:
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", *cfg], capture_output=True, text=True, timeout=900)
        tail = [l for l in (p.stdout + p.stderr).splitlines() if l.strip()][-6:]
        stage = [l for l in p.stdout.splitlines() if l.endswith("ok") or l.startswith("replay ok")]
        print(f"{' '.join(cfg):28s} rc={p.returncode:4d} {time.time() - t0:5.0f}s  reached: {stage[-1] if stage else 'nothing'}", flush=True)
        if p.returncode != 0:
            print("    " + "\n    ".join(tail), flush=True)
