"""Which intermediate of a stride-2 first block differs between the folded and the unfolded shortcut under autocast?"""
import copy, sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_bn_shortcut_in_add_gpu import _first_block, DEV
from peclr_amd import bn2d as B

a = _first_block("bottleneck", 256, 128, 2, seed=128)
b, c = copy.deepcopy(a), copy.deepcopy(a)
B.enable_hip_batchnorm(b); B.enable_hip_batchnorm(c)
x0 = torch.randn(4, 256, 56, 56, generator=torch.Generator().manual_seed(56)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

def run(blk, fold):
    outs = {}
    names = {m: n for n, m in blk.named_modules()}
    hs = [m.register_forward_hook(lambda mod, args, out: outs.__setitem__(names[mod], (out, getattr(out, "_peclr_deferred", None))))
          for m in blk.modules() if not list(m.children())]
    with torch.no_grad(), B.routing(force=True, bn_shortcut_in_add=fold), torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x0)
    for h in hs: h.remove()
    return y, outs

ya, oa = run(a, True)
yb, ob = run(b, False)
yc, oc = run(c, False)
print("y a==b", torch.equal(ya, yb), "b==c", torch.equal(yb, yc))
for k in ob:
    ta, da = oa[k]; tb, _ = ob[k]; tc, _ = oc[k]
    if da is not None:
        from peclr_amd import _capi
        ta = _capi.bn2d_apply(da[0], da[1], relu=da[2])
        print(k, "deferred: x dtype", da[0].dtype, "ss", da[1].dtype)
    print(f"{k:16s} {str(tb.dtype):16s} a==b {torch.equal(ta, tb)}  b==c {torch.equal(tb, tc)}  max|a-b| {float((ta.float()-tb.float()).abs().max()):.3e}")
