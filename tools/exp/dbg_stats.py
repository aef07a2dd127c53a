import sys, torch
sys.path.insert(0, '/root/repo')
from peclr_amd import _capi
torch.manual_seed(0)
for (m, n, k) in [(50176, 256, 1024), (200704, 128, 512), (12544, 512, 2048), (12544, 2048, 512)]:
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') * 0.05
    shift = torch.randn(n, device='cuda') * 0.1
    pk = _capi.X6Planes([(w, False)]).pack()
    for tr in (128, 256):
        y, partial, ns = _capi.gemm_x6p(a, pk.planes[0], n, tile_rows=tr, stat_shift=shift)
        d = (y.double() - shift.double())
        s_ref, q_ref = d.sum(0), (d * d).sum(0)
        s = partial[0:2 * ns:2].double().sum(0)
        q = partial[1:2 * ns:2].double().sum(0)
        es, eq = (s - s_ref).abs() / (d.abs().sum(0)), (q - q_ref).abs() / q_ref
        print(m, n, k, tr, ns, 'sum err', float(es.max()), 'sq err', float(eq.max()), 'shift row ok', bool(torch.equal(partial[2 * ns], shift)),
              'bad cols', (eq > 1e-4).nonzero().flatten()[:10].tolist(), int((eq > 1e-4).sum()))
