"""The persistent-workgroup form of the plain 1x1 products (gemm_x6p_kernel<.., PERSIST = true>, PECLR_X6P_PERSIST=1; off by
default: measured slower, csrc/gemm_x6p.hip) computes what the one-tile form computes, bit for bit: same tiles, same order of
products, same epilogue -- only which workgroup owns a tile changes.  Shapes: the four layers' 1x1 products of the torchvision
Bottleneck behind /root/reference/src/models/resnet_model.py:15, ragged row counts, both tile heights, 64- and 128-column tiles."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture
def persistent():
    def switch(on):
        os.environ["PECLR_X6P_PERSIST"] = "1" if on else "0"
    yield switch
    os.environ.pop("PECLR_X6P_PERSIST", None)


@pytest.mark.parametrize("m,n,k", [(8 * 56 * 56, 256, 64), (8 * 56 * 56, 64, 256), (8 * 28 * 28 + 37, 512, 128), (6272, 1024, 256),
                                   (1568 + 5, 2048, 512), (300, 128, 16), (70000, 64, 128)])
@pytest.mark.parametrize("mode", ["plain", "stats", "addend", "masked_addend_bn"])
def test_persistent_workgroups_compute_the_same_bits(persistent, m, n, k, mode):
    from peclr_amd import _capi as capi

    if mode == "masked_addend_bn" and n % 32:
        pytest.skip("1-bit masks come in words of 32 columns")
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(DEV)
    bt = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
    planes = capi.X6Planes([(bt, False)]).pack().planes[0]
    kw = {}
    if mode == "stats":
        kw["stat_shift"] = (torch.randn(n, generator=g) * 0.1).to(DEV)
    if mode in ("addend", "masked_addend_bn"):
        kw["addend"] = torch.randn(m, n, generator=g).to(DEV)
    if mode == "masked_addend_bn":
        kw["addend_mask"] = torch.randint(-2 ** 31, 2 ** 31 - 1, (m, n // 32), generator=g, dtype=torch.int32).to(DEV)
        x = torch.randn(m, n, generator=g).to(DEV)
        save = torch.stack([torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) + 0.5]).to(DEV)
        ss = torch.stack([torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g) * 0.1]).to(DEV)
        kw["bn_bwd"] = (x, save, ss, None, True)
    out = {}
    for on in (False, True):
        persistent(on)
        for tile_rows in (128, 256):
            r = capi.gemm_x6p(a, planes, n, tile_rows=tile_rows, **kw)
            out[on, tile_rows] = r if isinstance(r, tuple) else (r,)
    for tile_rows in (128, 256):
        for u, v in zip(out[True, tile_rows], out[False, tile_rows]):
            if torch.is_tensor(u):
                assert torch.equal(u, v), (mode, tile_rows)
            else:
                assert u == v
