"""ctypes binding of libpeclr_hip.so (include/peclr_hip.h) for torch device tensors.

This is the ONLY place the product touches the native library, and there is no fallback:
if the library is missing, or a tensor is not a contiguous fp32 HIP tensor, the call raises.
PyTorch is plumbing here (device memory + the current hipStream_t); every function below is a
thin argument marshaller around one C entry point.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch  # must be imported BEFORE the CDLL: the .so binds to torch's libamdhip64.so.7

# (PECLR_HIP_LIB: another build of the same library, for same-box A/B runs of a kernel change)
_LIB_PATH = os.environ.get("PECLR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpeclr_hip.so")
_LIB = None

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
ALIGN_CROP, ALIGN_ROTATE, ALIGN_SINGLE_NORM = 1, 2, 4
OPT_CHUNK = 4096

# name -> (restype, argtypes); mirrors include/peclr_hip.h one to one
_P = c_void_p
SIGNATURES = {
    "peclr_version": (c_int, []),
    "peclr_error_string": (c_char_p, [c_int]),
    "peclr_stream_capture_id": (c_int, [_P, _P]),
    "peclr_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P]),
    "peclr_gemm_add_f32": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "peclr_gemm_add_bf16": (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "peclr_gemm_add_f16": (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "peclr_gemm_pick_split_k": (c_int, [c_int, c_int, c_int]),
    "peclr_slab_reduce_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "peclr_bn_relu_fwd_f32": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, c_float, c_float, c_int, _P, _P,
                                      _P, _P, _P, _P, _P, _P]),
    "peclr_bn_relu_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "peclr_align_fwd_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float,
                                    _P, _P, _P, _P, _P, _P, _P]),
    "peclr_align_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "peclr_ntxent_jsplit": (c_int, [c_int, c_int, c_int]),
    "peclr_ntxent_fwd_f32": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, c_float, _P, _P, _P, c_int, _P]),
    "peclr_ntxent_finalize_f32": (c_int, [_P, c_int, _P, c_int, c_float, _P, _P, c_int, _P, _P]),
    "peclr_ntxent_bwd_f32": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, c_float, _P, _P, c_float, _P,
                                     c_int, _P]),
    "peclr_stem_pack_bytes": (c_int, [c_int]),
    "peclr_stem_pack": (c_int, [_P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, c_int, _P]),
    "peclr_stem_workgroups": (c_int, [c_int, c_int, c_int]),
    "peclr_stem_conv7x7_s2": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "peclr_stem_wgrad_slabs": (c_int, [c_int, c_int, c_int, c_int]),
    "peclr_stem_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "peclr_bn2d_n_split": (c_int, [c_int, c_int, c_int]),
    "peclr_bn2d_stats": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "peclr_bn2d_combine_f64": (c_int, [_P, c_int, c_int, _P, _P]),
    "peclr_bn2d_finalize_totals_f32": (c_int, [_P, _P, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P,
                                               _P]),
    "peclr_bn2d_bwd_finalize_totals_f32": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "peclr_bn2d_finalize_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P]),
    "peclr_bn2d_apply": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "peclr_bn2d_apply_res_bn": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "peclr_bn2d_bwd_reduce": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "peclr_bn2d_bwd_finalize_f32": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "peclr_bn2d_bwd_apply": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "peclr_bn2d_apply_avgpool": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "peclr_bn2d_bwd_reduce_avgpool": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "peclr_bn2d_bwd_apply_avgpool": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "peclr_bn2d_pool_n_split": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "peclr_bn2d_pool_apply": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "peclr_bn2d_pool_bwd_reduce": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "peclr_bn2d_pool_bwd_apply": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "peclr_augment_warp_crop_u8": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "peclr_augment_resize_color_norm": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, _P,
                                                _P]),
    "peclr_gemm_x6_f32": (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "peclr_gemm_x6_tn_slabs": (c_int, [c_int, c_int, c_int]),
    "peclr_gemm_x6_tn_f32": (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "peclr_x6_pack_bytes": (c_int64, [c_int, c_int]),
    "peclr_x6_pack_f32": (c_int, [_P, c_int, c_int, _P]),
    "peclr_x6_pack_pair_bytes": (c_int64, [c_int, c_int]),
    "peclr_x6_absmax_f32": (c_int, [_P, c_int, c_int, _P, _P]),
    "peclr_x6_pack_pair_f32": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "peclr_gemm_x6p_tile_rows": (c_int, [c_int, c_int, c_int]),
    "peclr_gemm_x6p_f32": (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "peclr_conv3x3_s2_dgrad_x6p_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "peclr_gemm_x6p_s2add_f32": (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "peclr_gemm_x6p_maskadd_f32": (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P]),
    "peclr_gemm_x6t_slabs": (c_int, [c_int, c_int, c_int, c_int]),
    "peclr_gemm_x6t_f32": (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "peclr_conv_s2_x6p_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "peclr_conv3x3_x6p_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "peclr_h_pack_bytes": (c_int64, [c_int, c_int]),
    "peclr_h_pack": (c_int, [_P, c_int, c_int, _P]),
    "peclr_conv_h_tile_rows": (c_int, [c_int, c_int]),
    "peclr_conv_h_row_blocks": (c_int, [c_int] * 7),
    "peclr_gemm_h": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "peclr_conv_h": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "peclr_conv3x3_s2_dgrad_h": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "peclr_wgrad3_h_slabs": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "peclr_wgrad3_h": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "peclr_wgrad_h_slabs": (c_int, [c_int, c_int, c_int]),
    "peclr_wgrad_h": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "peclr_lars_sumsq_f32": (c_int, [_P, _P, c_int, _P, _P, c_int, _P, _P]),
    "peclr_lars_adam_update_f32": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_float,
                                           c_float, c_float, c_float, c_float, c_int, c_float, c_float, c_int,
                                           _P]),
    "peclr_lars_sumsq_amp_f32": (c_int, [_P, _P, c_int, _P, _P, c_int, _P, _P, _P]),
    "peclr_lars_adam_update_amp_f32": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_double,
                                               c_double, c_float, c_int, c_float, c_float, c_int, _P, _P]),
    "peclr_amp_update": (c_int, [_P, c_float, c_float, c_int, _P]),
}


class PeclrHipError(RuntimeError):
    pass


def library_path() -> str:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    """Load libpeclr_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise PeclrHipError(
                f"{_LIB_PATH} is missing: build it with `make -C peclr_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "peclr_amd has no CPU or PyTorch fallback for its HIP kernels.")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().peclr_error_string(rc).decode()
        raise PeclrHipError(f"{what} failed: {msg} (code {rc})")


def _ptr(t: Optional[torch.Tensor], dtype=torch.float32, what: str = "tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise PeclrHipError(f"{what}: expected a HIP device tensor, got device={t.device} "
                            "(peclr_amd has no CPU path)")
    if t.dtype != dtype:
        raise PeclrHipError(f"{what}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise PeclrHipError(f"{what}: tensor must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def capture_id() -> int:
    """Identity of the hipGraph capture the current stream is in; 0 when it is not capturing."""
    if not torch.cuda.is_current_stream_capturing():
        return 0
    out = ctypes.c_ulonglong(0)
    _check(lib().peclr_stream_capture_id(_stream(), ctypes.addressof(out)), "peclr_stream_capture_id")
    return int(out.value) or 1


# ---- optional per-kernel HIP-event timing (bench.py): one entry point = one launch, so an event
# pair recorded on the launch stream around a call times exactly that kernel.
EVENT_LOG = None  # None = off; dict name -> list[(start, end)] when bench.py turns it on
WEIGHTS_EPOCH = 0    # bumped by every fused optimiser launch (it updates the parameters through raw pointers)
TAG_BOUND_SUFFIX = False   # bench.py: launches of the six-product GEMMs whose OWN roof is HBM (layer1 / layer2's 1x1 shapes: 4 B/elem
#                            x (K + N) columns take longer at 8 TB/s than 2 K N flops at 417 TFLOP/s) log as "<tag>~hbm", so that a tag's
#                            average is never a mix of MFMA-bound and HBM-bound launches priced against one roof
LAUNCH_ORDER = None  # None = off; list of names in launch order (one entry per launch) while EVENT_LOG is on:
#                      lets tools/pmc_mfma.py align a rocprofv3 dispatch table with the bench's kernel names


class _timed:
    """`nbytes` / `flops`: the launch's ALGORITHMIC work when it is shape-dependent (summed per name); `kernel`: the
    kernel family that runs under this name when the entry point has more than one (bench.py prices a launch against
    the roof of the kernel that actually ran)."""
    __slots__ = ("name", "s", "e", "nbytes", "flops", "kernel")

    def __init__(self, name, nbytes=0, flops=0, kernel=None):
        # (the kernel's own matrix-core roof: dense bf16 / fp16 peak over six products, or over three in pair arithmetic)
        if (TAG_BOUND_SUFFIX and flops and nbytes and kernel and kernel.startswith("gemm_x6")
                and nbytes / 8e12 > flops / (833.3e12 if "<pair>" in kernel else 416.7e12)):
            name += "~hbm"
        self.name, self.nbytes, self.flops, self.kernel = name, nbytes, flops, kernel

    def __enter__(self):
        if EVENT_LOG is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()  # current stream == the stream the kernel is launched on
            if LAUNCH_ORDER is not None:
                LAUNCH_ORDER.append(self.name)

    def __exit__(self, *exc):
        if EVENT_LOG is not None:
            self.e.record()
            EVENT_LOG.setdefault(self.name, []).append((self.s, self.e, self.nbytes, self.flops, self.kernel))
        return False


# ------------------------------------------------------------------ GEMM
def pick_split_k(m: int, n: int, k: int) -> int:
    return lib().peclr_gemm_pick_split_k(m, n, k)


def gemm(layout: int, a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None,
         split_k: int = 1, tag: Optional[str] = None) -> torch.Tensor:
    """Returns C [M,N] (split_k == 1) or slabs [split_k, M, N]."""
    if layout == GEMM_NT:
        (m, k), (n, k2) = a.shape, b.shape
    elif layout == GEMM_NN:
        (m, k), (k2, n) = a.shape, b.shape
    else:
        (k, m), (k2, n) = a.shape, b.shape
    if k != k2:
        raise PeclrHipError(f"gemm: contraction mismatch {tuple(a.shape)} x {tuple(b.shape)} layout {layout}")
    if split_k == 1:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
        c_ptr, slab_ptr = out.data_ptr(), None
    else:
        out = torch.empty((split_k, m, n), device=a.device, dtype=torch.float32)
        c_ptr, slab_ptr = None, out.data_ptr()
    with _timed(tag or f"gemm_{('nt', 'nn', 'tn')[layout]}_{m}x{n}x{k}"):
        rc = lib().peclr_gemm_f32(layout, m, n, k, _ptr(a, what="gemm A"), a.stride(0), _ptr(b, what="gemm B"),
                                  b.stride(0), c_ptr, n, _ptr(bias, what="gemm bias"), split_k, slab_ptr,
                                  _stream())
    _check(rc, "peclr_gemm_f32")
    return out


def slab_reduce(slabs: torch.Tensor, bias: Optional[torch.Tensor] = None, tag: str = "slab_reduce") -> torch.Tensor:
    s, rows, cols = slabs.shape
    out = torch.empty((rows, cols), device=slabs.device, dtype=torch.float32)
    with _timed(tag, 4 * (s + 1) * rows * cols, kernel="slab_reduce_kernel"):
        rc = lib().peclr_slab_reduce_f32(_ptr(slabs), s, rows, cols, _ptr(bias), out.data_ptr(), _stream())
    _check(rc, "peclr_slab_reduce_f32")
    return out


# ------------------------------------------------------------------ BN + ReLU
def bn_relu_fwd(a_slabs, bias, gamma, beta, eps, momentum, training, running_mean, running_var,
                num_batches_tracked):
    s, m, h = a_slabs.shape
    dev = a_slabs.device
    a_pre = torch.empty((m, h), device=dev, dtype=torch.float32)
    a_out = torch.empty((m, h), device=dev, dtype=torch.float32)
    save = torch.empty((2, h), device=dev, dtype=torch.float32)
    with _timed("bn_relu_fwd"):
        rc = lib().peclr_bn_relu_fwd_f32(
            _ptr(a_slabs), s, _ptr(bias), m, h, _ptr(gamma), _ptr(beta), eps, momentum, int(training),
            _ptr(running_mean), _ptr(running_var),
            _ptr(num_batches_tracked, torch.int64, "num_batches_tracked"), a_pre.data_ptr(), a_out.data_ptr(),
            save[0].data_ptr(), save[1].data_ptr(), _stream())
    _check(rc, "peclr_bn_relu_fwd_f32")
    return a_pre, a_out, save


def bn_relu_bwd(d_a_out, a_pre, save, gamma, beta, training=True):
    m, h = a_pre.shape
    d_a_pre = torch.empty_like(a_pre)
    dparams = torch.empty((3, h), device=a_pre.device, dtype=torch.float32)  # dgamma, dbeta, dbias
    with _timed("bn_relu_bwd"):
        rc = lib().peclr_bn_relu_bwd_f32(_ptr(d_a_out), _ptr(a_pre), save[0].data_ptr(), save[1].data_ptr(),
                                         _ptr(gamma), _ptr(beta), m, h, int(training), d_a_pre.data_ptr(),
                                         dparams[0].data_ptr(), dparams[1].data_ptr(), dparams[2].data_ptr(),
                                         _stream())
    _check(rc, "peclr_bn_relu_bwd_f32")
    return d_a_pre, dparams[0], dparams[1], dparams[2]


# ------------------------------------------------------------------ align
def align_fwd(p_slabs, n_pairs, flags, jitter, extents, angles, want_stats=True):
    """jitter = (jx1, jx2, jy1, jy2) int64 tensors or None; angles = (a1, a2) float64 or None."""
    s, m, d = p_slabs.shape
    dev = p_slabs.device
    p = torch.empty((m, d), device=dev, dtype=torch.float32)
    z = torch.empty((m, d), device=dev, dtype=torch.float32)
    norms = torch.empty((2, m), device=dev, dtype=torch.float32)
    row_stats = torch.empty((m, 8), device=dev, dtype=torch.float32) if want_stats else None
    j = [_ptr(t, torch.int64, "jitter") for t in jitter] if jitter is not None else [None] * 4
    a = [_ptr(t, torch.float64, "angle") for t in angles] if angles is not None else [None] * 2
    with _timed("align_fwd"):
        rc = lib().peclr_align_fwd_f32(_ptr(p_slabs), s, m, d, n_pairs, flags, j[0], j[1], j[2], j[3],
                                       float(extents[0]), float(extents[1]), a[0], a[1], p.data_ptr(),
                                       z.data_ptr(), norms.data_ptr(), _ptr(row_stats), _stream())
    _check(rc, "peclr_align_fwd_f32")
    return p, z, norms, row_stats


def align_bwd(dz, p, z, norms, n_pairs, flags, angles):
    m, d = p.shape
    dp = torch.empty_like(p)
    a = [_ptr(t, torch.float64, "angle") for t in angles] if angles is not None else [None] * 2
    with _timed("align_bwd"):
        rc = lib().peclr_align_bwd_f32(_ptr(dz, what="dz"), _ptr(p), _ptr(z), _ptr(norms), m, d, n_pairs,
                                       flags, a[0], a[1], dp.data_ptr(), _stream())
    _check(rc, "peclr_align_bwd_f32")
    return dp


# ------------------------------------------------------------------ NT-Xent
def ntxent_jsplit(mr: int, mg: int, backward: bool) -> int:
    js = lib().peclr_ntxent_jsplit(mr, mg, int(backward))
    if js < 1:
        raise PeclrHipError(f"peclr_ntxent_jsplit: unsupported shape Mr={mr} Mg={mg}")
    return js


def ntxent_fwd(z_rows, row_offset, z_all, n_half, inv_tau, loss_scale, row_stats=None, n_pairs_stats=0,
               want_sim=False):
    """Main kernel + finalize kernel.  Returns (out17, row_lse, sim)."""
    mr, d = z_rows.shape
    mg = z_all.shape[0]
    dev = z_rows.device
    js = ntxent_jsplit(mr, mg, False)
    partial = torch.empty((js, mr), device=dev, dtype=torch.float32)
    pos = torch.empty(mr, device=dev, dtype=torch.float32)
    row_lse = torch.empty(mr, device=dev, dtype=torch.float32)
    out17 = torch.zeros(17, device=dev, dtype=torch.float32)
    sim = torch.empty((mr, mg), device=dev, dtype=torch.float32) if want_sim else None
    with _timed("ntxent_fwd"):
        rc = lib().peclr_ntxent_fwd_f32(_ptr(z_rows, what="z_rows"), mr, row_offset, _ptr(z_all, what="z_all"),
                                        mg, d, n_half, inv_tau, _ptr(sim), partial.data_ptr(), pos.data_ptr(),
                                        js, _stream())
    _check(rc, "peclr_ntxent_fwd_f32")
    with _timed("ntxent_finalize"):
        rc = lib().peclr_ntxent_finalize_f32(partial.data_ptr(), js, pos.data_ptr(), mr, loss_scale,
                                             row_lse.data_ptr(), _ptr(row_stats), n_pairs_stats,
                                             out17.data_ptr(), _stream())
    _check(rc, "peclr_ntxent_finalize_f32")
    return out17, row_lse, sim


def ntxent_bwd(z_rows, row_offset, z_all, n_half, inv_tau, lse_all, dloss, grad_scale):
    mr, d = z_rows.shape
    mg = z_all.shape[0]
    js = ntxent_jsplit(mr, mg, True)
    slabs = torch.empty((js, mr, d), device=z_rows.device, dtype=torch.float32)
    with _timed("ntxent_bwd"):
        rc = lib().peclr_ntxent_bwd_f32(_ptr(z_rows, what="z_rows"), mr, row_offset, _ptr(z_all, what="z_all"),
                                        mg, d, n_half, inv_tau, _ptr(lse_all, what="lse_all"),
                                        _ptr(dloss, what="dloss"), grad_scale, slabs.data_ptr(), js, _stream())
    _check(rc, "peclr_ntxent_bwd_f32")
    return slabs[0] if js == 1 else slab_reduce(slabs)


# ------------------------------------------------------------------ optimiser
def lars_adam_step(ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, tensor_chunk_begin, tensor_group, n_chunks,
                   norms_ws, group_lr, group_wd, beta1, beta2, adam_eps, bias_corr1, bias_corr2, use_lars,
                   lars_eta, lars_eps, lars_clip, device_hyper=None, amp=None):
    """group_lr / group_wd: Python float lists, one entry per parameter group (host arrays).
    device_hyper: optional device float[18] that overrides lr / wd / bias corrections (graph replay).
    amp: optional (state int32[4] device tensor = peclr_amp_state, growth_factor, backoff_factor, growth_interval):
    the gradients hold scale*g; inf/nan check + unscale + skip + scale update on the device (three launches)."""
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1
    p = _ptr(ptrs, torch.int64, "ptrs")
    sz = _ptr(sizes, torch.int64, "sizes")
    ct = _ptr(chunk_tensor, torch.int32, "chunk_tensor")
    co = _ptr(chunk_offset, torch.int64, "chunk_offset")
    ng = len(group_lr)
    lr_arr = (c_float * ng)(*group_lr)
    wd_arr = (c_float * ng)(*group_wd)
    if amp is not None:
        state, growth, backoff, interval = amp
        st = _ptr(state, torch.int32, "amp state")
        if state.numel() != 4:
            raise PeclrHipError("amp state: expected 4 x 32-bit words (peclr_amp_state)")
        with _timed("lars_sumsq"):
            rc = lib().peclr_lars_sumsq_amp_f32(p, sz, n_tensors, ct, co, n_chunks, _ptr(norms_ws), st, _stream())
        _check(rc, "peclr_lars_sumsq_amp_f32")
        with _timed("lars_adam_update"):
            rc = lib().peclr_lars_adam_update_amp_f32(
                p, sz, n_tensors, ct, co, _ptr(tensor_chunk_begin, torch.int32, "tensor_chunk_begin"),
                _ptr(tensor_group, torch.int32, "tensor_group"), n_chunks, _ptr(norms_ws), _ptr(device_hyper),
                ctypes.cast(lr_arr, c_void_p), ctypes.cast(wd_arr, c_void_p), ng, beta1, beta2, adam_eps,
                int(use_lars), lars_eta, lars_eps, int(lars_clip), st, _stream())
        _check(rc, "peclr_lars_adam_update_amp_f32")
        with _timed("amp_update", nbytes=16):
            rc = lib().peclr_amp_update(st, growth, backoff, int(interval), _stream())
        _check(rc, "peclr_amp_update")
        return
    if use_lars:
        with _timed("lars_sumsq"):
            rc = lib().peclr_lars_sumsq_f32(p, sz, n_tensors, ct, co, n_chunks, _ptr(norms_ws), _stream())
        _check(rc, "peclr_lars_sumsq_f32")
    with _timed("lars_adam_update"):
        rc = lib().peclr_lars_adam_update_f32(
            p, sz, n_tensors, ct, co, _ptr(tensor_chunk_begin, torch.int32, "tensor_chunk_begin"),
            _ptr(tensor_group, torch.int32, "tensor_group"), n_chunks, _ptr(norms_ws), _ptr(device_hyper),
            ctypes.cast(lr_arr, c_void_p), ctypes.cast(wd_arr, c_void_p), ng, beta1, beta2, adam_eps, bias_corr1,
            bias_corr2, int(use_lars), lars_eta, lars_eps, int(lars_clip), _stream())
    _check(rc, "peclr_lars_adam_update_f32")


def gemm_add(layout: int, a: torch.Tensor, b: torch.Tensor, addend: torch.Tensor, tag: str = "gemm_add") -> torch.Tensor:
    """C = op(A) op(B) + addend, all fp32 row-major contiguous 2-D HIP tensors."""
    if layout == GEMM_NT:
        (m, k), (n, k2) = a.shape, b.shape
    elif layout == GEMM_NN:
        (m, k), (k2, n) = a.shape, b.shape
    else:
        (k, m), (k2, n) = a.shape, b.shape
    if k != k2 or tuple(addend.shape) != (m, n):
        raise PeclrHipError(f"gemm_add: shapes {tuple(a.shape)} x {tuple(b.shape)} + {tuple(addend.shape)} (layout {layout})")
    out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    with _timed(tag, 4 * (m * k + k * n + 2 * m * n), 2 * m * n * k, kernel="gemm_f32 (v_mfma_f32)"):
        rc = lib().peclr_gemm_add_f32(layout, m, n, k, _ptr(a), a.shape[1], _ptr(b), b.shape[1], out.data_ptr(), n,
                                      _ptr(addend), n, _stream())
    _check(rc, "peclr_gemm_add_f32")
    return out


def gemm_x6(a: torch.Tensor, b_t: torch.Tensor, addend: Optional[torch.Tensor] = None, tag: str = "gemm_x6",
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C (fp32) = A[M,K] . B_t[N,K]^T (+ addend), fp32 row-major contiguous 2-D HIP tensors, computed on the bf16
    matrix cores at fp32 accuracy (exact three-way bf16 split of both operands, six products: peclr_gemm_x6_f32)."""
    (m, k), (n, k2) = a.shape, b_t.shape
    if k != k2 or (addend is not None and tuple(addend.shape) != (m, n)):
        raise PeclrHipError(f"gemm_x6: shapes {tuple(a.shape)} x {tuple(b_t.shape)}^T")
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    with _timed(tag, 4 * (m * k + k * n + (2 if addend is not None else 1) * m * n), 2 * m * n * k, kernel="gemm_x6_nt128_kernel"):
        rc = lib().peclr_gemm_x6_f32(m, n, k, _ptr(a), k, _ptr(b_t), k, _ptr(out), n, _ptr(addend), n, _stream())
    _check(rc, "peclr_gemm_x6_f32")
    return out


_X6P_TILE_ROWS = int(os.environ.get("PECLR_X6P_TILE_ROWS", "0"))   # experiments: force 128- or 256-row tiles for the 1x1 GEMMs


class _BnBwdFuse(ctypes.Structure):
    """peclr_bn_bwd_fuse (include/peclr_hip.h)."""
    _fields_ = [("x", c_void_p), ("mean", c_void_p), ("invstd", c_void_p), ("scale_shift", c_void_p), ("relu_mask", c_void_p),
                ("relu", c_int), ("partial", c_void_p)]


def _bn_bwd_fuse(bn_bwd, m: int, n: int, tile_rows: int, groups: int = 1, dtype=torch.float32, row_blocks: int = 0):
    """bn_bwd = (x [.., n] NHWC/2-D fp32 of m rows, save [2, n], ss [2, n], mask or None, relu) of the BatchNorm layer whose
    incoming gradient this GEMM produces -> (struct, partial [2 * n_split, n], n_split); keeps the tensors alive."""
    x, save, ss, mask, relu = bn_bwd
    if x.dtype != dtype or x.numel() != m * n or not x.is_cuda:
        raise PeclrHipError(f"bn backward fusion: layer input of {x.numel()} {x.dtype} elements for a {dtype} [{m}, {n}] gradient")
    ns = row_blocks or groups * ((m // groups + tile_rows - 1) // tile_rows)       # (groups: row blocks of each of several equal row sets)
    partial = torch.empty((2 * ns, n), device=x.device, dtype=torch.float32)
    st = _BnBwdFuse(x.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), _ptr(ss), _ptr(mask, torch.int32, "relu mask"), int(relu),
                    partial.data_ptr())
    return st, partial, ns


class X6Planes:
    """Weight matrices split once into fragment-ordered bf16 planes (peclr_x6_pack_f32) for peclr_gemm_x6p_f32.
    `specs`: list of (fp32 2-D HIP tensor W, transposed) -- B_t = W ([N, K]) or W^T (W is [K, N]).  The device table is
    built once (the tensors' storage must stay where it is: parameters do); `pack()` is ONE launch that re-splits every
    matrix from its current values -- call it after the weights changed (once per optimiser step)."""

    def __init__(self, specs, pair: bool = False):
        """pair: the fp16-pair format of peclr_x6_pack_pair_f32 (two planes, one power of two per matrix: `scale(i)`) instead of
        the three bf16 planes."""
        if not specs:
            raise PeclrHipError("X6Planes: nothing to pack")
        dev = specs[0][0].device
        self.pair = bool(pair)
        rows, self.planes, self.shapes, chunk = [], [], [], 0
        for w, transposed in specs:
            _ptr(w, what="x6 weight")
            if w.dim() != 2:
                raise PeclrHipError("X6Planes: 2-D weight matrices expected")
            t = int(transposed)                     # 0 plain, 1 transposed, T > 1: T-tap filter [Cout * T, Cin] for its input gradient
            n, k = (w.shape[1], w.shape[0]) if t else (w.shape[0], w.shape[1])
            nbytes = lib().peclr_x6_pack_pair_bytes(n, k) if self.pair else lib().peclr_x6_pack_bytes(n, k)
            if nbytes <= 0:
                raise PeclrHipError(f"X6Planes: B_t[{n}, {k}] needs n % 64 == 0 and k % 16 == 0")
            planes = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            rows.append([w.data_ptr(), planes.data_ptr(), n, k, w.stride(0), t, chunk, 0])
            chunk += ((n + 127) // 128) * (k // 16)
            self.planes.append(planes)
            self.shapes.append((n, k))
        self._sources = [w for w, _ in specs]          # keep the storage alive
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.count, self.chunks = len(rows), chunk
        self.nbytes = sum(p.numel() for p in self.planes) + 4 * sum(w.numel() for w in self._sources)
        if self.pair:
            self.wbytes = 4 * sum(w.numel() for w in self._sources)       # (the maxima's pass reads the weights once more)
            self.absmax = torch.zeros(self.count, device=dev, dtype=torch.float32)
            self.scales = torch.ones(self.count, device=dev, dtype=torch.float32)

    def scale(self, i: int) -> torch.Tensor:
        """The device float holding the power of two matrix i was multiplied by (pair format)."""
        return self.scales[i:i + 1]

    def pack(self):
        if self.pair:
            with _timed("x6_absmax", self.wbytes, kernel="x6_pair_kernel"):
                rc = lib().peclr_x6_absmax_f32(self.table.data_ptr(), self.count, self.chunks, self.absmax.data_ptr(), _stream())
            _check(rc, "peclr_x6_absmax_f32")
            with _timed("x6_pack", self.nbytes, kernel="x6_pair_kernel"):
                rc = lib().peclr_x6_pack_pair_f32(self.table.data_ptr(), self.count, self.chunks, self.absmax.data_ptr(),
                                                  self.scales.data_ptr(), _stream())
            _check(rc, "peclr_x6_pack_pair_f32")
            return self
        with _timed("x6_pack", self.nbytes, kernel="x6_pack_kernel"):
            rc = lib().peclr_x6_pack_f32(self.table.data_ptr(), self.count, self.chunks, _stream())
        _check(rc, "peclr_x6_pack_f32")
        return self


class _X6Pair(ctypes.Structure):
    _fields_ = [("a_absmax", ctypes.c_void_p), ("w_scale", ctypes.c_void_p)]


def _pair_arg(pair, who: str):
    """pair = None (six-product arithmetic on bf16-triple planes) or (a_absmax, w_scale): one-element fp32 HIP tensors -- the
    maximum of |A| over the whole activation tensor (written by the pass that produced it) and the weight planes' power of two
    (`X6Planes(pair=True).scale(i)`).  Returns (struct or None to pass, bytes per weight element of the planes)."""
    if pair is None:
        return None, 6
    a_absmax, w_scale = pair
    for t, name in ((a_absmax, "a_absmax"), (w_scale, "w_scale")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
            raise PeclrHipError(f"{who}: pair.{name} is a one-element fp32 HIP tensor")
    return _X6Pair(a_absmax.data_ptr(), w_scale.data_ptr()), 4


def gemm_x6p(a: torch.Tensor, planes: torch.Tensor, n: int, addend: Optional[torch.Tensor] = None, tag: str = "gemm_x6p",
             tile_rows: int = 0, stat_shift: Optional[torch.Tensor] = None, bn_bwd=None, addend_s2=None,
             addend_mask: Optional[torch.Tensor] = None, pair=None):
    """C (fp32) [M, n] = A[M, K] . B_t^T (+ addend) with B_t given as packed planes (X6Planes): fp32 accuracy on the
    bf16 matrix cores, the weight operand split once per step (peclr_gemm_x6p_f32).
    stat_shift (fp32 [n]): also return the training-mode BatchNorm statistics of C as `(partial, n_split)` in the layout
    of peclr_bn2d_stats (sums of (C - shift) and its square per row block; the shift in the last row) -> (C, partial, n_split).
    bn_bwd (see `_bn_bwd_fuse`): C is the gradient arriving at that BatchNorm layer; also return its backward reduction
    `(partial, n_split)` in peclr_bn2d_bwd_reduce's layout -> (C, partial, n_split).
    addend_s2 = (H, W): the rows are the pixels of H x W images and `addend` [M / 4, n] holds every second pixel only (the
    compact input gradient of a 1x1 / stride-2 convolution): added at the even (h, w) rows (peclr_gemm_x6p_s2add_f32).
    addend_mask (int32 [M, n / 32], the 1-bit ReLU mask of peclr_bn2d_apply): addend elements whose bit is clear count as
    zero (peclr_gemm_x6p_maskadd_f32).
    pair (see `_pair_arg`): fp16-pair arithmetic, `planes` of `X6Planes(pair=True)`."""
    m, k = a.shape
    add_rows = m if addend_s2 is None else m // 4
    pst, wb = _pair_arg(pair, "gemm_x6p")
    pref = ctypes.byref(pst) if pst is not None else None
    if planes.dtype != torch.uint8 or planes.numel() != wb * ((n + 127) // 128 * 128) * k or (addend is not None and tuple(addend.shape) != (add_rows, n)):
        raise PeclrHipError(f"gemm_x6p: A {tuple(a.shape)}, planes of {planes.numel()} bytes for B_t[{n}, {k}]")
    if addend_s2 is not None and (addend is None or stat_shift is not None or addend_mask is not None):
        raise PeclrHipError("gemm_x6p: addend_s2 needs the compact addend (and has no statistics output / mask)")
    if addend_mask is not None and (addend is None or stat_shift is not None or n % 32 or addend_mask.dtype != torch.int32
                                    or addend_mask.numel() != m * (n // 32) or not addend_mask.is_contiguous()):
        raise PeclrHipError("gemm_x6p: addend_mask is the int32 [M, n / 32] bit mask of a dense addend (no statistics output)")
    out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    partial, ns, fuse = None, 0, None
    tile_rows = tile_rows or _X6P_TILE_ROWS
    if stat_shift is not None or bn_bwd is not None:
        tile_rows = tile_rows or lib().peclr_gemm_x6p_tile_rows(m, n, k)
    if stat_shift is not None:
        if stat_shift.numel() != n:
            raise PeclrHipError(f"gemm_x6p: stat_shift has {stat_shift.numel()} entries for {n} columns")
        ns = (m + tile_rows - 1) // tile_rows
        partial = torch.empty((2 * ns + 1, n), device=a.device, dtype=torch.float32)
    elif bn_bwd is not None:
        fuse, partial, ns = _bn_bwd_fuse(bn_bwd, m, n, tile_rows)
    add_elems = 0 if addend is None else addend.numel() + (0 if addend_mask is None else addend_mask.numel())
    with _timed(tag, 4 * (m * k + m * n + add_elems + (m * n if fuse is not None else 0)) + wb * k * n, 2 * m * n * k,
                kernel="gemm_x6p_kernel" if pst is None else "gemm_x6p_kernel<pair>"):
        if addend_mask is not None:
            rc = lib().peclr_gemm_x6p_maskadd_f32(m, n, k, _ptr(a), k, _ptr(planes, torch.uint8), out.data_ptr(), n, _ptr(addend), n,
                                                  _ptr(addend_mask, torch.int32), tile_rows,
                                                  ctypes.byref(fuse) if fuse is not None else None, pref, _stream())
        elif addend_s2 is not None:
            rc = lib().peclr_gemm_x6p_s2add_f32(m, n, k, _ptr(a), k, _ptr(planes, torch.uint8), out.data_ptr(), n, _ptr(addend), n,
                                                int(addend_s2[0]), int(addend_s2[1]), tile_rows,
                                                ctypes.byref(fuse) if fuse is not None else None, pref, _stream())
        else:
            rc = lib().peclr_gemm_x6p_f32(m, n, k, _ptr(a), k, _ptr(planes, torch.uint8), out.data_ptr(), n, _ptr(addend), n,
                                          tile_rows, _ptr(stat_shift), partial.data_ptr() if stat_shift is not None else None,
                                          ctypes.byref(fuse) if fuse is not None else None, pref, _stream())
    _check(rc, "peclr_gemm_x6p_maskadd_f32" if addend_mask is not None else "peclr_gemm_x6p_s2add_f32" if addend_s2 is not None
           else "peclr_gemm_x6p_f32")
    return out if partial is None else (out, partial, ns)


_CONV3X3_VARIANT = int(os.environ.get("PECLR_CONV3X3_HALO", "1"))   # A/B: 1 = one split per 16-channel chunk and workgroup (halo patch in LDS)
_ZEROS = {}


def _zeros(device):
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(64, device=device, dtype=torch.float32)
    return z


def conv3x3_x6p(x: torch.Tensor, planes: torch.Tensor, cout: int, flip: bool = False, addend: Optional[torch.Tensor] = None,
                tag: str = "conv3x3_x6p", tile_rows: int = 0, stat_shift: Optional[torch.Tensor] = None, bn_bwd=None, variant: Optional[int] = None,
                pair=None):
    """3x3 / stride-1 / padding-1 convolution of an NHWC (channels_last) fp32 tensor x [N, Cin, H, W] as an implicit GEMM on
    the bf16 matrix cores at fp32 accuracy (peclr_conv3x3_x6p_f32); `planes` = X6Planes of W seen as [Cout, 9 * Cin]
    (flip=False) or, for the input gradient (flip=True, x = dY), of [Cout_w * 9, Cin_w] packed with transposed = 9.
    Returns y [N, cout, H, W] channels_last (and (partial, n_split) when stat_shift or bn_bwd is given, as in gemm_x6p)."""
    nb, cin, h, w = x.shape
    xp = _nhwc_ptr(x, "conv3x3 x", torch.float32)
    pst, wb = _pair_arg(pair, "conv3x3_x6p")
    if planes.dtype != torch.uint8 or planes.numel() != wb * ((cout + 127) // 128 * 128) * 9 * cin:
        raise PeclrHipError(f"conv3x3_x6p: planes of {planes.numel()} bytes for [{cout}, 9 * {cin}]")
    y = torch.empty((nb, cout, h, w), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    m = nb * h * w
    partial, ns, fuse = None, 0, None
    if stat_shift is not None or bn_bwd is not None:
        tile_rows = tile_rows or lib().peclr_gemm_x6p_tile_rows(m, cout, 9 * cin)
    if stat_shift is not None:
        ns = (m + tile_rows - 1) // tile_rows
        partial = torch.empty((2 * ns + 1, cout), device=x.device, dtype=torch.float32)
    elif bn_bwd is not None:
        fuse, partial, ns = _bn_bwd_fuse(bn_bwd, m, cout, tile_rows)
    ap = _nhwc_ptr(addend, "conv3x3 addend", torch.float32) if addend is not None else None
    with _timed(tag, 4 * (m * cin + (2 if addend is not None else 1) * m * cout + (m * cout if fuse is not None else 0)) + 9 * wb * cin * cout,
                18 * m * cin * cout, kernel="gemm_x6p_kernel (3x3)" if pst is None else "gemm_x6p_kernel<pair> (3x3)"):
        rc = lib().peclr_conv3x3_x6p_f32(nb, h, w, cin, cout, xp, _ptr(planes, torch.uint8), y.data_ptr(), ap, int(flip), tile_rows,
                                         _CONV3X3_VARIANT if variant is None else int(variant), _zeros(x.device).data_ptr(), _ptr(stat_shift),
                                         partial.data_ptr() if stat_shift is not None else None,
                                         ctypes.byref(fuse) if fuse is not None else None,
                                         ctypes.byref(pst) if pst is not None else None, _stream())
    _check(rc, "peclr_conv3x3_x6p_f32")
    return y if partial is None else (y, partial, ns)


def conv3x3_s2_dgrad_x6p(gy: torch.Tensor, planes: torch.Tensor, cin: int, tag: str = "conv3x3_s2_dgrad", tile_rows: int = 0, bn_bwd=None,
                         pair=None):
    """Input gradient of a 3x3 / padding-1 / stride-2 convolution: gy [N, Cout, Ho, Wo] channels_last fp32 -> dx
    [N, cin, 2 Ho, 2 Wo], one implicit GEMM per parity class of input pixels (1, 2, 2 and 4 of the nine taps:
    peclr_conv3x3_s2_dgrad_x6p_f32); `planes` as for the stride-1 input gradient ([Cout * 9, Cin] packed with
    transposed = 9).  bn_bwd: as in gemm_x6p (the BatchNorm layer dx arrives at) -> (dx, partial, n_split)."""
    nb, cout, ho, wo = gy.shape
    gp = _nhwc_ptr(gy, "conv_s2 dgrad gy", torch.float32)
    pst, wb = _pair_arg(pair, "conv3x3_s2_dgrad_x6p")
    if planes.dtype != torch.uint8 or planes.numel() != wb * ((cin + 127) // 128 * 128) * 9 * cout:
        raise PeclrHipError(f"conv3x3_s2_dgrad_x6p: planes of {planes.numel()} bytes for [{cin}, 9 * {cout}]")
    dx = torch.empty((nb, cin, 2 * ho, 2 * wo), device=gy.device, dtype=torch.float32, memory_format=torch.channels_last)
    mc = nb * ho * wo
    partial, ns, fuse = None, 0, None
    if bn_bwd is not None:
        tile_rows = tile_rows or lib().peclr_gemm_x6p_tile_rows(mc, cin, 4 * cout)
        fuse, partial, ns = _bn_bwd_fuse(bn_bwd, 4 * mc, cin, tile_rows, groups=4)
    with _timed(tag, 4 * (mc * cout + 4 * mc * cin * (2 if fuse is not None else 1)) + 9 * wb * cin * cout, 18 * mc * cin * cout,
                kernel="gemm_x6p_kernel (3x3)" if pst is None else "gemm_x6p_kernel<pair> (3x3)"):
        rc = lib().peclr_conv3x3_s2_dgrad_x6p_f32(nb, ho, wo, cout, cin, gp, _ptr(planes, torch.uint8), dx.data_ptr(), tile_rows,
                                                  _zeros(gy.device).data_ptr(), ctypes.byref(fuse) if fuse is not None else None,
                                                  ctypes.byref(pst) if pst is not None else None, _stream())
    _check(rc, "peclr_conv3x3_s2_dgrad_x6p_f32")
    return dx if partial is None else (dx, partial, ns)


def conv_s2_x6p(x: torch.Tensor, planes: torch.Tensor, cout: int, taps: int, tag: str = "conv_s2_x6p", tile_rows: int = 0,
                stat_shift: Optional[torch.Tensor] = None, pair=None):
    """Forward of a stride-2 convolution (taps = 9: 3x3 / padding 1; taps = 1: 1x1) of an NHWC fp32 tensor x [N, Cin, H, W]
    (H, W even) on the six-product kernel (peclr_conv_s2_x6p_f32) -> y [N, cout, H/2, W/2] channels_last (and
    (partial, n_split) of the output's BatchNorm statistics when stat_shift is given)."""
    nb, cin, h, w = x.shape
    xp = _nhwc_ptr(x, "conv_s2 x", torch.float32)
    pst, wb = _pair_arg(pair, "conv_s2_x6p")
    if planes.dtype != torch.uint8 or planes.numel() != wb * ((cout + 127) // 128 * 128) * taps * cin or h % 2 or w % 2:
        raise PeclrHipError(f"conv_s2_x6p: planes of {planes.numel()} bytes for [{cout}, {taps} * {cin}], input {h} x {w}")
    y = torch.empty((nb, cout, h // 2, w // 2), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    m = nb * (h // 2) * (w // 2)
    partial, ns = None, 0
    if stat_shift is not None:
        tile_rows = tile_rows or lib().peclr_gemm_x6p_tile_rows(m, cout, taps * cin)
        ns = (m + tile_rows - 1) // tile_rows
        partial = torch.empty((2 * ns + 1, cout), device=x.device, dtype=torch.float32)
    with _timed(tag, 4 * (nb * h * w * cin + m * cout) + wb * taps * cin * cout, 2 * m * taps * cin * cout,
                kernel="gemm_x6p_kernel (stride 2)" if pst is None else "gemm_x6p_kernel<pair> (stride 2)"):
        rc = lib().peclr_conv_s2_x6p_f32(nb, h, w, cin, cout, taps, xp, _ptr(planes, torch.uint8), y.data_ptr(), tile_rows,
                                         _zeros(x.device).data_ptr(), _ptr(stat_shift), _ptr(partial),
                                         ctypes.byref(pst) if pst is not None else None, _stream())
    _check(rc, "peclr_conv_s2_x6p_f32")
    return y if stat_shift is None else (y, partial, ns)


def gemm_x6_tn(a: torch.Tensor, b: torch.Tensor, tag: str = "gemm_x6_tn") -> torch.Tensor:
    """C[M,N] (fp32) = A[K,M]^T . B[K,N], fp32 row-major contiguous 2-D HIP tensors whose ROWS are the contraction
    index (the 1x1 weight gradient dW = dY^T X on NHWC storage), on the bf16 matrix cores at fp32 accuracy; split-K
    slabs summed in a fixed order (peclr_gemm_x6_tn_f32 + peclr_slab_reduce_f32): deterministic."""
    (k, m), (k2, n) = a.shape, b.shape
    if k != k2:
        raise PeclrHipError(f"gemm_x6_tn: shapes {tuple(a.shape)}^T x {tuple(b.shape)}")
    ns = lib().peclr_gemm_x6_tn_slabs(m, n, k)
    if ns < 1:
        raise PeclrHipError(f"gemm_x6_tn: unsupported shape M={m} N={n} K={k}")
    slabs = torch.empty((ns, m, n), device=a.device, dtype=torch.float32)
    with _timed(tag, 4 * (k * m + k * n + ns * m * n), 2 * m * n * k, kernel="gemm_x6_tn128_kernel"):
        rc = lib().peclr_gemm_x6_tn_f32(m, n, k, _ptr(a), m, _ptr(b), n, slabs.data_ptr(), ns, _stream())
    _check(rc, "peclr_gemm_x6_tn_f32")
    return slabs[0] if ns == 1 else slab_reduce(slabs, tag="wgrad_slab_reduce")


def gemm_x6t(a: torch.Tensor, b: torch.Tensor, taps: int = 1, hw=None, stride: int = 1, tag: str = "gemm_x6t") -> torch.Tensor:
    """C[M, taps * N] (fp32) = sum over rows of A[K, M]^T . B[K (shifted by the tap), N] -- the weight gradient of a 1x1
    (taps = 1) or 3x3 / padding-1 (taps = 9, hw = (H, W) of the images A's rows are the pixels of) convolution on NHWC
    storage, on the bf16 matrix cores at fp32 accuracy (peclr_gemm_x6t_f32 + peclr_slab_reduce_f32: fixed-order split-K,
    deterministic).  stride = 2: A = dY over the H x W output pixels, B = X over the 2H x 2W input pixels (4 K rows).
    (Weight gradients stay on six products: in pair arithmetic they were 21 - 29 % faster and, on real layer tensors, 3 - 46 % less
    accurate than this kernel -- tools/exp/pair_wgrad.patch)"""
    (k, m), (k2, n) = a.shape, b.shape
    if k * stride * stride != k2 or taps not in (1, 9) or stride not in (1, 2) or ((taps == 9 or stride == 2) and hw is None):
        raise PeclrHipError(f"gemm_x6t: shapes {tuple(a.shape)}^T x {tuple(b.shape)}, taps {taps}, stride {stride}")
    h, w = hw if hw is not None else (1, 1)
    ns = lib().peclr_gemm_x6t_slabs(m, n, k, taps)
    if ns < 1:
        raise PeclrHipError(f"gemm_x6t: unsupported shape M={m} N={n} K={k}")
    slabs = torch.empty((ns, m, taps * n), device=a.device, dtype=torch.float32)
    with _timed(tag, 4 * (k * m + k2 * n // (stride * stride) * (1 if taps == 1 else stride * stride) + ns * m * n * taps),
                2 * m * n * k * taps, kernel="gemm_x6w2_kernel | gemm_x6w_kernel" if taps == 9 else "gemm_x6t2_kernel | gemm_x6t_kernel"):
        rc = lib().peclr_gemm_x6t_f32(m, n, k, _ptr(a), a.stride(0), _ptr(b), b.stride(0), slabs.data_ptr(), ns, taps, h, w, stride,
                                      _zeros(a.device).data_ptr(), _stream())
    _check(rc, "peclr_gemm_x6t_f32")
    return slabs[0] if ns == 1 else slab_reduce(slabs, tag="wgrad_slab_reduce")


def gemm_add_half(a: torch.Tensor, b_t: torch.Tensor, addend: Optional[torch.Tensor], tag: str = "gemm_add") -> torch.Tensor:
    """C (16-bit) = A[M,K] . B_t[N,K]^T + addend, row-major contiguous 2-D HIP tensors that are ALL bf16 or ALL
    fp16, fp32 accumulate (peclr_gemm_add_bf16 / peclr_gemm_add_f16)."""
    (m, k), (n, k2) = a.shape, b_t.shape
    half = a.dtype
    if half not in (torch.bfloat16, torch.float16):
        raise PeclrHipError(f"gemm_add_half: bf16 or fp16 tensors expected, got {half}")
    for t in (a, b_t) + ((addend,) if addend is not None else ()):
        if not t.is_cuda or t.dtype != half or not t.is_contiguous():
            raise PeclrHipError(f"gemm_add_half: contiguous {half} HIP tensors expected (peclr_amd has no CPU path)")
    if k != k2 or (addend is not None and tuple(addend.shape) != (m, n)):
        raise PeclrHipError(f"gemm_add_half: shapes {tuple(a.shape)} x {tuple(b_t.shape)}^T")
    out = torch.empty((m, n), device=a.device, dtype=half)
    fn = lib().peclr_gemm_add_bf16 if half == torch.bfloat16 else lib().peclr_gemm_add_f16
    with _timed(tag, 2 * (m * k + k * n + 2 * m * n), 2 * m * n * k):
        rc = fn(m, n, k, a.data_ptr(), k, b_t.data_ptr(), k, out.data_ptr(), n,
                addend.data_ptr() if addend is not None else None, n, _stream())
    _check(rc, "peclr_gemm_add_bf16" if half == torch.bfloat16 else "peclr_gemm_add_f16")
    return out


gemm_add_bf16 = gemm_add_half   # round-1 name


# ------------------------------------------------------------------ 16-bit convolutions (csrc/conv_h.hip)
_HALF_IO = {torch.bfloat16: 1, torch.float16: 2}      # PECLR_DTYPE_BF16 / _F16
_HZEROS = {}


def _hzeros(device, dtype):
    z = _HZEROS.get((device, dtype))
    if z is None:
        z = _HZEROS[(device, dtype)] = torch.zeros(64, device=device, dtype=dtype)
    return z


def _half_io(t: torch.Tensor, what: str) -> int:
    if t.dtype not in _HALF_IO or not t.is_cuda:
        raise PeclrHipError(f"{what}: a bf16 / fp16 HIP tensor expected, got {t.dtype} on {t.device} (peclr_amd has no CPU path)")
    return _HALF_IO[t.dtype]


class HPlanes:
    """Weight matrices packed once per optimiser step from the fp32 MASTER weights into 16-bit MFMA-fragment order
    (peclr_h_pack) for peclr_gemm_h / peclr_conv_h -- the cast autocast performs per forward rides in that launch.
    `specs`: list of (fp32 2-D HIP tensor W, transposed) exactly as X6Planes; dtype: torch.bfloat16 or torch.float16."""

    def __init__(self, specs, dtype):
        if not specs:
            raise PeclrHipError("HPlanes: nothing to pack")
        if dtype not in _HALF_IO:
            raise PeclrHipError(f"HPlanes: bf16 or fp16, got {dtype}")
        dev = specs[0][0].device
        self.dtype = dtype
        rows, self.planes, self.shapes, chunk = [], [], [], 0
        for w, transposed in specs:
            _ptr(w, what="16-bit pack: fp32 master weight")
            if w.dim() != 2:
                raise PeclrHipError("HPlanes: 2-D weight matrices expected")
            t = int(transposed)
            n, k = (w.shape[1], w.shape[0]) if t else (w.shape[0], w.shape[1])
            nbytes = lib().peclr_h_pack_bytes(n, k)
            if nbytes <= 0:
                raise PeclrHipError(f"HPlanes: B_t[{n}, {k}] needs n % 64 == 0 and k % 32 == 0")
            planes = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            rows.append([w.data_ptr(), planes.data_ptr(), n, k, w.stride(0), t, chunk, _HALF_IO[dtype]])
            chunk += ((n + 127) // 128) * (k // 32)
            self.planes.append(planes)
            self.shapes.append((n, k))
        self._sources = [w for w, _ in specs]
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.count, self.chunks = len(rows), chunk
        self.nbytes = sum(p.numel() for p in self.planes) + 4 * sum(w.numel() for w in self._sources)

    def pack(self):
        with _timed("h_pack", self.nbytes, kernel="h_pack_kernel"):
            rc = lib().peclr_h_pack(self.table.data_ptr(), self.count, self.chunks, _stream())
        _check(rc, "peclr_h_pack")
        return self


def _h_planes_ok(planes, n, k, what):
    if planes.dtype != torch.uint8 or planes.numel() != 2 * ((n + 127) // 128 * 128) * k:
        raise PeclrHipError(f"{what}: planes of {planes.numel()} bytes for B_t[{n}, {k}]")


def gemm_h(a: torch.Tensor, planes: torch.Tensor, n: int, addend: Optional[torch.Tensor] = None, tag: str = "gemm_h",
           tile_rows: int = 0, stat_shift: Optional[torch.Tensor] = None, bn_bwd=None, addend_s2=None,
           addend_mask: Optional[torch.Tensor] = None):
    """C (16-bit) [M, n] = A[M, K] . B_t^T (+ addend), A / C / addend bf16 or fp16, B_t packed by HPlanes, fp32 accumulation
    (peclr_gemm_h).  Arguments and returns as `gemm_x6p` (statistics / backward-reduction partials are fp32, in the same
    layouts; they are sums over the ROUNDED 16-bit outputs)."""
    m, k = a.shape
    io = _half_io(a, "gemm_h A")
    _h_planes_ok(planes, n, k, "gemm_h")
    add_rows = m if addend_s2 is None else m // 4
    if not a.is_contiguous() or (addend is not None and (tuple(addend.shape) != (add_rows, n) or addend.dtype != a.dtype or not addend.is_contiguous())):
        raise PeclrHipError(f"gemm_h: contiguous {a.dtype} A {tuple(a.shape)} / addend expected")
    if addend_s2 is not None and (addend is None or stat_shift is not None or addend_mask is not None):
        raise PeclrHipError("gemm_h: addend_s2 needs the compact addend (and has no statistics output / mask)")
    if addend_mask is not None and (addend is None or stat_shift is not None or n % 32 or addend_mask.dtype != torch.int32
                                    or addend_mask.numel() != m * (n // 32) or not addend_mask.is_contiguous()):
        raise PeclrHipError("gemm_h: addend_mask is the int32 [M, n / 32] bit mask of a dense addend (no statistics output)")
    out = torch.empty((m, n), device=a.device, dtype=a.dtype)
    partial, ns, fuse = None, 0, None
    if stat_shift is not None or bn_bwd is not None:
        tile_rows = tile_rows or lib().peclr_conv_h_tile_rows(m, n)
    if stat_shift is not None:
        if stat_shift.numel() != n:
            raise PeclrHipError(f"gemm_h: stat_shift has {stat_shift.numel()} entries for {n} columns")
        ns = (m + tile_rows - 1) // tile_rows
        partial = torch.empty((2 * ns + 1, n), device=a.device, dtype=torch.float32)
    elif bn_bwd is not None:
        fuse, partial, ns = _bn_bwd_fuse(bn_bwd, m, n, tile_rows, dtype=a.dtype)
    add_elems = 0 if addend is None else addend.numel()
    mask_bytes = 0 if addend_mask is None else 4 * addend_mask.numel()
    with _timed(tag, 2 * (m * k + m * n + add_elems + (m * n if fuse is not None else 0) + k * n) + mask_bytes, 2 * m * n * k,
                kernel="conv_h_kernel"):
        rc = lib().peclr_gemm_h(io, m, n, k, a.data_ptr(), k, _ptr(planes, torch.uint8), out.data_ptr(), n,
                                addend.data_ptr() if addend is not None else None, n,
                                int(addend_s2[0]) if addend_s2 is not None else 0, int(addend_s2[1]) if addend_s2 is not None else 0,
                                _ptr(addend_mask, torch.int32), tile_rows, _ptr(stat_shift),
                                partial.data_ptr() if stat_shift is not None else None,
                                ctypes.byref(fuse) if fuse is not None else None, _stream())
    _check(rc, "peclr_gemm_h")
    return out if partial is None else (out, partial, ns)


def conv_h(x: torch.Tensor, planes: torch.Tensor, cout: int, taps: int = 9, stride: int = 1, flip: bool = False,
           tag: str = "conv_h", tile_rows: int = 0, stat_shift: Optional[torch.Tensor] = None, bn_bwd=None):
    """3x3 / padding-1 (taps = 9) or 1x1 (taps = 1, stride 2) convolution of an NHWC bf16 / fp16 tensor x [N, Cin, H, W]
    (peclr_conv_h): forward, or -- flip=True, stride 1 -- the input gradient with x = dY and planes packed with
    transposed = 9.  Returns y [N, cout, H / stride, W / stride] channels_last (+ (partial, n_split) as gemm_h)."""
    nb, cin, h, w = x.shape
    io = _half_io(x, "conv_h x")
    xp = _nhwc_ptr(x, "conv_h x", x.dtype)
    _h_planes_ok(planes, cout, taps * cin, "conv_h")
    if h % stride or w % stride:
        raise PeclrHipError(f"conv_h: {h} x {w} input with stride {stride}")
    ho, wo = h // stride, w // stride
    y = torch.empty((nb, cout, ho, wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    m = nb * ho * wo
    partial, ns, fuse = None, 0, None
    if stat_shift is not None or bn_bwd is not None:
        ns = lib().peclr_conv_h_row_blocks(nb, h, w, cout, taps, stride, tile_rows)     # (tile_rows 0: the library's choice)
        if ns <= 0:
            raise PeclrHipError(f"conv_h: no launch for tile_rows = {tile_rows} at {h} x {w}, taps {taps}, stride {stride}")
    if stat_shift is not None:
        partial = torch.empty((2 * ns + 1, cout), device=x.device, dtype=torch.float32)
    elif bn_bwd is not None:
        fuse, partial, ns = _bn_bwd_fuse(bn_bwd, m, cout, tile_rows, dtype=x.dtype, row_blocks=ns)
    with _timed(tag, 2 * (nb * h * w * cin + m * cout * (2 if fuse is not None else 1) + taps * cin * cout), 2 * m * taps * cin * cout,
                kernel="conv_h_kernel (3x3)" if taps == 9 else "conv_h_kernel (stride 2)"):
        rc = lib().peclr_conv_h(io, nb, h, w, cin, cout, taps, stride, xp, _ptr(planes, torch.uint8), y.data_ptr(), int(flip), tile_rows,
                                _hzeros(x.device, x.dtype).data_ptr(), _ptr(stat_shift),
                                partial.data_ptr() if stat_shift is not None else None,
                                ctypes.byref(fuse) if fuse is not None else None, _stream())
    _check(rc, "peclr_conv_h")
    return y if partial is None else (y, partial, ns)


def wgrad_h_ok(gy: torch.Tensor, x: torch.Tensor, taps: int, stride: int) -> bool:
    """Does peclr_wgrad_h take this weight gradient?  (1x1 convolutions, stride 1 or 2, channel counts multiples of 32.)"""
    cout, cin = gy.shape[1], x.shape[1]
    if taps == 9:       # 3x3 / padding 1 / stride 1 (peclr_wgrad3_h); the stride-2 ones stay on MIOpen (faster there)
        return (stride == 1 and cout % 64 == 0 and cin % 64 == 0 and gy.dtype in _HALF_IO and x.dtype == gy.dtype
                and x.shape[2] == stride * gy.shape[2] and x.shape[3] == stride * gy.shape[3] and gy.shape[3] <= 62
                and gy.shape[0] * gy.shape[2] * gy.shape[3] >= 512)
    return (taps == 1 and stride in (1, 2) and cout % 32 == 0 and cin % 32 == 0 and gy.dtype in _HALF_IO and x.dtype == gy.dtype
            and gy.shape[0] * gy.shape[2] * gy.shape[3] >= 32
            and (stride == 1 or (x.shape[2] == 2 * gy.shape[2] and x.shape[3] == 2 * gy.shape[3])))


def wgrad_h(gy: torch.Tensor, x: torch.Tensor, taps: int = 1, stride: int = 1, tag: str = "conv1x1_wgrad") -> torch.Tensor:
    """dW [Cout, taps * Cin] (fp32) of a 1x1 (taps = 1) or 3x3 / padding-1 (taps = 9) convolution, stride 1 or 2, from 16-bit NHWC
    activations gy [N, Cout, Ho, Wo], x [N, Cin, H, W] (peclr_wgrad_h / peclr_wgrad3_h +
    peclr_slab_reduce_f32: fixed-order split-K, deterministic)."""
    if not wgrad_h_ok(gy, x, taps, stride):
        raise PeclrHipError(f"wgrad_h: unsupported problem gy {tuple(gy.shape)} x {tuple(x.shape)} taps {taps} stride {stride}")
    io = _half_io(gy, "wgrad_h gy")
    nb, cout, ho, wo = gy.shape
    cin = x.shape[1]
    gp, xp = _nhwc_ptr(gy, "wgrad_h gy", gy.dtype), _nhwc_ptr(x, "wgrad_h x", gy.dtype)
    if taps == 9:
        ns = lib().peclr_wgrad3_h_slabs(cout, cin, nb, ho, wo)
        if ns < 1:
            raise PeclrHipError(f"wgrad_h: unsupported 3x3 shape M={cout} N={cin} {nb} x {ho} x {wo}")
        slabs = torch.empty((ns, cout, 9 * cin), device=gy.device, dtype=torch.float32)
        with _timed("conv3x3_wgrad" if tag == "conv1x1_wgrad" else tag, 2 * nb * ho * wo * (cout + cin) + 4 * ns * cout * 9 * cin,
                    18 * cout * cin * nb * ho * wo, kernel="wgrad3_h_kernel"):
            rc = lib().peclr_wgrad3_h(io, cout, cin, nb, ho, wo, gp, xp, slabs.data_ptr(), ns, _hzeros(gy.device, gy.dtype).data_ptr(), _stream())
        _check(rc, "peclr_wgrad3_h")
        return slabs[0] if ns == 1 else slab_reduce(slabs, tag="wgrad_slab_reduce")
    k = nb * ho * wo
    ns = lib().peclr_wgrad_h_slabs(cout, cin, k)
    if ns < 1:
        raise PeclrHipError(f"wgrad_h: unsupported shape M={cout} N={cin} K={k}")
    slabs = torch.empty((ns, cout, cin), device=gy.device, dtype=torch.float32)
    with _timed(tag, 2 * k * (cout + cin) + 4 * ns * cout * cin, 2 * cout * cin * k, kernel="wgrad_h_kernel"):
        rc = lib().peclr_wgrad_h(io, cout, cin, k, gp, cout, xp, cin, slabs.data_ptr(), ns, stride, ho, wo,
                                 _hzeros(gy.device, gy.dtype).data_ptr(), _stream())
    _check(rc, "peclr_wgrad_h")
    return slabs[0] if ns == 1 else slab_reduce(slabs, tag="wgrad_slab_reduce")


def conv3x3_s2_dgrad_h(gy: torch.Tensor, planes: torch.Tensor, cin: int, tag: str = "conv3x3_s2_dgrad", tile_rows: int = 0, bn_bwd=None):
    """Input gradient of a 3x3 / padding-1 / stride-2 convolution, 16-bit: gy [N, Cout, Ho, Wo] channels_last -> dx
    [N, cin, 2 Ho, 2 Wo] (peclr_conv3x3_s2_dgrad_h: one implicit GEMM per parity class of input pixels)."""
    nb, cout, ho, wo = gy.shape
    io = _half_io(gy, "conv3x3_s2_dgrad_h gy")
    gp = _nhwc_ptr(gy, "conv3x3_s2_dgrad_h gy", gy.dtype)
    _h_planes_ok(planes, cin, 9 * cout, "conv3x3_s2_dgrad_h")
    dx = torch.empty((nb, cin, 2 * ho, 2 * wo), device=gy.device, dtype=gy.dtype, memory_format=torch.channels_last)
    mc = nb * ho * wo
    partial, ns, fuse = None, 0, None
    if bn_bwd is not None:
        tile_rows = tile_rows or lib().peclr_conv_h_tile_rows(mc, cin)
        fuse, partial, ns = _bn_bwd_fuse(bn_bwd, 4 * mc, cin, tile_rows, groups=4, dtype=gy.dtype)
    with _timed(tag, 2 * (mc * cout + 4 * mc * cin * (2 if fuse is not None else 1) + 9 * cin * cout), 18 * mc * cin * cout,
                kernel="conv_h_kernel (3x3)"):
        rc = lib().peclr_conv3x3_s2_dgrad_h(io, nb, ho, wo, cout, cin, gp, _ptr(planes, torch.uint8), dx.data_ptr(), tile_rows,
                                            _hzeros(gy.device, gy.dtype).data_ptr(), ctypes.byref(fuse) if fuse is not None else None,
                                            _stream())
    _check(rc, "peclr_conv3x3_s2_dgrad_h")
    return dx if partial is None else (dx, partial, ns)



# ------------------------------------------------------------------ backbone glue: BN2d (+add) (+ReLU), NHWC
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
# ------------------------------------------------------------------ stem (csrc/stem.hip)
STEM_FMT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}      # output / arithmetic of peclr_stem_conv7x7_s2


class StemPlanes:
    """The 7x7 stem filter [64, 3, 7, 7] (any strides) in the fragment order of peclr_stem_conv7x7_s2: three bf16 planes for
    fp32 runs (six products), one bf16 / fp16 plane for autocast runs.  `pack()` is one launch; call it after the weight changed."""

    def __init__(self, weight: torch.Tensor, dtype=torch.float32):
        if tuple(weight.shape) != (64, 3, 7, 7) or weight.dtype != torch.float32 or not weight.is_cuda:
            raise PeclrHipError(f"StemPlanes: an fp32 HIP [64, 3, 7, 7] filter expected, got {weight.dtype} {tuple(weight.shape)} on {weight.device}")
        self.fmt = STEM_FMT[dtype]
        self.weight = weight
        self.planes = torch.empty(lib().peclr_stem_pack_bytes(self.fmt), device=weight.device, dtype=torch.uint8)

    def pack(self):
        w = self.weight
        with _timed("stem_pack", 4 * w.numel() + self.planes.numel(), kernel="stem_pack_kernel"):
            rc = lib().peclr_stem_pack(w.data_ptr(), *w.stride(), self.planes.data_ptr(), self.fmt, _stream())
        _check(rc, "peclr_stem_pack")
        return self


def stem_conv(x: torch.Tensor, planes: "StemPlanes", stat_shift: Optional[torch.Tensor] = None, tag: str = "stem_fwd"):
    """y [N, 64, H/2, W/2] channels_last = conv2d(x, W, stride 2, padding 3) for fp32 channels_last images x [N, 3, H, W]
    (peclr_stem_conv7x7_s2); y is fp32 (fp32 accuracy: six bf16 products) or bf16 / fp16 (one product of the rounded operands:
    autocast) as `planes` was packed.  stat_shift (fp32 [64]): also the training statistics of y as (partial, n_split) in the
    layout of peclr_bn2d_stats -> (y, partial, n_split)."""
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous(memory_format=torch.channels_last):
        raise PeclrHipError(f"stem_conv: fp32 channels_last HIP images [N, 3, H, W] expected, got {x.dtype} {tuple(x.shape)} on {x.device} "
                            "(peclr_amd has no CPU path)")
    n, _, h, w = x.shape
    ns = lib().peclr_stem_workgroups(n, h, w)
    if ns < 1:
        raise PeclrHipError(f"stem_conv: unsupported image size {h} x {w}")
    out_dtype = {v: k for k, v in STEM_FMT.items()}[planes.fmt]
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, 64, ho, wo), device=x.device, dtype=out_dtype, memory_format=torch.channels_last)
    partial = None
    if stat_shift is not None:
        if stat_shift.numel() != 64:
            raise PeclrHipError("stem_conv: stat_shift has 64 entries")
        partial = torch.empty((2 * ns + 1, 64), device=x.device, dtype=torch.float32)
    e = y.element_size()
    k_mfma = 14 * 16                                        # the contraction the matrix cores run (147 padded to 224)
    with _timed(tag, 4 * x.numel() + e * y.numel() + planes.planes.numel(), 2 * n * ho * wo * 64 * 147,
                kernel="stem_fwd_kernel"):
        rc = lib().peclr_stem_conv7x7_s2(x.data_ptr(), n, h, w, planes.planes.data_ptr(), planes.fmt, y.data_ptr(), _ptr(stat_shift),
                                         partial.data_ptr() if partial is not None else None, _stream())
    _check(rc, "peclr_stem_conv7x7_s2")
    del k_mfma
    return y if partial is None else (y, partial, ns)


def stem_wgrad(gy: torch.Tensor, x: torch.Tensor, tag: str = "stem_wgrad") -> torch.Tensor:
    """dW [64, 3, 7, 7] (fp32; a view of a [64, 7, 8, 4] buffer: the layout the kernel accumulates in) of the stem convolution
    for channels_last gy [N, 64, H/2, W/2] (fp32, or bf16 / fp16 under autocast) and the fp32 channels_last images x [N, 3, H, W]
    (peclr_stem_wgrad + peclr_slab_reduce_f32: fixed-order slabs, deterministic)."""
    if (x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous(memory_format=torch.channels_last)
            or gy.dtype not in STEM_FMT or gy.dim() != 4 or gy.shape[1] != 64 or not gy.is_contiguous(memory_format=torch.channels_last)):
        raise PeclrHipError(f"stem_wgrad: fp32 channels_last images and a channels_last [N, 64, H/2, W/2] gradient expected, got "
                            f"{x.dtype} {tuple(x.shape)}, {gy.dtype} {tuple(gy.shape)} (peclr_amd has no CPU path)")
    n, _, h, w = x.shape
    if tuple(gy.shape) != (n, 64, (h - 1) // 2 + 1, (w - 1) // 2 + 1):
        raise PeclrHipError(f"stem_wgrad: gradient {tuple(gy.shape)} for images {tuple(x.shape)}")
    fmt = STEM_FMT[gy.dtype]
    ns = lib().peclr_stem_wgrad_slabs(n, h, w, fmt)
    if ns < 1:
        raise PeclrHipError(f"stem_wgrad: unsupported image size {h} x {w}")
    slabs = torch.empty((ns, 64, 224), device=x.device, dtype=torch.float32)
    with _timed(tag, 4 * x.numel() + gy.element_size() * gy.numel() + 4 * slabs.numel(), 2 * gy.numel() * 147, kernel="stem_wgrad_kernel"):
        rc = lib().peclr_stem_wgrad(x.data_ptr(), gy.data_ptr(), n, h, w, fmt, slabs.data_ptr(), ns, _stream())
    _check(rc, "peclr_stem_wgrad")
    dw = slabs[0] if ns == 1 else slab_reduce(slabs, tag="wgrad_slab_reduce")
    return dw.view(64, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2)          # [n][kh][kw][c] -> [n][c][kh][kw]


_IO = {torch.float32: (DTYPE_F32, 4), torch.bfloat16: (DTYPE_BF16, 2), torch.float16: (DTYPE_F16, 2)}


def _nhwc_ptr(t: torch.Tensor, what: str, dtype=None):
    if not t.is_cuda:
        raise PeclrHipError(f"{what}: expected a HIP device tensor (peclr_amd has no CPU path)")
    if t.dtype not in _IO or t.dim() != 4 or (dtype is not None and t.dtype != dtype):
        raise PeclrHipError(f"{what}: expected a 4-D {dtype or 'fp32/bf16/fp16'} tensor, got {t.dtype} {tuple(t.shape)}")
    if not t.is_contiguous(memory_format=torch.channels_last):
        raise PeclrHipError(f"{what}: tensor must be channels_last (NHWC) contiguous")
    return t.data_ptr()


def bn2d_n_split(r: int, c: int, io: int) -> int:
    n = lib().peclr_bn2d_n_split(r, c, io)
    if n < 1:
        raise PeclrHipError(f"fused BatchNorm2d: unsupported shape R={r} C={c} (C must be a ResNet width)")
    return n


def _sync_totals(partial, ns, c, rows, group):
    """Synchronised BatchNorm: combine this rank's slice partials into double [2][C] totals, append the
    row count and SUM-all-reduce over `group`.  Returns (local totals, global totals, global rows)."""
    import torch.distributed as td

    local = torch.empty(2 * c + 1, device=partial.device, dtype=torch.float64)
    with _timed("bn2d_combine", 8 * ns * c):
        rc = lib().peclr_bn2d_combine_f64(partial.data_ptr(), ns, c, local.data_ptr(), _stream())
    _check(rc, "peclr_bn2d_combine_f64")
    local[2 * c] = float(rows)
    total = local.clone()
    td.all_reduce(total, op=td.ReduceOp.SUM, group=group)
    return local, total


def _bn2d_scale_shift(x, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, sync_group=None, sync_shift=None,
                      pre=None):
    """stats -> finalize: (save [mean, invstd], scale_shift) for x [N,C,H,W] NHWC; updates the running
    statistics in training mode."""
    n, c, h, w = x.shape
    r = n * h * w
    dev = x.device
    xp = _nhwc_ptr(x, "bn2d x")
    io, e = _IO[x.dtype]
    save = torch.empty((2, c), device=dev, dtype=torch.float32)
    ss = torch.empty((2, c), device=dev, dtype=torch.float32)
    part_ptr, ns = None, 0
    sync = sync_group is not None
    if training and pre is not None:
        # the producer of x (a GEMM epilogue) already summed the statistics per row block: `pre` = (partial, n_split,
        # shift it used) in peclr_bn2d_stats' layout; nothing re-reads x here
        partial, ns, shift = pre
        if tuple(partial.shape) != (2 * ns + 1, c):
            raise PeclrHipError(f"bn2d: precomputed statistics of shape {tuple(partial.shape)} for C = {c}, n_split = {ns}")
        part_ptr = _ptr(partial, what="bn2d partial statistics")
    elif training:
        ns = bn2d_n_split(r, c, io)
        partial = torch.empty((2 * ns + 1, c), device=dev, dtype=torch.float32)
        part_ptr = partial.data_ptr()
        if sync and running_mean is None and sync_shift is None:
            raise PeclrHipError("synchronised BatchNorm needs running statistics (their mean is the common shift)")
        # synchronised: every rank must subtract the SAME shift before summing -> the (replicated) running mean, or the
        # copy of it the caller kept (re-run of a checkpointed block: the running statistics have moved since)
        shift = (sync_shift if sync_shift is not None else running_mean.detach().clone()) if sync else None
        with _timed("bn2d_stats", e * r * c):
            rc = lib().peclr_bn2d_stats(xp, io, r, c, _ptr(shift), part_ptr, ns, _stream())
        _check(rc, "peclr_bn2d_stats")
    if training and sync:
        _, total = _sync_totals(partial, ns, c, r, sync_group)
        with _timed("bn2d_finalize", 16 * c):
            rc = lib().peclr_bn2d_finalize_totals_f32(total.data_ptr(), shift.data_ptr(), c, eps, momentum,
                                                      _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                                      _ptr(nbt, torch.int64, "num_batches_tracked"),
                                                      save[0].data_ptr(), save[1].data_ptr(), ss.data_ptr(), _stream())
        _check(rc, "peclr_bn2d_finalize_totals_f32")
    else:
        with _timed("bn2d_finalize", 8 * ns * c):
            rc = lib().peclr_bn2d_finalize_f32(part_ptr, ns, r, c, int(training), eps, momentum, _ptr(gamma), _ptr(beta),
                                               _ptr(running_mean), _ptr(running_var),
                                               _ptr(nbt, torch.int64, "num_batches_tracked") if training else None,
                                               save[0].data_ptr(), save[1].data_ptr(), ss.data_ptr(), _stream())
        _check(rc, "peclr_bn2d_finalize_f32")
    return save, ss


_ABSMAX_SLABS: dict = {}


def absmax_slot(device) -> torch.Tensor:
    """A zeroed one-element fp32 tensor for a pass's `absmax_out` (the "pair" GEMMs' peclr_x6_pair.a_absmax).  Slots are views of a
    256-float slab that ONE fill launch zeroes; a new slab is taken when the current one is used up -- and whenever a stream capture
    begins, so that the fill is part of the captured graph and every replay starts from zeros."""
    key = (device, capture_id())
    slab = _ABSMAX_SLABS.get(key)
    if slab is None or slab[1] >= slab[0].numel():
        for k in [k for k in _ABSMAX_SLABS if k[0] == device and k != key]:
            del _ABSMAX_SLABS[k]                       # (slots handed out stay alive through their views)
        slab = _ABSMAX_SLABS[key] = [torch.zeros(256, device=device, dtype=torch.float32), 0]
    i = slab[1]
    slab[1] += 1
    return slab[0][i:i + 1]


def _absmax_ptr(absmax, x):
    if absmax is None:
        return None
    if x.dtype != torch.float32 or not (absmax.is_cuda and absmax.dtype == torch.float32 and absmax.numel() == 1):
        raise PeclrHipError("absmax: a zeroed one-element fp32 HIP tensor, for fp32 passes")
    return absmax.data_ptr()


def bn2d_apply(x, ss, relu: bool = True, absmax=None):
    """y = (relu)(fmaf(x, scale, shift)) from a finished scale / shift table: the apply pass of `bn2d_fwd` on its own (the
    fallback of a layer whose apply was left to its consumer, `bn2d_fwd(..., apply=False)`)."""
    n, c, h, w = x.shape
    r = n * h * w
    io, e = _IO[x.dtype]
    y = torch.empty_like(x, memory_format=torch.channels_last)
    with _timed("bn2d_apply", 2 * e * r * c):
        rc = lib().peclr_bn2d_apply(_nhwc_ptr(x, "bn2d x"), None, io, r, c, ss.data_ptr(), int(relu), y.data_ptr(), None,
                                    _absmax_ptr(absmax, x), _stream())
    _check(rc, "peclr_bn2d_apply")
    return y


def bn2d_fwd(x, residual, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, relu,
             want_mask=False, sync_group=None, sync_shift=None, pre=None, apply=True, residual_bn=None, absmax=None):
    """want_mask: also write the 1-bit ReLU mask ([R, C/32] int32) the backward reads instead of y.
    absmax: a zeroed one-element fp32 tensor (`absmax_slot`) that receives max |y|.
    residual_bn = (x_s, scale_shift_s) instead of `residual`: the residual is the output of the shortcut's BatchNorm2d, which
    was not written -- this pass computes it from that layer's input and table (peclr_bn2d_apply_res_bn).
    sync_group: a process group -> training statistics are those of the rows of ALL its ranks
    (mean/var of the global batch, as one device holding the concatenated batch would compute)."""
    n, c, h, w = x.shape
    r = n * h * w
    dev = x.device
    xp = _nhwc_ptr(x, "bn2d x")
    io, e = _IO[x.dtype]
    save, ss = _bn2d_scale_shift(x, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, sync_group, sync_shift, pre)
    if not apply:                        # statistics and the table only: the consumer applies them in its operand path
        if residual is not None or want_mask:
            raise PeclrHipError("bn2d_fwd(apply=False): plain BatchNorm (+ ReLU) layers only")
        return None, save, ss, None
    y = torch.empty_like(x, memory_format=torch.channels_last)
    mask = torch.empty((r, c // 32), device=dev, dtype=torch.int32) if (want_mask and relu and c % 32 == 0) else None
    if residual_bn is not None:
        xs, ss_s = residual_bn
        if residual is not None or tuple(xs.shape) != tuple(x.shape) or ss_s.numel() != 2 * c or ss_s.dtype != torch.float32:
            raise PeclrHipError("bn2d_fwd: residual_bn = (input of the shortcut's BatchNorm, its fp32 [2, C] table), no residual tensor")
        with _timed("bn2d_apply", 3 * e * r * c + (r * c // 8 if mask is not None else 0)):
            rc = lib().peclr_bn2d_apply_res_bn(xp, _nhwc_ptr(xs, "bn2d shortcut x", x.dtype), ss_s.data_ptr(), io, r, c, ss.data_ptr(),
                                               int(relu), y.data_ptr(), mask.data_ptr() if mask is not None else None,
                                               _absmax_ptr(absmax, x), _stream())
        _check(rc, "peclr_bn2d_apply_res_bn")
        return y, save, ss, mask
    with _timed("bn2d_apply", (3 if residual is not None else 2) * e * r * c + (r * c // 8 if mask is not None else 0)):
        rc = lib().peclr_bn2d_apply(xp, _nhwc_ptr(residual, "bn2d residual", x.dtype) if residual is not None else None,
                                    io, r, c, ss.data_ptr(), int(relu), y.data_ptr(),
                                    mask.data_ptr() if mask is not None else None, _absmax_ptr(absmax, x), _stream())
    _check(rc, "peclr_bn2d_apply")
    return y, save, ss, mask


def bn2d_bwd(dy, x, y, mask, save, ss, training, relu, want_dres, sync_group=None, pre=None, absmax=None):
    """ReLU mask source: `mask` (bit mask from the forward) > `y` (forward output) > recomputed from x.
    sync_group: as in bn2d_fwd; dgamma/dbeta stay this rank's local sums (the gradient all-reduce sums
    them later), dx uses the global sums."""
    n, c, h, w = x.shape
    r = n * h * w
    dev = x.device
    io, e = _IO[x.dtype]
    dparams = torch.empty((2, c), device=dev, dtype=torch.float32)  # dgamma, dbeta
    coef = torch.empty((2, c), device=dev, dtype=torch.float32)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dres = torch.empty_like(x, memory_format=torch.channels_last) if want_dres else None
    dyp, xp = _nhwc_ptr(dy, "bn2d dy", x.dtype), _nhwc_ptr(x, "bn2d x")
    yp = _nhwc_ptr(y, "bn2d y", x.dtype) if (y is not None and mask is None) else None
    mp = _ptr(mask, torch.int32, "relu mask")
    extra = (r * c // 8) if mask is not None else (e * r * c if yp is not None else 0)
    if pre is not None:
        # the GEMM that produced dy already reduced it against this layer's x in its epilogue (peclr_bn_bwd_fuse)
        partial, ns = pre
        if tuple(partial.shape) != (2 * ns, c):
            raise PeclrHipError(f"bn2d backward: precomputed reduction of shape {tuple(partial.shape)} for C = {c}, n_split = {ns}")
    else:
        ns = bn2d_n_split(r, c, io)
        partial = torch.empty((2 * ns, c), device=dev, dtype=torch.float32)
        with _timed("bn2d_bwd_reduce", 2 * e * r * c + extra):
            rc = lib().peclr_bn2d_bwd_reduce(dyp, xp, yp, mp, io, r, c, int(relu), save[0].data_ptr(), save[1].data_ptr(),
                                             ss.data_ptr(), partial.data_ptr(), ns, _stream())
        _check(rc, "peclr_bn2d_bwd_reduce")
    if training and sync_group is not None:
        local, total = _sync_totals(partial, ns, c, r, sync_group)
        with _timed("bn2d_bwd_finalize", 32 * c):
            rc = lib().peclr_bn2d_bwd_finalize_totals_f32(local.data_ptr(), total.data_ptr(), c, 1, ss.data_ptr(),
                                                          dparams[0].data_ptr(), dparams[1].data_ptr(), coef.data_ptr(),
                                                          _stream())
        _check(rc, "peclr_bn2d_bwd_finalize_totals_f32")
    else:
        with _timed("bn2d_bwd_finalize", 8 * ns * c):
            rc = lib().peclr_bn2d_bwd_finalize_f32(partial.data_ptr(), ns, r, c, int(training), ss.data_ptr(),
                                                   dparams[0].data_ptr(), dparams[1].data_ptr(), coef.data_ptr(),
                                                   _stream())
        _check(rc, "peclr_bn2d_bwd_finalize_f32")
    with _timed("bn2d_bwd_apply", (3 + (1 if want_dres else 0)) * e * r * c + extra):
        rc = lib().peclr_bn2d_bwd_apply(dyp, xp, yp, mp, io, r, c, int(relu), save[0].data_ptr(), save[1].data_ptr(),
                                        ss.data_ptr(), coef.data_ptr(), dx.data_ptr(),
                                        dres.data_ptr() if dres is not None else None, _absmax_ptr(absmax, x), _stream())
    _check(rc, "peclr_bn2d_bwd_apply")
    return dx, dparams[0], dparams[1], dres


def bn2d_avgpool_fwd(x, residual, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, sync_group=None,
                     sync_shift=None, pre=None):
    """Encoder tail: mean over H x W of relu(bn(x) + residual) as fp32 [N, C]; the activation itself is not
    written.  Returns (pooled, relu mask, save, scale_shift)."""
    n, c, h, w = x.shape
    r = n * h * w
    if c % 32 or residual is None:
        raise PeclrHipError("fused BN + add + ReLU + average pool needs a residual and C % 32 == 0")
    io, e = _IO[x.dtype]
    save, ss = _bn2d_scale_shift(x, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, sync_group, sync_shift, pre)
    pooled = torch.empty((n, c), device=x.device, dtype=torch.float32)
    mask = torch.empty((r, c // 32), device=x.device, dtype=torch.int32)
    with _timed("bn2d_apply_avgpool", 2 * e * r * c + r * c // 8 + 4 * n * c):
        rc = lib().peclr_bn2d_apply_avgpool(_nhwc_ptr(x, "bn2d x"), _nhwc_ptr(residual, "bn2d residual", x.dtype), io, n,
                                            h * w, c, ss.data_ptr(), pooled.data_ptr(), mask.data_ptr(), _stream())
    _check(rc, "peclr_bn2d_apply_avgpool")
    return pooled, mask, save, ss


def bn2d_avgpool_bwd(d_pooled, x, mask, save, ss, training, sync_group=None, absmax=None):
    """Backward of bn2d_avgpool_fwd from the fp32 [N, C] gradient of the pooled output: (dx, dgamma, dbeta,
    d_residual)."""
    n, c, h, w = x.shape
    r = n * h * w
    io, e = _IO[x.dtype]
    dev = x.device
    ns = bn2d_n_split(r, c, io)
    partial = torch.empty((2 * ns, c), device=dev, dtype=torch.float32)
    dparams = torch.empty((2, c), device=dev, dtype=torch.float32)
    coef = torch.empty((2, c), device=dev, dtype=torch.float32)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dres = torch.empty_like(x, memory_format=torch.channels_last)
    dp = _ptr(d_pooled, what="d_pooled")
    xp, mp = _nhwc_ptr(x, "bn2d x"), _ptr(mask, torch.int32, "relu mask")
    with _timed("bn2d_bwd_reduce_avgpool", e * r * c + r * c // 8 + 4 * n * c):
        rc = lib().peclr_bn2d_bwd_reduce_avgpool(dp, xp, mp, io, n, h * w, c, save[0].data_ptr(), save[1].data_ptr(),
                                                 ss.data_ptr(), partial.data_ptr(), ns, _stream())
    _check(rc, "peclr_bn2d_bwd_reduce_avgpool")
    if training and sync_group is not None:
        local, total = _sync_totals(partial, ns, c, r, sync_group)
        with _timed("bn2d_bwd_finalize", 32 * c):
            rc = lib().peclr_bn2d_bwd_finalize_totals_f32(local.data_ptr(), total.data_ptr(), c, 1, ss.data_ptr(),
                                                          dparams[0].data_ptr(), dparams[1].data_ptr(), coef.data_ptr(),
                                                          _stream())
        _check(rc, "peclr_bn2d_bwd_finalize_totals_f32")
    else:
        with _timed("bn2d_bwd_finalize", 8 * ns * c):
            rc = lib().peclr_bn2d_bwd_finalize_f32(partial.data_ptr(), ns, r, c, int(training), ss.data_ptr(),
                                                   dparams[0].data_ptr(), dparams[1].data_ptr(), coef.data_ptr(),
                                                   _stream())
        _check(rc, "peclr_bn2d_bwd_finalize_f32")
    with _timed("bn2d_bwd_apply_avgpool", 3 * e * r * c + r * c // 8 + 4 * n * c):
        rc = lib().peclr_bn2d_bwd_apply_avgpool(dp, xp, mp, io, n, h * w, c, save[0].data_ptr(), save[1].data_ptr(),
                                                ss.data_ptr(), coef.data_ptr(), dx.data_ptr(), dres.data_ptr(),
                                                _absmax_ptr(absmax, x), _stream())
    _check(rc, "peclr_bn2d_bwd_apply_avgpool")
    return dx, dparams[0], dparams[1], dres


def bn2d_pool_fwd(x, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, sync_group=None, sync_shift=None,
                  pre=None, absmax=None):
    """Stem: y = maxpool3x3/2(relu(bn(x))) in one pass; returns (y, tap codes, save, scale_shift).
    pre: (partial, n_split, shift) -- the statistics the stem convolution summed in its epilogue (no pass over x then)."""
    n, c, h, w = x.shape
    io, e = _IO[x.dtype]
    save, ss = _bn2d_scale_shift(x, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, sync_group, sync_shift, pre)
    ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, c, ph, pw), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    x_at_max = torch.empty_like(y)
    code = torch.empty((n, ph, pw, c), device=x.device, dtype=torch.uint8)
    with _timed("bn2d_pool_apply", e * n * c * (h * w + 2 * ph * pw) + n * c * ph * pw):
        rc = lib().peclr_bn2d_pool_apply(_nhwc_ptr(x, "bn2d x"), io, n, h, w, c, ss.data_ptr(), y.data_ptr(),
                                         x_at_max.data_ptr(), code.data_ptr(), _absmax_ptr(absmax, x), _stream())
    _check(rc, "peclr_bn2d_pool_apply")
    return y, x_at_max, code, save, ss


def bn2d_pool_bwd(dy, x, x_at_max, code, save, ss, training, sync_group=None):
    n, c, h, w = x.shape
    r = n * h * w
    ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    io, e = _IO[x.dtype]
    dev = x.device
    ns = lib().peclr_bn2d_pool_n_split(n, h, w, c, io)
    if ns < 1:
        raise PeclrHipError(f"fused stem BN+ReLU+max-pool: unsupported shape {tuple(x.shape)}")
    partial = torch.empty((2 * ns, c), device=dev, dtype=torch.float32)
    dparams = torch.empty((2, c), device=dev, dtype=torch.float32)
    coef = torch.empty((2, c), device=dev, dtype=torch.float32)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dyp, xp = _nhwc_ptr(dy, "pooled dy", x.dtype), _nhwc_ptr(x, "bn2d x")
    pooled = n * c * ph * pw
    with _timed("bn2d_pool_bwd_reduce", 2 * e * pooled):
        rc = lib().peclr_bn2d_pool_bwd_reduce(dyp, _nhwc_ptr(x_at_max, "x at the maximum", x.dtype), io, n, h, w, c,
                                              save[0].data_ptr(), save[1].data_ptr(), ss.data_ptr(), partial.data_ptr(),
                                              ns, _stream())
    _check(rc, "peclr_bn2d_pool_bwd_reduce")
    if training and sync_group is not None:
        local, total = _sync_totals(partial, ns, c, r, sync_group)
        with _timed("bn2d_bwd_finalize", 32 * c):
            rc = lib().peclr_bn2d_bwd_finalize_totals_f32(local.data_ptr(), total.data_ptr(), c, 1, ss.data_ptr(),
                                                          dparams[0].data_ptr(), dparams[1].data_ptr(), coef.data_ptr(),
                                                          _stream())
        _check(rc, "peclr_bn2d_bwd_finalize_totals_f32")
    else:
        with _timed("bn2d_bwd_finalize", 8 * ns * c):
            rc = lib().peclr_bn2d_bwd_finalize_f32(partial.data_ptr(), ns, r, c, int(training), ss.data_ptr(),
                                                   dparams[0].data_ptr(), dparams[1].data_ptr(), coef.data_ptr(),
                                                   _stream())
        _check(rc, "peclr_bn2d_bwd_finalize_f32")
    with _timed("bn2d_pool_bwd_apply", (e + 1) * pooled + 2 * e * r * c):
        rc = lib().peclr_bn2d_pool_bwd_apply(dyp, xp, code.data_ptr(), io, n, h, w, c, save[0].data_ptr(),
                                             save[1].data_ptr(), ss.data_ptr(), coef.data_ptr(), dx.data_ptr(), _stream())
    _check(rc, "peclr_bn2d_pool_bwd_apply")
    return dx, dparams[0], dparams[1]


# ------------------------------------------------------------------ two-view augmentation (pixel side)
AUG_PARAM_DOUBLES = 16


def augment_views(images: torch.Tensor, params: torch.Tensor, out_hw, mean, std, channels_last: bool = True):
    """images [B,H,W,3] uint8 (HIP), params [V,B,16] float64 (HIP) -> float32 [V*B,3,out_h,out_w]
    (channels_last storage if asked).  Two launches: rotate+crop window, then resize+colour+normalise."""
    if not images.is_cuda or images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise PeclrHipError(f"augment: images must be a [B,H,W,3] uint8 HIP tensor, got {images.dtype} "
                            f"{tuple(images.shape)} on {images.device} (peclr_amd has no CPU path)")
    if not images.is_contiguous():
        raise PeclrHipError("augment: images must be contiguous")
    b, h, w, _ = images.shape
    if (params.dtype != torch.float64 or params.dim() != 3 or params.shape[1] != b
            or params.shape[2] != AUG_PARAM_DOUBLES or not params.is_cuda or not params.is_contiguous()):
        raise PeclrHipError(f"augment: params must be a contiguous [V,{b},{AUG_PARAM_DOUBLES}] float64 HIP tensor")
    v = params.shape[0]
    oh, ow = out_hw
    crops = torch.empty((v, b, h, w, 3), device=images.device, dtype=torch.uint8)
    out = torch.empty((v * b, 3, oh, ow), device=images.device, dtype=torch.float32,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    mean_arr, std_arr = (c_float * 3)(*mean), (c_float * 3)(*std)
    with _timed("augment_warp_crop", 2 * v * b * h * w * 3):
        rc = lib().peclr_augment_warp_crop_u8(images.data_ptr(), b, h, w, v, params.data_ptr(), crops.data_ptr(), _stream())
    _check(rc, "peclr_augment_warp_crop_u8")
    with _timed("augment_resize_color_norm", v * b * (h * w * 3 + oh * ow * 12)):
        rc = lib().peclr_augment_resize_color_norm(crops.data_ptr(), b, h, w, v, params.data_ptr(), oh, ow,
                                                   ctypes.cast(mean_arr, c_void_p), ctypes.cast(std_arr, c_void_p),
                                                   int(channels_last), out.data_ptr(), _stream())
    _check(rc, "peclr_augment_resize_color_norm")
    return out, crops
