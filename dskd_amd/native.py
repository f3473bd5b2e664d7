"""ctypes binding of the gfx950 hot-path library (``include/dskd_hip.h``).

PyTorch is used only as the owner of device memory and of the HIP stream: every
call hands raw device pointers and sizes across the C-ABI.  There is NO CPU
fallback here -- if ``libdskd_hip.so`` is missing or a tensor is not on the GPU the
call raises.  (Tests and ``bench.py``'s CPU baseline may *inject* a checker
implementation with :func:`install_cpu_checker`; the package itself never imports
``oracle/``.)
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

# DSKD_HIP_LIB: another build of the same library (A/B runs of kernel variants built with different -D flags)
_LIB_PATH = os.environ.get("DSKD_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_C", "libdskd_hip.so")
_lib: Optional[C.CDLL] = None

DTYPE_F32, DTYPE_BF16 = 0, 1
ERR_INVALID_COST, ERR_INFEASIBLE = -3, -4

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_SIGNATURES = {
    "dskd_abi_version": (C.c_int, []),
    "dskd_last_error": (C.c_char_p, []),
    "dskd_device_count": (C.c_int, []),
    "dskd_zero_fill": (C.c_int, [_vp, _i64, _vp]),
    "dskd_msda_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 8 + [_vp]),
    "dskd_msda_fwd_fused": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 8 + [_vp]),
    "dskd_msda_bwd": (C.c_int, [_vp] * 9 + [C.c_int] * 8 + [_vp]),
    "dskd_msda_bwd_workspace": (C.c_int64, [C.c_int] * 6),
    "dskd_msda_bwd_ws": (C.c_int, [_vp] * 9 + [C.c_int] * 8 + [_vp, _i64, _vp]),
    "dskd_msda_prep_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_msda_prep_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_msda_grad_ref": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_add_ln_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _f32, _f32,
                                   C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_add_ln_bwd": (C.c_int, [_vp] * 9 + [C.c_int, _i64, C.c_int, _f32, C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_add_ln_bwd2": (C.c_int, [_vp] * 10 + [C.c_int, _i64, C.c_int, _f32, C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_add_pos": (C.c_int, [_vp, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp]),
    "dskd_bias_act": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_bias_relu_maxpool": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 5 + [_vp]),
    "dskd_dropout_fwd": (C.c_int, [_vp, _i64, _f32, C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_relu_dropout_bwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, _f32, C.c_int, _vp]),
    "dskd_colsum": (C.c_int, [_vp, _vp, C.c_int, _i64, C.c_int, C.c_int, _vp]),
    "dskd_ffn_packed_bytes": (_i64, [C.c_int, C.c_int]),
    "dskd_ffn_pack": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_ffn_fwd": (C.c_int, [_vp] * 6 + [_i64, C.c_int, C.c_int, _f32, C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_ffn_bwd": (C.c_int, [_vp] * 7 + [C.c_int, _i64, C.c_int, C.c_int, _f32, C.c_int, _vp]),
    "dskd_lin256_packed_bytes": (_i64, [C.c_int]),
    "dskd_lin256_pack": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_lin256_pack_many": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "dskd_lin256_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_gemm_nt": (C.c_int, [_vp] * 5 + [_i64] + [C.c_int] * 9 + [_vp]),
    "dskd_conv3x3": (C.c_int, [_vp] * 5 + [C.c_int] * 8 + [_vp]),
    "dskd_sum_clear": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "dskd_colsum_short": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, _vp]),
    "dskd_gemm_nt_dx": (C.c_int, [_vp] * 5 + [_i64] + [C.c_int] * 3 + [_vp]),
    "dskd_conv3x3_dx": (C.c_int, [_vp] * 4 + [C.c_int] * 6 + [_vp]),
    "dskd_gemm_nt_scratch_bytes": (_i64, []),
    "dskd_gemm_nt_ws": (C.c_int, [_vp] * 6 + [_i64] + [C.c_int] * 9 + [_vp, _i64, _vp]),
    "dskd_conv3x3_ws": (C.c_int, [_vp] * 6 + [C.c_int] * 8 + [_vp, _i64, _vp]),
    "dskd_gemm_nt_tune": (C.c_int, [C.c_int, C.c_int]),
    "dskd_gemm_tn": (C.c_int, [_vp] * 3 + [_i64] + [C.c_int] * 5 + [_vp]),
    "dskd_cvt_clear": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "dskd_gemm_tn_scratch_bytes": (_i64, [_i64, C.c_int, C.c_int]),
    "dskd_gemm_tn_bf16": (C.c_int, [_vp] * 4 + [_i64, _i64] + [C.c_int] * 5 + [_vp]),
    "dskd_winattn_fwd": (C.c_int, [_vp] * 4 + [C.c_int] * 5 + [_f32, C.c_int, _vp]),
    "dskd_winattn_bwd": (C.c_int, [_vp] * 6 + [C.c_int] * 5 + [_f32, C.c_int, _vp]),
    "dskd_gemm_tn_bias_bf16": (C.c_int, [_vp] * 5 + [_i64, _i64] + [C.c_int] * 5 + [_vp]),
    "dskd_conv3x3_wgrad_scratch_bytes": (C.c_int64, [C.c_int] * 6),
    "dskd_conv3x3_wgrad": (C.c_int, [_vp] * 4 + [C.c_int64] + [C.c_int] * 7 + [_vp]),
    "dskd_conv3x3_wgrad_bias": (C.c_int, [_vp] * 5 + [C.c_int64] + [C.c_int] * 7 + [_vp]),
    "dskd_weight_t_many": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_attn_fwd": (C.c_int, [_vp] * 5 + [C.c_int] * 4 + [_vp, _f32, _f32, C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_attn_bwd": (C.c_int, [_vp] * 10 + [C.c_int] * 4 + [_vp, _f32, _f32, C.c_uint64, C.c_uint64, _vp, C.c_int, _vp]),
    "dskd_clip_adamw_chunk": (C.c_int, []),
    "dskd_clip_adamw": (C.c_int, [_vp] * 5 + [C.c_int, C.c_int, _vp, _vp, C.c_int, _f32, _f32, _f32, _i64, _f32, _vp]),
    "dskd_cast_scale_chunk": (C.c_int, []),
    "dskd_cast_scale_many": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dskd_gn_workspace": (_i64, [C.c_int, _i64]),
    "dskd_nhwc_to_nchw_f32": (C.c_int, [_vp, _vp, C.c_int, _i64, C.c_int, _i64, C.c_int, _vp]),
    "dskd_gn_fwd": (C.c_int, [_vp] * 6 + [C.c_int, _i64, C.c_int, C.c_int, _i64, _i64, _f32, C.c_int, C.c_int, _vp]),
    "dskd_gn_bwd": (C.c_int, [_vp] * 8 + [C.c_int, C.c_int, _i64, C.c_int, C.c_int, _i64, _i64, _i64, C.c_int, C.c_int, _vp]),
    "dskd_lsap_host": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "dskd_lsap_batched": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "dskd_lsap_tune": (C.c_int, [C.c_int]),
    "dskd_match_cost": (C.c_int, [_vp] * 7 + [C.c_int] * 3 + [_f32] * 3 + [_vp]),
    "dskd_dense_loss_fwd": (C.c_int, [_vp] * 13 + [C.c_int] * 4 + [_f32] * 4 + [_vp]),
    "dskd_dense_loss_bwd": (C.c_int, [_vp] * 8 + [C.c_int] * 4 + [_f32] * 4 + [_vp]),
    "dskd_proto_corr_workspace": (_i64, [C.c_int, C.c_int]),
    "dskd_proto_corr_fwd": (C.c_int, [_vp] * 6 + [C.c_int] * 5 + [_f32] + [_vp] * 4),
    "dskd_fgkd_workspace": (_i64, [C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    "dskd_fgkd_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, _f32,
                                 _vp, _vp, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class NativeError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


# every DSKD_* environment switch something in this repository reads (package, library, bench.py, tools/): a variable that
# is set but not listed here does nothing -- load() says so, instead of an A/B run quietly timing the same code twice
KNOWN_ENV = frozenset((
    "DSKD_HIP_LIB", "DSKD_CONV_LIB", "DSKD_CONV3_WGRAD_LIB", "DSKD_SDPA_ATTN", "DSKD_NO_GRAPHS", "DSKD_FORCE_GRAPHS", "DSKD_EAGER_HEAD",
    "DSKD_EAGER_LOSSES", "DSKD_GRAPH_TRACE", "DSKD_MSDA_MM", "DSKD_MSDA_PULL_LEVELS", "DSKD_GRADSYNC_NOCOMM",
    "DSKD_GRADSYNC_BUCKET_MB", "DSKD_WRAP_DDP", "DSKD_BENCH_REHEARSE", "DSKD_BENCH_STEPTIMES", "DSKD_BENCH_DDP1",
    "DSKD_BENCH_WRAP_DDP"))


def unknown_env():
    """Names of set ``DSKD_*`` variables that nothing reads."""
    return sorted(k for k in os.environ if k.startswith("DSKD_") and k not in KNOWN_ENV)


def load() -> C.CDLL:
    """Load the HIP library or raise: the product path has no fallback."""
    global _lib
    if _lib is None:
        stray = unknown_env()
        if stray:
            import warnings
            warnings.warn(f"environment variables {stray} are set but no DSKD switch of that name exists (known: "
                          f"{sorted(KNOWN_ENV)}): they have no effect", RuntimeWarning, stacklevel=2)
        if not os.path.exists(_LIB_PATH):
            raise NativeError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (dskd_amd/csrc/build.sh); there is no CPU fallback")
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.dskd_abi_version() != 2:
            raise NativeError("libdskd_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def zeros(shape, dtype, device) -> torch.Tensor:
    """``torch.zeros`` for the accumulators the kernels add into, filled by a KERNEL: ``torch.zeros`` issues
    hipMemsetAsync, and a memset node captured into a hipGraph replays with a garbage fill value on this ROCm runtime
    (csrc/common.h) -- a replayed backward would then accumulate into garbage."""
    t = torch.empty(shape, dtype=dtype, device=device)
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return t
    if nbytes % 16 or t.data_ptr() % 16 or not t.is_cuda:
        return t.zero_()
    _check(load().dskd_zero_fill(t.data_ptr(), nbytes, _stream(t)), "dskd_zero_fill")
    return t


_acc_cache = {}


def _persistent_acc(shape, device) -> torch.Tensor:
    """A zeroed f32 accumulator of ``shape`` that STAYS zeroed between uses: the kernels add their column sums into it and
    :func:`sum_clear` hands the result over and clears it again -- no zero-fill launch per call.  One per (shape, device,
    STREAM): two streams never add into the same words.  Use it under :func:`_acc_guard`."""
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    key = (tuple(shape), device, torch.cuda.current_stream(device).cuda_stream)
    t = _acc_cache.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # no warm-up ran on the capture stream: an accumulator of the graph's own pool, zeroed by a captured fill
            # kernel on every replay (torch.zeros would be a memset node: see zeros()); not cached -- it lives with the graph
            return zeros(shape, torch.float32, device)
        t = _acc_cache[key] = torch.zeros(shape, dtype=torch.float32, device=device)
    return t


class _acc_guard:
    """``with _acc_guard(acc):`` around the accumulating launch and its hand-over: if anything between them raises (a failed
    launch, an allocation failure in the hand-over), the accumulator is zeroed again before the error travels on -- a
    training loop that catches the error and skips the batch must not find the residue in every later gradient."""

    def __init__(self, acc):
        self.acc = acc

    def __enter__(self):
        return self.acc

    def __exit__(self, etype, evalue, tb):
        if etype is not None and self.acc is not None:
            try:
                self.acc.zero_()
            except Exception:          # the device itself is gone: nothing left to protect
                pass
        return False


def sum_clear(acc: torch.Tensor, planes: int, copies: int, Cc: int, out_dtype=torch.float32) -> torch.Tensor:
    """``acc.view(planes, copies, Cc).sum(1)`` in ``out_dtype`` (f32 | bf16), leaving ``acc`` zeroed (dskd_sum_clear)."""
    out = torch.empty((planes, Cc), dtype=out_dtype, device=acc.device)
    dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[out_dtype]
    _check(load().dskd_sum_clear(acc.data_ptr(), planes, copies, Cc, out.data_ptr(), dt, _stream(acc)), "dskd_sum_clear")
    return out


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dskd_last_error().decode()
        if rc in (ERR_INVALID_COST, ERR_INFEASIBLE):
            raise ValueError(msg)  # scipy raises ValueError for these
        raise NativeError(f"{what} failed ({rc}): {msg}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise NativeError("dskd_amd hot-path ops need GPU tensors (no CPU fallback in the product path)")


def _host_i64(vals: Sequence[int]):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


def _host_i32(vals: Sequence[int]):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def _host_f32(vals: Sequence[float]):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


# --------------------------------------------------------------------------- checker hook
_cpu_checker = None


def install_cpu_checker(impl) -> None:
    """TEST / CPU-BASELINE ONLY.  ``impl`` provides the same ops for CPU tensors
    (the oracle); it is consulted only when an op receives CPU tensors.  The
    package never installs one itself."""
    global _cpu_checker
    _cpu_checker = impl


def cpu_checker():
    return _cpu_checker


def _dispatch_cpu(name: str, t: torch.Tensor):
    if t.is_cuda:
        return None
    if _cpu_checker is None:
        raise NativeError(f"{name}: CPU tensor given; the HIP path needs GPU tensors (no CPU fallback)")
    return getattr(_cpu_checker, name)


# --------------------------------------------------------------------------- live kernel timing
class _KernelTiming:
    """HIP-event brackets around individual kernel launches (bench.py roofline): events are
    recorded on the stream the kernel is launched on (torch's current stream)."""
    enabled = False
    records = []


def timing_enable(on: bool = True) -> None:
    _KernelTiming.enabled = on
    _KernelTiming.records = []


def timing_collect():
    """-> {tag: (launches, total_ms)}; call after a device synchronize."""
    out = {}
    for tag, e0, e1 in _KernelTiming.records:
        n, t = out.get(tag, (0, 0.0))
        out[tag] = (n + 1, t + e0.elapsed_time(e1))
    _KernelTiming.records = []
    return out


class _timed:
    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if _KernelTiming.enabled:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _KernelTiming.enabled:
            self.e1.record()
            _KernelTiming.records.append((self.tag, self.e0, self.e1))
        return False


# --------------------------------------------------------------------------- MSDA
def _geom(spatial_shapes: Sequence[Tuple[int, int]]):
    flat, starts, acc = [], [], 0
    for h, w in spatial_shapes:
        flat += [int(h), int(w)]
        starts.append(acc)
        acc += int(h) * int(w)
    return _host_i64(flat), _host_i64(starts), acc


def msda_forward_raw(value, shapes, loc, attn):
    _need_gpu(value, loc, attn)
    B, Nv, heads, ch = value.shape
    _, Nq, _, L, P, _ = loc.shape
    ss, ls, tot = _geom(shapes)
    assert tot == Nv, f"spatial shapes cover {tot} rows, value has {Nv}"
    dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[value.dtype]
    value, loc, attn = value.contiguous(), loc.contiguous().float(), attn.contiguous().float()
    out = torch.empty((B, Nq, heads * ch), dtype=value.dtype, device=value.device)
    with _timed("msda_fwd_enc" if Nq == Nv else "msda_fwd_dec"):
        rc = load().dskd_msda_fwd(value.data_ptr(), ss, ls, loc.data_ptr(), attn.data_ptr(), out.data_ptr(),
                                  B, Nv, Nq, heads, ch, L, P, dt, _stream(value))
    _check(rc, "dskd_msda_fwd")
    return out


def msda_backward_raw(value, shapes, loc, attn, grad_out, use_workspace=True):
    """(grad_value f32, grad_loc, grad_attn) of the sampling step.  The encoder shape (queries == pixels, 4 levels x 4
    points) goes through ``dskd_msda_bwd_ws`` (tiled pull for level 0, fused levels-2+3 launch, statistics by-product;
    grad_value written, not accumulated -> no zero fill); ``use_workspace=False`` (tests) and every other shape through the
    plain entry point ``dskd_msda_bwd`` (windowed LDS accumulation for the encoder shape, global atomics otherwise)."""
    _need_gpu(value, loc, attn, grad_out)
    B, Nv, heads, ch = value.shape
    _, Nq, _, L, P, _ = loc.shape
    ss, ls, _ = _geom(shapes)
    dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[value.dtype]
    value, loc, attn = value.contiguous(), loc.contiguous().float(), attn.contiguous().float()
    grad_out = grad_out.contiguous().to(value.dtype)
    gl = torch.empty_like(loc)
    ga = torch.empty_like(attn)
    if Nq == Nv and L == 4 and P == 4 and use_workspace:
        ws = _msda_bwd_workspace(value.device, B, Nv, Nq, heads, L, P)
        gv = torch.empty((B, Nv, heads, ch), dtype=torch.float32, device=value.device)
        with _timed("msda_bwd_enc"):
            rc = load().dskd_msda_bwd_ws(value.data_ptr(), ss, ls, loc.data_ptr(), attn.data_ptr(), grad_out.data_ptr(),
                                         gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, Nv, Nq, heads, ch, L, P, dt,
                                         ws.data_ptr(), ws.numel(), _stream(value))
        _check(rc, "dskd_msda_bwd_ws")
        return gv, gl, ga
    gv = zeros((B, Nv, heads, ch), torch.float32, value.device)
    with _timed("msda_bwd_enc" if Nq == Nv else "msda_bwd_dec"):
        rc = load().dskd_msda_bwd(value.data_ptr(), ss, ls, loc.data_ptr(), attn.data_ptr(), grad_out.data_ptr(),
                                  gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, Nv, Nq, heads, ch, L, P, dt,
                                  _stream(value))
    _check(rc, "dskd_msda_bwd")
    return gv, gl, ga


_msda_ws_cache = {}


def _msda_bwd_workspace(device, B, Nv, Nq, heads, L, P) -> torch.Tensor:
    """Workspace of ``dskd_msda_bwd_ws`` (stray-sample list; header zeroed once, the library leaves it zeroed),
    one per (device, stream): launches of one stream run one after the other, two streams never share a stray list."""
    need = int(load().dskd_msda_bwd_workspace(B, Nv, Nq, heads, L, P))
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _msda_ws_cache.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            return zeros((need + 15) // 16 * 16, torch.uint8, device)     # the graph's own (see _persistent_acc)
        ws = _msda_ws_cache[key] = torch.zeros(need, dtype=torch.uint8, device=device)
    return ws


def graph_pins(device):
    """What a hipGraph captured on ``device`` must keep alive of this module's caches: the MSDA backward workspace
    (re-allocated when a larger shape arrives -- the captured launches keep writing their stray list into the one they
    were captured on) and the dropout epoch word."""
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    return [t for t in (_drop_epochs.get(device),) if t is not None] + [t for k, t in _msda_ws_cache.items() if k[0] == device] + \
        [t for k, t in _tn_acc.items() if k[2] == device] + [t for k, t in _acc_cache.items() if k[1] == device] + \
        [t for (d, _), t in _tn_scratch.items() if d == device] + \
        [t for (d, _), t in _gemm_ws_cache.items() if d == device.index] + \
        [t for pre in set(_prepacked.values()) for t in pre.pins() if t.device == device]


class _MSDAFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, loc, attn, shapes):
        ctx.shapes = shapes
        ctx.save_for_backward(value, loc, attn)
        return msda_forward_raw(value, shapes, loc, attn)

    @staticmethod
    def backward(ctx, grad_out):
        value, loc, attn = ctx.saved_tensors
        gv, gl, ga = msda_backward_raw(value, ctx.shapes, loc, attn, grad_out)
        return gv.to(value.dtype), gl.to(loc.dtype), ga.to(attn.dtype), None


def ms_deform_attn(value: torch.Tensor, spatial_shapes: Sequence[Tuple[int, int]],
                   sampling_locations: torch.Tensor, attention_weights: torch.Tensor) -> torch.Tensor:
    """``MultiScaleDeformableAttnFunction.apply`` of ext-mmcv (call sites
    /root/reference/mmdet/models/utils/transformer.py:985-995, :1032-1043).
    value [B,Nv,heads,ch], sampling_locations [B,Nq,heads,L,P,2],
    attention_weights [B,Nq,heads,L,P] -> [B,Nq,heads*ch]; differentiable."""
    f = _dispatch_cpu("ms_deform_attn", value)
    if f is not None:
        return f(value, spatial_shapes, sampling_locations, attention_weights)
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    return _MSDAFunction.apply(value, sampling_locations, attention_weights, shapes)


def ms_deform_attn_fused(value: torch.Tensor, spatial_shapes, both: torch.Tensor, reference_points: torch.Tensor,
                         levels: int, points: int) -> torch.Tensor:
    """No-gradient forward of the whole sampling step of the module: softmax + sampling locations
    (:func:`msda_prepare`) folded into the sampling kernel, so loc / attn are never materialised.
    value [B,Nv,heads,ch], both [B,Nq,heads*16*3] (same dtype), reference_points [B,Nq,levels,2].
    Bit-identical to ``ms_deform_attn(value, shapes, *msda_prepare(...))``.  Raw op: no autograd."""
    _need_gpu(value, both, reference_points)
    B, Nv, heads, ch = value.shape
    Nq = both.shape[1]
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    ss, ls, _ = _geom(shapes)
    dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[value.dtype]
    value = value.contiguous()
    both = both.contiguous().to(value.dtype)
    ref = reference_points.detach().contiguous().float()
    out = torch.empty((B, Nq, heads * ch), dtype=value.dtype, device=value.device)
    with _timed("msda_fwd_enc_fused" if Nq == Nv else "msda_fwd_dec_fused"):
        rc = load().dskd_msda_fwd_fused(value.data_ptr(), ss, ls, both.data_ptr(), ref.data_ptr(), out.data_ptr(), B, Nv, Nq,
                                        heads, ch, levels, points, dt, _stream(value))
    _check(rc, "dskd_msda_fwd_fused")
    return out


class _MSDAPrepFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, both, ref, shapes, heads, levels, points):
        dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[both.dtype]
        both = both.contiguous()
        ref_f = ref.detach().contiguous().float()
        lead = both.shape[:-1]
        nq = both.numel() // both.shape[-1]
        loc = torch.empty(lead + (heads, levels, points, 2), dtype=torch.float32, device=both.device)
        attn = torch.empty(lead + (heads, levels, points), dtype=torch.float32, device=both.device)
        ss, _, _ = _geom(shapes)
        rc = load().dskd_msda_prep_fwd(both.data_ptr(), ref_f.data_ptr(), ss, loc.data_ptr(), attn.data_ptr(), nq, heads,
                                       levels, points, dt, _stream(both))
        _check(rc, "dskd_msda_prep_fwd")
        ctx.save_for_backward(attn)
        ctx.meta = (shapes, heads, levels, points, dt, both.dtype, both.shape, nq, ref.shape, ref.dtype)
        return loc, attn

    @staticmethod
    def backward(ctx, grad_loc, grad_attn):
        (attn,) = ctx.saved_tensors
        shapes, heads, levels, points, dt, dtype, shape, nq, ref_shape, ref_dtype = ctx.meta
        ss, _, _ = _geom(shapes)
        grad_both = torch.empty(shape, dtype=dtype, device=attn.device)
        gl, ga = grad_loc.contiguous().float(), grad_attn.contiguous().float()
        rc = load().dskd_msda_prep_bwd(gl.data_ptr(), ga.data_ptr(), attn.data_ptr(), ss, grad_both.data_ptr(), nq, heads,
                                       levels, points, dt, _stream(attn))
        _check(rc, "dskd_msda_prep_bwd")
        grad_ref = None
        if ctx.needs_input_grad[1]:
            # loc = ref[..., None, :, None, :] + off / (W, H): d(ref) = sum of d(loc) over heads and points
            gr = torch.empty((nq, levels, 2), dtype=torch.float32, device=gl.device)
            _check(load().dskd_msda_grad_ref(gl.data_ptr(), gr.data_ptr(), nq, heads, levels, points, _stream(gl)),
                   "dskd_msda_grad_ref")
            grad_ref = gr.view(ref_shape).to(ref_dtype)
        return grad_both, grad_ref, None, None, None, None


def msda_prepare(both: torch.Tensor, reference_points: torch.Tensor, spatial_shapes, heads: int, levels: int,
                 points: int):
    """softmax of the attention logits + sampling locations from the projection output
    (``both[..., :heads*L*P*2]`` offsets, the rest logits) and reference points [.., levels, 2]
    (differentiable: the decoder's come from a trainable Linear); one launch each way (plus one
    small reduction for d(reference points)) instead of the module's elementwise chain
    (ext-mmcv MultiScaleDeformableAttention.forward).  Returns (loc, attn) in f32."""
    f = _dispatch_cpu("msda_prepare", both)
    if f is not None:
        return f(both, reference_points, spatial_shapes, heads, levels, points)
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    return _MSDAPrepFunction.apply(both, reference_points, shapes, heads, levels, points)


# --------------------------------------------------------------------------- add + dropout + LayerNorm
_drop_calls = 0


_drop_epochs = {}


def _next_drop_key():
    """(seed, offset) of the next dropout mask: torch's seed (so ``torch.manual_seed`` governs
    it) and a per-process call counter; no device work, no synchronisation."""
    global _drop_calls
    _drop_calls += 1
    return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, _drop_calls


def dropout_epoch(device) -> torch.Tensor:
    """The device word every dropout kernel adds to its ``offset`` (one int64 per device).  (seed, offset) are
    launch arguments: captured into a hipGraph they are frozen, and every replay would redraw the SAME masks.  The
    kernels therefore read this word as well; whoever replays a graph that contains dropout advances it between
    replays (:func:`advance_dropout_epoch`) -- never between a forward and its backward.  Call-site offsets advance
    by one per launch, the epoch by 2^32 per step, so (offset + epoch) never repeats."""
    device = torch.device(device)
    t = _drop_epochs.get(device)
    if t is None:
        t = _drop_epochs[device] = torch.zeros((), dtype=torch.int64, device=device)
    return t


def advance_dropout_epoch(device) -> None:
    """New dropout masks for the next replay of any captured graph (one tiny launch; not to be captured)."""
    if torch.cuda.is_current_stream_capturing():
        raise NativeError("advance_dropout_epoch() inside a hipGraph capture: the increment would be replayed, "
                          "which is fine, but the captured forward / backward pair must see ONE value")
    dropout_epoch(device).add_(1 << 32)


def _pos_f32(pos):
    """The f32 table the kernels read for a positional input.  A caller that keeps the table in the compute dtype as the
    AUTOGRAD input (so that its gradient can be handed back without a cast or a summing pass) attaches the f32 values as
    ``pos._dskd_f32``; otherwise one cast per call."""
    if pos is None:
        return None
    f = getattr(pos, "_dskd_f32", None)
    if f is not None and f.shape == pos.shape and f.dtype == torch.float32 and f.is_contiguous():
        return f
    return pos.detach().float().contiguous()


class _AddLNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, res, gamma, beta, pos, eps, p, want_q, fork=False):
        dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[h.dtype]
        h, res = h.contiguous(), res.contiguous()
        rows, D = h.numel() // h.shape[-1], h.shape[-1]
        train = any(ctx.needs_input_grad[:5])
        y = torch.empty_like(h)
        q = torch.empty_like(h) if want_q else None
        z = torch.empty_like(h) if train else None
        stats = torch.empty((rows, 2), dtype=torch.float32, device=h.device) if train else None
        gamma_f, beta_f = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        pos_f = _pos_f32(pos)
        seed, offset = _next_drop_key() if p > 0 else (0, 0)
        rc = load().dskd_add_ln_fwd(
            h.data_ptr(), res.data_ptr(), None if pos_f is None else pos_f.data_ptr(),
            0 if pos_f is None else pos_f.numel() // D, gamma_f.data_ptr(), beta_f.data_ptr(), y.data_ptr(),
            None if q is None else q.data_ptr(), None if z is None else z.data_ptr(),
            None if stats is None else stats.data_ptr(), rows, D, eps, p, seed, offset,
            dropout_epoch(h.device).data_ptr() if p > 0 else None, dt, _stream(h))
        _check(rc, "dskd_add_ln_fwd")
        if train:
            ctx.save_for_backward(z, stats, gamma_f)
        ctx.meta = (dt, rows, D, p, seed, offset, None if pos is None else tuple(pos.shape), gamma.dtype, want_q)
        ctx.pos_dtype = None if pos is None else pos.dtype
        # fork: y is handed out TWICE (the second an alias): a caller that feeds y to two consumers gives each its own, and
        # their gradients come back here as two tensors that the backward kernel sums itself (no add launch by autograd)
        return y, (y.detach() if fork else None), q

    @staticmethod
    def backward(ctx, dy, dy2, dq):
        z, stats, gamma_f = ctx.saved_tensors
        dt, rows, D, p, seed, offset, pos_shape, gdtype, want_q = ctx.meta
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            dy = zeros(z.shape, z.dtype, z.device)
        dy = dy.contiguous().to(z.dtype)
        dy2 = dy2.contiguous().to(z.dtype) if dy2 is not None else None
        dq = dq.contiguous().to(z.dtype) if (want_q and dq is not None) else None
        dres = torch.empty_like(z)
        dh = torch.empty_like(z) if p > 0 else None
        copies = _colsum_copies(rows)
        dgb = _persistent_acc((2, copies, D), z.device)
        with _acc_guard(dgb):
            rc = load().dskd_add_ln_bwd2(dy.data_ptr(), None if dy2 is None else dy2.data_ptr(),
                                         None if dq is None else dq.data_ptr(), z.data_ptr(), stats.data_ptr(),
                                         gamma_f.data_ptr(), dres.data_ptr(), None if dh is None else dh.data_ptr(),
                                         dgb[0].data_ptr(), dgb[1].data_ptr(), copies, rows, D, p, seed, offset,
                                         dropout_epoch(z.device).data_ptr() if p > 0 else None, dt, _stream(z))
            _check(rc, "dskd_add_ln_bwd2")
            dgb = sum_clear(dgb, 2, copies, D)
        dpos = None
        if pos_shape is not None and ctx.needs_input_grad[4] and dq is not None:
            # q = y + pos[r % pos_rows]: d(pos) = sum of dq over the repeats (the images of a batch)
            pos_rows = 1
            for d_ in pos_shape[:-1]:
                pos_rows *= d_
            if pos_rows == rows and dq.dtype == ctx.pos_dtype:
                dpos = dq.view(pos_shape)          # nothing to sum: hand the gradient over as it is (a view, no pass)
            else:
                dpos = dq.view(rows // pos_rows, pos_rows, D).sum(0, dtype=ctx.pos_dtype).view(pos_shape)    # (f32 accumulation inside)
        return (dres if dh is None else dh), dres, dgb[0].to(gdtype), dgb[1].to(gdtype), dpos, None, None, None, None


class _AddPosFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos):
        x = x.contiguous()
        pos_f = _pos_f32(pos)
        D = x.shape[-1]
        rows, pos_rows = x.numel() // D, pos_f.numel() // D
        q = torch.empty_like(x)
        _check(load().dskd_add_pos(x.data_ptr(), pos_f.data_ptr(), q.data_ptr(), rows, pos_rows, D, DTYPE_BF16, _stream(x)),
               "dskd_add_pos")
        ctx.meta = (tuple(pos.shape), pos.dtype, rows, pos_rows, D)
        return q

    @staticmethod
    def backward(ctx, dq):
        pos_shape, pos_dtype, rows, pos_rows, D = ctx.meta
        dpos = None
        if ctx.needs_input_grad[1]:
            if pos_rows == rows:
                dpos = (dq if dq.dtype == pos_dtype else dq.to(pos_dtype)).view(pos_shape)
            else:
                dpos = dq.reshape(rows // pos_rows, pos_rows, D).sum(0, dtype=pos_dtype).view(pos_shape)
        return (dq if ctx.needs_input_grad[0] else None), dpos


def add_pos(x: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """``(x + pos).to(x.dtype)`` for a bf16 CUDA token tensor ``x`` [..., D] and a float positional table ``pos`` whose
    rows repeat over the leading dimension (or match it) -- one streaming pass; anything else goes through ATen."""
    D = x.shape[-1]
    if x.is_cuda and x.dtype == torch.bfloat16 and pos.is_floating_point() and pos.shape[-1] == D and D % 8 == 0 \
            and pos.numel() > 0 and x.numel() % pos.numel() == 0 and x.numel() > 0:
        return _AddPosFunction.apply(x, pos)
    return (x + pos).to(x.dtype)


def add_layer_norm(h: torch.Tensor, res: torch.Tensor, norm: torch.nn.LayerNorm, p: float = 0.0,
                   pos: Optional[torch.Tensor] = None, want_q: bool = False, fork: bool = False):
    """``y = norm(res + dropout_p(h))`` and, with ``want_q``, ``q = y + pos`` -- the tail of a
    transformer sub-layer (ext-mmcv BaseTransformerLayer: ``identity + dropout(out)`` then
    'norm', then the next layer's ``query + query_pos``) as ONE launch each way.
    h, res: [..., 256] f32 | bf16 (same dtype); pos: [..., Nv, 256] broadcast over the leading
    (batch) dimension of a batch-first token tensor.  Returns (y, q or None); with ``fork`` (y, y', q or None) where
    y' is y again as a second autograd output: feed y to one consumer and y' to the other (e.g. the FFN and the residual of
    the next LayerNorm) and their two gradients are summed inside the backward launch instead of by an add launch."""
    f = _dispatch_cpu("add_layer_norm", h)
    if f is not None:
        y, q = f(h, res, norm, p, pos, want_q)
        return (y, y, q) if fork else (y, q)
    _need_gpu(h, res)
    if res.dtype != h.dtype:
        res = res.to(h.dtype)
    y, y2, q = _AddLNFunction.apply(h, res, norm.weight, norm.bias, pos, float(norm.eps), float(p), bool(want_q), bool(fork))
    return (y, y2, q) if fork else (y, q)


# --------------------------------------------------------------------------- FFN hidden activation
def dropout_(y: torch.Tensor, p: float) -> torch.Tensor:
    """In-place dropout of a contiguous bf16 tensor, Philox mask not stored (the backward of the
    FFN recovers it from ``y != 0``, see :func:`relu_dropout_bwd`).  Raw op, no autograd."""
    _need_gpu(y)
    if p > 0:
        seed, offset = _next_drop_key()
        rc = load().dskd_dropout_fwd(y.data_ptr(), y.numel(), p, seed, offset, dropout_epoch(y.device).data_ptr(),
                                     DTYPE_BF16, _stream(y))
        _check(rc, "dskd_dropout_fwd")
    return y


def relu_dropout_bwd(g: torch.Tensor, y_dropped: torch.Tensor, p: float, want_colsum: bool = True,
                     colsum_dtype=torch.float32):
    """Backward of ``dropout_p(relu(.))`` given its OUTPUT: ``g * (y_dropped != 0) / (1 - p)`` and
    the column sums of that (the bias gradient of the Linear in front).  [rows, C] bf16."""
    _need_gpu(g, y_dropped)
    g = g.contiguous()
    Cc = g.shape[-1]
    rows = g.numel() // Cc
    out = torch.empty_like(g)
    copies = _colsum_copies(rows)
    colsum = _persistent_acc((copies, Cc), g.device) if want_colsum else None
    with _acc_guard(colsum):
        rc = load().dskd_relu_dropout_bwd(g.data_ptr(), y_dropped.data_ptr(), out.data_ptr(),
                                          None if colsum is None else colsum.data_ptr(), copies, rows, Cc, p, DTYPE_BF16,
                                          _stream(g))
        _check(rc, "dskd_relu_dropout_bwd")
        if colsum is not None:
            colsum = sum_clear(colsum, 1, copies, Cc, colsum_dtype)[0]
    return out, colsum


FFN_FUSED_DIMS = (256, 1024)      # (d_model, hidden) the MFMA kernels of csrc/ffn_mfma.hip are built for
_ffn_flops = 0


def ffn_flops_launched() -> int:
    """FLOPs of the fused FFN launches enqueued from Python so far (two GEMMs of 2 * tokens * d * hidden per launch);
    bench.py's MFMA-utilisation probe reads the difference over one eager step."""
    return _ffn_flops



def ffn_pack(w1: torch.Tensor, w2: torch.Tensor, want_bwd: bool = True):
    """MFMA fragment-order images of the FFN weights (w1 [hidden, d], w2 [d, hidden], bf16, contiguous): the forward
    image and, with ``want_bwd``, the backward (transposed) one.  Valid until the weights change."""
    _need_gpu(w1, w2)
    if w1.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or not (w1.is_contiguous() and w2.is_contiguous()):
        raise NativeError("ffn_pack: contiguous bf16 weights expected")
    hidden, d = w1.shape
    if tuple(w2.shape) != (d, hidden):
        raise NativeError(f"ffn_pack: w2 {tuple(w2.shape)} does not match w1 {tuple(w1.shape)}")
    nbytes = load().dskd_ffn_packed_bytes(d, hidden)
    if nbytes < 0:
        _check(-1, "dskd_ffn_packed_bytes")
    fwd = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w1.device)
    bwd = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w1.device) if want_bwd else None
    rc = load().dskd_ffn_pack(w1.data_ptr(), w2.data_ptr(), fwd.data_ptr(), None if bwd is None else bwd.data_ptr(),
                              d, hidden, DTYPE_BF16, _stream(w1))
    _check(rc, "dskd_ffn_pack")
    return fwd, bwd


def ffn_fwd_raw(x: torch.Tensor, packed_fwd: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor, p: float,
                store_h: bool, hidden: int = 1024):
    """``y = dropout_p(relu(x w1^T + b1)) w2^T + b2`` on [tokens, d] bf16 in one MFMA launch; returns (y, H) with
    H = the dropped hidden activation [tokens, hidden] when ``store_h`` (needed by :func:`ffn_bwd_raw`), else None."""
    _need_gpu(x, packed_fwd)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.dim() != 2:
        raise NativeError("ffn_fwd_raw: contiguous [tokens, d] bf16 input expected")
    if p > 0 and not store_h:
        raise NativeError("ffn_fwd_raw: dropout needs store_h")
    tokens, d = x.shape
    y = torch.empty_like(x)
    h = torch.empty((tokens, hidden), dtype=x.dtype, device=x.device) if store_h else None
    seed, offset = _next_drop_key() if p > 0 else (0, 0)
    rc = load().dskd_ffn_fwd(x.data_ptr(), packed_fwd.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                             None if h is None else h.data_ptr(), y.data_ptr(), tokens, d, hidden, p, seed, offset,
                             dropout_epoch(x.device).data_ptr() if p > 0 else None, DTYPE_BF16, _stream(x))
    _check(rc, "dskd_ffn_fwd")
    global _ffn_flops
    _ffn_flops += 4 * tokens * d * hidden
    return y, h


def ffn_bwd_raw(grad_y: torch.Tensor, h: torch.Tensor, packed_bwd: torch.Tensor, p: float, want_colsum: bool = False,
                add_to_gx: Optional[torch.Tensor] = None, colsum_dtype=torch.float32):
    """(grad_h, grad_x[, column sums of grad_h in f32 = grad of b1]) of :func:`ffn_fwd_raw` given grad_y [tokens, d]
    and the stored H: one MFMA launch.  ``add_to_gx`` [tokens, d] bf16 is added to grad_x in the kernel's epilogue."""
    _need_gpu(grad_y, h, packed_bwd)
    grad_y = grad_y.contiguous()
    tokens, d = grad_y.shape
    gh = torch.empty_like(h)
    gx = torch.empty_like(grad_y)
    copies = _colsum_copies(tokens)
    cs = _persistent_acc((copies, h.shape[1]), h.device) if want_colsum else None
    if add_to_gx is not None and (add_to_gx.dtype != grad_y.dtype or add_to_gx.shape != grad_y.shape or
                                  not add_to_gx.is_contiguous()):
        raise NativeError("ffn_bwd_raw: add_to_gx must be a contiguous bf16 [tokens, d] tensor")
    global _ffn_flops
    with _acc_guard(cs):
        rc = load().dskd_ffn_bwd(grad_y.data_ptr(), h.data_ptr(), packed_bwd.data_ptr(), gh.data_ptr(), gx.data_ptr(),
                                 None if add_to_gx is None else add_to_gx.data_ptr(),
                                 None if cs is None else cs.data_ptr(), copies, tokens, d, h.shape[1], p, DTYPE_BF16,
                                 _stream(grad_y))
        _check(rc, "dskd_ffn_bwd")
        _ffn_flops += 4 * tokens * d * h.shape[1]
        if want_colsum:
            return gh, gx, sum_clear(cs, 1, copies, h.shape[1], colsum_dtype)[0]
    return gh, gx


COLSUM_WIDTHS = (256, 384, 512, 1024, 2048)


def _colsum_copies(rows: int) -> int:
    """Accumulator copies for the column-sum atomics: tall inputs run ~1000-2000 workgroups, all
    adding to the same C addresses; 32 copies cut that contention 32x for one tiny extra sum."""
    return 32 if rows >= 8192 else 1


def colsum(x: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    """Column sums (f32 accumulation; result f32 or bf16) of a contiguous [rows, C] bf16 GPU matrix: the bias gradient of a
    Linear.  Two launches: partial sums into a persistent accumulator, hand-over + clear."""
    _need_gpu(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    copies = _colsum_copies(rows)
    acc = _persistent_acc((copies, Cc), x.device)
    with _acc_guard(acc):
        rc = load().dskd_colsum(x.data_ptr(), acc.data_ptr(), copies, rows, Cc, DTYPE_BF16, _stream(x))
        _check(rc, "dskd_colsum")
        return sum_clear(acc, 1, copies, Cc, out_dtype)[0]


def colsum_short_ok(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous() and 0 < x.shape[0] < 16384
            and x.shape[1] % 8 == 0 and x.data_ptr() % 16 == 0)


def colsum_short(x: torch.Tensor) -> torch.Tensor:
    """bf16 column sums (f32 accumulation) of a SHORT contiguous [rows, C] bf16 GPU matrix in one launch (dskd_colsum_short)."""
    _need_gpu(x)
    out = torch.empty((x.shape[1],), dtype=x.dtype, device=x.device)
    _check(load().dskd_colsum_short(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], DTYPE_BF16, _stream(x)),
           "dskd_colsum_short")
    return out


# --------------------------------------------------------------------------- tall Linear with 256 inputs (MFMA kernel)
LIN256_ENABLED = True      # tests set this to False to get the library GEMMs (the control of the kernel's parity tests)


def lin256_ok(x: torch.Tensor, n_out: int, k_in: int) -> bool:
    """Can csrc/ffn_mfma.hip::lin256_kernel take ``x [tokens, 256] @ W^T`` (tall contiguous bf16 CUDA input, 256 inputs,
    32..512 outputs in steps of 32)?"""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and k_in == 256 and x.shape[1] == 256
            and n_out % 32 == 0 and 32 <= n_out <= 512 and x.shape[0] >= 16384 and x.is_contiguous()
            and x.data_ptr() % 16 == 0 and LIN256_ENABLED)


class Lin256Prepack:
    """Fragment-order images of a fixed set of weights (persistent bf16 buffers whose CONTENTS change once per step: the
    low-precision parameter copies of transformer.lowp_params), refreshed by ONE launch (:meth:`refresh`) instead of one
    ``lin256_pack`` launch in front of every use.  :func:`lin256_pack` returns an image from here when it was refreshed
    after the last write to its source (``stamp == epoch[0]``; the owner bumps ``epoch[0]`` whenever it rewrites the
    sources) -- anything else packs on the spot as before.  The object keeps sources and images alive, so a data pointer
    cannot come back as another tensor while its entry exists."""

    def __init__(self, weights, epoch):
        self.epoch, self.stamp = epoch, -1
        self.sources, self.images, rows = [], {}, []
        for w in weights:
            if not (w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 2 and w.is_contiguous() and w.data_ptr() % 16 == 0):
                continue
            forms = []
            if w.shape[1] == 256 and w.shape[0] % 32 == 0 and 32 <= w.shape[0] <= 512:
                forms.append((False, w.shape[0]))
            if w.shape[0] == 256 and w.shape[1] % 32 == 0 and 32 <= w.shape[1] <= 512:
                forms.append((True, w.shape[1]))
            for transposed, n in forms:
                key = (w.data_ptr(), tuple(w.shape), transposed)
                if key in self.images:
                    continue
                img = torch.empty(int(load().dskd_lin256_packed_bytes(n)) // 2, dtype=torch.bfloat16, device=w.device)
                self.images[key] = img
                rows.append([w.data_ptr(), img.data_ptr(), n, 1 if transposed else 0])
            if forms:
                self.sources.append(w)
        self.n = len(rows)
        self.table = torch.tensor(rows, dtype=torch.int64, device=weights[0].device) if rows else None

    def refresh(self):
        if self.n:
            _check(load().dskd_lin256_pack_many(self.table.data_ptr(), self.n, DTYPE_BF16, _stream(self.table)),
                   "dskd_lin256_pack_many")
        self.stamp = self.epoch[0]
        for key in self.images:
            _prepacked[key] = self

    def drop(self):
        for key in self.images:
            if _prepacked.get(key) is self:
                del _prepacked[key]

    def pins(self):
        return ([self.table] if self.table is not None else []) + list(self.images.values())


_prepacked = {}      # (data_ptr, shape, transposed) -> the Lin256Prepack that holds this weight's image


def lin256_pack(w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """Fragment-order image of a bf16 weight: ``w`` [N, 256] (nn.Linear layout), or with ``transposed`` ``w`` [256, N]
    whose TRANSPOSE is the layer (the input-gradient GEMM ``g @ w`` of a [256, 256] Linear)."""
    _need_gpu(w)
    if w.dtype != torch.bfloat16 or not w.is_contiguous() or w.dim() != 2:
        raise NativeError("lin256_pack: contiguous 2-D bf16 weight expected")
    pre = _prepacked.get((w.data_ptr(), tuple(w.shape), bool(transposed)))
    if pre is not None and pre.stamp == pre.epoch[0]:
        return pre.images[(w.data_ptr(), tuple(w.shape), bool(transposed))]
    n, k = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    nbytes = load().dskd_lin256_packed_bytes(n)
    if nbytes < 0 or k != 256:
        raise NativeError(f"lin256_pack: unsupported shape {tuple(w.shape)} (transposed={transposed})")
    packed = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    _check(load().dskd_lin256_pack(w.data_ptr(), packed.data_ptr(), n, k, 1 if transposed else 0, DTYPE_BF16, _stream(w)),
           "dskd_lin256_pack")
    return packed


def lin256(x: torch.Tensor, packed: torch.Tensor, n_out: int, bias: Optional[torch.Tensor] = None, relu: bool = False):
    """``act(x @ W^T + bias)`` with ``packed`` = :func:`lin256_pack` of W; x [tokens, 256] bf16 -> [tokens, n_out] bf16."""
    _need_gpu(x, packed)
    tokens = x.shape[0]
    y = torch.empty((tokens, n_out), dtype=x.dtype, device=x.device)
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous()):
        bias = bias.to(torch.bfloat16).contiguous()
    rc = load().dskd_lin256_fwd(x.data_ptr(), packed.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                tokens, n_out, 256, 1 if relu else 0, DTYPE_BF16, _stream(x))
    _check(rc, "dskd_lin256_fwd")
    global _ffn_flops
    _ffn_flops += 2 * tokens * 256 * n_out
    return y


# --------------------------------------------------------------------------- 1x1 convolution + epilogue (MFMA GEMM)
def conv1x1_ok(x: torch.Tensor, w: torch.Tensor, conv) -> bool:
    """Can csrc/gemm_nt.hip take this convolution: 1x1, stride 1 or 2, no padding / dilation / groups, on a channels_last
    bf16 CUDA activation with channel counts that are multiples of 64?"""
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and conv.kernel_size == (1, 1) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and x.shape[1] % 64 == 0 and w.shape[0] % 64 == 0
            and x.numel() > 0 and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0 and (w.stride(1) == 1 or w.is_contiguous()) and w.stride(0) == w.shape[1])


def gemm_nt_2d_ok(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    """Can dskd_gemm_nt form ``x [M, K] @ w [N, K]^T (+ bias)`` for a SHORT bf16 CUDA input (the decoder's 1 200 query rows,
    the head branches' 7 200): 6.6-9 us per launch against 10-19 us for the library GEMM at these sizes
    (profiles/r03_lin_vs_gemm_microbench.txt).  Tall inputs keep lin256 / the fused FFN / the library (faster there)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2 and w.dtype == torch.bfloat16
            and 0 < x.shape[0] < 16384 and x.shape[1] == w.shape[1] and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
            and x.is_contiguous() and w.is_contiguous() and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.data_ptr() % 16 == 0
                                  and bias.numel() == w.shape[0])))


def gemm_nt_2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False) -> torch.Tensor:
    """``act(x @ w^T + bias)`` through dskd_gemm_nt for inputs :func:`gemm_nt_2d_ok` accepts (raw, no autograd)."""
    y = torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device)
    return gemm_nt_raw(x, w, bias, None, x.shape[0], w.shape[0], w.shape[1], relu, y)


_gemm_ws_cache = {}


def _gemm_scratch(t: torch.Tensor):
    """(pointer, bytes) of the split-K scratch of dskd_gemm_nt_ws / dskd_conv3x3_ws for the CURRENT stream of ``t``'s device:
    one buffer per (device, stream) -- the teacher's stream and the training stream run convolutions side by side.  The
    partial tiles live only between the two launches of one call.  A stream first seen during a hipGraph capture gets no
    scratch (NULL: the library then never splits; same results, the last round of the grid is just emptier)."""
    key = (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)
    ws = _gemm_ws_cache.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None, 0
        ws = _gemm_ws_cache[key] = torch.empty(int(load().dskd_gemm_nt_scratch_bytes()), dtype=torch.uint8, device=t.device)
    return ws.data_ptr(), ws.numel()


def gemm_nt_raw(x, w2d, bias, res, M, N, K, relu, out, stride=0, Ho=0, Wo=0, Hi=0, Wi=0, gate=None):
    """``out[M, N] = act(x[M, K] w2d[N, K]^T + bias (+ res))``, zeroed where ``gate <= 0`` -- raw launch of dskd_gemm_nt_ws
    (bf16, no autograd)."""
    ws, ws_bytes = _gemm_scratch(x)
    rc = load().dskd_gemm_nt_ws(x.data_ptr(), w2d.data_ptr(), None if bias is None else bias.data_ptr(),
                                None if res is None else res.data_ptr(), None if gate is None else gate.data_ptr(),
                                out.data_ptr(), M, N, K, 1 if relu else 0, stride, Ho, Wo, Hi, Wi, DTYPE_BF16, ws, ws_bytes,
                                _stream(x))
    _check(rc, "dskd_gemm_nt_ws")
    global _ffn_flops
    _ffn_flops += 2 * M * N * K
    return out


class _Conv1x1Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, identity, relu, stride):
        B, K, H, W = x.shape
        N = w.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((B, N, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        gemm_nt_raw(x, w, bias, identity, B * Ho * Wo, N, K, relu, y, *((0, 0, 0, 0, 0) if stride == 1 else
                                                                       (stride, Ho, Wo, H, W)))
        ctx.relu, ctx.stride = relu, stride
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        need = ctx.needs_input_grad
        gx = gw = gb = None
        B, K, H, W = x.shape
        N = w.shape[0]
        if need[0]:
            if ctx.stride == 1:         # dX = dY W: the same kernel on the transposed weight
                wt = w.reshape(N, K).t().contiguous()
                gx = torch.empty_like(x)
                gemm_nt_raw(g, wt, None, None, B * H * W, K, N, False, gx)
            else:
                gx = torch.ops.aten.convolution_backward(g, x, w, None, [ctx.stride] * 2, [0, 0], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if need[1]:
            g2, x2 = g.permute(0, 2, 3, 1).reshape(-1, N), x.permute(0, 2, 3, 1).reshape(-1, K)     # views: NHWC rows
            if ctx.stride == 1 and gemm_tn_ok(g2, x2):
                # dW = dY^T X over the B*H*W rows: the split-K MFMA kernel (the library's weight-gradient convolution comes
                # with workspace zero-fill / cast helper launches around it)
                gw = (gemm_tn_bf16(g2, x2) if w.dtype == torch.bfloat16 else gemm_tn(g2, x2).to(w.dtype)).view(N, K, 1, 1)
                if w.stride() != gw.stride():
                    gw = gw.as_strided(w.shape, w.stride())
            else:
                gw = torch.ops.aten.convolution_backward(g, x, w, None, [ctx.stride] * 2, [0, 0], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        if need[2]:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, (g if need[3] else None), None, None


def conv1x1(x, w, bias=None, identity=None, relu=False, stride=1):
    """``act(conv2d(x, w, stride) + bias (+ identity))`` for a 1x1 convolution that :func:`conv1x1_ok` accepts: ONE MFMA
    launch (csrc/gemm_nt.hip) instead of the library convolution plus an epilogue pass; dX through the same kernel, dW
    through the library's weight-gradient convolution.  Reference: Bottleneck.forward, resnet.py:271-303."""
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous()):
        bias = bias.to(torch.bfloat16).contiguous()
    if identity is not None and (identity.dtype != x.dtype or not identity.is_contiguous(memory_format=torch.channels_last)):
        identity = identity.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return _Conv1x1Function.apply(x, w, bias, identity, bool(relu), int(stride))


def gemm_tn_ok(g2: torch.Tensor, x2: torch.Tensor) -> bool:
    """Can dskd_gemm_tn form ``g2^T @ x2`` ([M, N]^T [M, K], bf16 CUDA rows with unit column stride; N, K multiples of 128)?"""
    return (g2.is_cuda and g2.dim() == 2 and x2.dim() == 2 and g2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16
            and g2.shape[0] == x2.shape[0] and g2.shape[1] % 128 == 0 and x2.shape[1] % 128 == 0 and g2.stride(1) == 1
            and x2.stride(1) == 1 and g2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0 and g2.shape[0] >= 1024
            and g2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0)


def gemm_tn(g2: torch.Tensor, x2: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``g2^T @ x2`` in f32 (the weight gradient of ``y = x W^T``: dW = dY^T X) as ONE split-K MFMA launch with transposing LDS
    reads (csrc/gemm_nt.hip::gemm_tn_kernel).  ``out`` [N, K] f32: accumulated onto; default: a zero-filled buffer."""
    M, N = g2.shape
    K = x2.shape[1]
    if out is None:
        out = zeros((N, K), torch.float32, g2.device)
    rc = load().dskd_gemm_tn(g2.data_ptr(), x2.data_ptr(), out.data_ptr(), M, N, K, g2.stride(0), x2.stride(0), DTYPE_BF16,
                             _stream(g2))
    _check(rc, "dskd_gemm_tn")
    global _ffn_flops
    _ffn_flops += 2 * M * N * K
    return out


_tn_acc = {}          # (N, K, device, stream) -> f32 accumulators of gemm_tn_bf16_atomic (kept for A/B runs and tests)
_tn_scratch = {}      # (device, stream) -> the split-K scratch of gemm_tn_bf16


def gemm_tn_bf16_atomic(g2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """The round-3 form of :func:`gemm_tn_bf16`: the split-K kernel adds into a PERSISTENT f32 accumulator of that shape
    with float atomics, ``dskd_cvt_clear`` hands the result over as bf16 and zeroes the accumulator again."""
    N, K = g2.shape[1], x2.shape[1]
    key = (N, K, g2.device, _stream(g2))
    acc = _tn_acc.get(key)
    if acc is None:
        if torch.cuda.is_current_stream_capturing():
            acc = zeros((N, K), torch.float32, g2.device)            # the graph's own (see _persistent_acc)
        else:
            acc = _tn_acc[key] = torch.zeros((N, K), dtype=torch.float32, device=g2.device)
    with _acc_guard(acc):
        gemm_tn(g2, x2, out=acc)
        out = torch.empty((N, K), dtype=torch.bfloat16, device=g2.device)
        _check(load().dskd_cvt_clear(acc.data_ptr(), out.data_ptr(), N * K, DTYPE_BF16, _stream(g2)), "dskd_cvt_clear")
    return out


def _tn_ws(dev, need):
    """The split-K scratch of gemm_tn_bf16 / conv3x3_wgrad for the current stream (inside a capture: the graph's own)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _tn_scratch.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(need, dtype=torch.uint8, device=dev)     # from the graph's own pool: lives with the graph
        else:
            ws = _tn_scratch[key] = torch.empty(max(need, 32 << 20), dtype=torch.uint8, device=dev)
    return ws


def conv3x3_wgrad_ok(g: torch.Tensor, x: torch.Tensor, stride: int) -> bool:
    """Can dskd_conv3x3_wgrad take this weight gradient: channels_last bf16 CUDA dY [B, N, Ho, Wo] and input [B, C, Hi, Wi],
    C and N multiples of 128, stride 1 or 2?"""
    cl = torch.channels_last
    return (g.is_cuda and x.is_cuda and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and g.dim() == 4 and x.dim() == 4
            and g.shape[1] % 128 == 0 and x.shape[1] % 128 == 0 and stride in (1, 2) and g.shape[0] == x.shape[0]
            and g.shape[2] == (x.shape[2] - 1) // stride + 1 and g.shape[3] == (x.shape[3] - 1) // stride + 1
            and g.is_contiguous(memory_format=cl) and x.is_contiguous(memory_format=cl)
            and g.shape[0] * g.shape[2] * g.shape[3] < (1 << 24) and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def conv3x3_wgrad(g: torch.Tensor, x: torch.Tensor, stride: int, want_bias: bool = False):
    """d(weight) of ``conv2d(x, w, stride, padding=1)`` for a 3x3 kernel: [N, C, 3, 3] channels_last bf16, two launches
    (split-K products + fixed-order reduction), deterministic (dskd_conv3x3_wgrad)."""
    _need_gpu(g, x)
    B, Cc, Hi, Wi = x.shape
    N = g.shape[1]
    need = int(load().dskd_conv3x3_wgrad_scratch_bytes(B, Hi, Wi, Cc, N, stride))
    if need < 0:
        raise NativeError("conv3x3_wgrad: " + load().dskd_last_error().decode())
    ws = _tn_ws(x.device, need)
    dw = torch.empty((N, Cc, 3, 3), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    global _ffn_flops
    _ffn_flops += 2 * B * g.shape[2] * g.shape[3] * N * 9 * Cc
    if want_bias:        # (dW, db): the sums of g over the pixels from the same two launches
        db = torch.empty((N,), dtype=torch.bfloat16, device=x.device)
        rc = load().dskd_conv3x3_wgrad_bias(g.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), B,
                                            Hi, Wi, Cc, N, stride, DTYPE_BF16, _stream(x))
        _check(rc, "dskd_conv3x3_wgrad_bias")
        return dw, db
    rc = load().dskd_conv3x3_wgrad(g.data_ptr(), x.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), B, Hi, Wi, Cc, N,
                                   stride, DTYPE_BF16, _stream(x))
    _check(rc, "dskd_conv3x3_wgrad")
    return dw


def gemm_tn_bf16(g2: torch.Tensor, x2: torch.Tensor, want_bias: bool = False):
    """``(g2^T @ x2).to(bf16)`` -- a weight gradient in the low-precision parameter's dtype -- in two launches without
    atomics: the split-K kernel writes every split's partial product into a persistent scratch ([splits, N, K] f32, plain
    stores), a second launch sums the planes in a fixed order and casts (dskd_gemm_tn_bf16).  The float-atomic flush of the
    earlier form ran at the chip's ~1.3 TB/s atomic rate: 12 us of every launch.  One scratch per (device, stream)."""
    _need_gpu(g2, x2)
    M, N = g2.shape
    K = x2.shape[1]
    need = int(load().dskd_gemm_tn_scratch_bytes(M, N, K))
    if need < 0:
        raise NativeError("gemm_tn_bf16: " + load().dskd_last_error().decode())
    dev = g2.device
    ws = _tn_ws(dev, need)
    out = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
    global _ffn_flops
    _ffn_flops += 2 * M * N * K
    if want_bias:         # (dW, db): the column sums of g2 come out of the same two launches (dskd_gemm_tn_bias_bf16)
        db = torch.empty((N,), dtype=torch.bfloat16, device=dev)
        rc = load().dskd_gemm_tn_bias_bf16(g2.data_ptr(), x2.data_ptr(), out.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                           M, N, K, g2.stride(0), x2.stride(0), DTYPE_BF16, _stream(g2))
        _check(rc, "dskd_gemm_tn_bias_bf16")
        return out, db
    rc = load().dskd_gemm_tn_bf16(g2.data_ptr(), x2.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), M, N, K,
                                  g2.stride(0), x2.stride(0), DTYPE_BF16, _stream(g2))
    _check(rc, "dskd_gemm_tn_bf16")
    return out


def conv3x3_ok(x: torch.Tensor, w: torch.Tensor, conv) -> bool:
    """Can dskd_conv3x3 take this convolution: 3x3, padding 1, stride 1 or 2, no dilation / groups, channels_last bf16 CUDA
    activation and channels_last weight, C in {64, 128, .. 1024}, Cout a multiple of 64?"""
    Cin = x.shape[1] if x.dim() == 4 else 0
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and conv.kernel_size == (3, 3) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and Cin in (64, 128, 256, 512, 1024) and w.shape[0] % 64 == 0
            and x.numel() > 0 and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0
            and w.is_contiguous(memory_format=torch.channels_last) and w.data_ptr() % 16 == 0
            and x.shape[2] * x.shape[3] * Cin * 2 < 2 ** 31 - 1)


def conv3x3_raw(x, w, bias, res, relu, stride, out=None, gate=None):
    """Raw launch of dskd_conv3x3_ws (no autograd): x [B, C, H, W] channels_last, w [N, C, 3, 3] channels_last."""
    B, Cin, H, W = x.shape
    N = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, N, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ws, ws_bytes = _gemm_scratch(x)
    rc = load().dskd_conv3x3_ws(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                None if res is None else res.data_ptr(), None if gate is None else gate.data_ptr(),
                                out.data_ptr(), B, H, W, Cin, N, stride, 1 if relu else 0, DTYPE_BF16, ws, ws_bytes,
                                _stream(x))
    _check(rc, "dskd_conv3x3_ws")
    global _ffn_flops
    _ffn_flops += 2 * B * Ho * Wo * N * 9 * Cin
    return out


CONV3_WGRAD = not os.environ.get("DSKD_CONV3_WGRAD_LIB")      # A/B switch: MIOpen's weight gradient for the 3x3 convolutions


def _dw3x3(g, x, w, stride, want_bias=False):
    """d(weight) of a 3x3 convolution: the split-K MFMA kernel where its shape rules hold (C, N multiples of 128: ResNet
    stages 2-4), the library otherwise (stage 1: 64 channels).  ``want_bias``: (dW, sum of g over batch and pixels)."""
    if CONV3_WGRAD and w.dtype == torch.bfloat16 and conv3x3_wgrad_ok(g, x, stride):
        return conv3x3_wgrad(g, x, stride, want_bias)
    gw = torch.ops.aten.convolution_backward(g, x, w, None, [stride] * 2, [1, 1], [1, 1], False, [0, 0], 1,
                                             [False, True, False])[1]
    return (gw, g.sum((0, 2, 3))) if want_bias else gw


class _Conv3x3Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, identity, relu, stride):
        y = conv3x3_raw(x, w, bias, identity, relu, stride)
        ctx.relu, ctx.stride = relu, stride
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        need = ctx.needs_input_grad
        gx = gw = gb = None
        N, Cin = w.shape[0], w.shape[1]
        if need[0]:
            if ctx.stride == 1 and N in (64, 128, 256, 512, 1024) and Cin % 64 == 0:
                # dX = conv3x3(dY, W') with the taps flipped and the channel roles swapped: W'[c][ky][kx][n] = W[n][2-ky][2-kx][c]
                wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
                gx = conv3x3_raw(g, wt, None, None, False, 1)
            else:
                gx = torch.ops.aten.convolution_backward(g, x, w, None, [ctx.stride] * 2, [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if need[1]:
            gw = _dw3x3(g, x, w, ctx.stride)
        if need[2]:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, (g if need[3] else None), None, None


def conv3x3(x, w, bias=None, identity=None, relu=False, stride=1):
    """``act(conv2d(x, w, stride, padding=1) + bias (+ identity))`` for a 3x3 convolution that :func:`conv3x3_ok` accepts:
    one implicit-GEMM MFMA launch (csrc/gemm_nt.hip) with the epilogue fused; dX (stride 1) through the same kernel, dW
    through the library.  Reference: conv2 -> bn2 -> relu of Bottleneck.forward, resnet.py:283-288."""
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous()):
        bias = bias.to(torch.bfloat16).contiguous()
    if identity is not None and (identity.dtype != x.dtype or not identity.is_contiguous(memory_format=torch.channels_last)):
        identity = identity.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return _Conv3x3Function.apply(x, w, bias, identity, bool(relu), int(stride))


# --------------------------------------------------------------------------- multi-tensor cast (+ per-row scale)
class MultiCast:
    """One multi-tensor cast of dskd_cast_scale_many: ``dst_t = src_t * scale_t[row]`` for a list of tensors in ONE launch
    (direction 0: f32 -> bf16, 1: bf16 -> f32).  The device table is rebuilt only when an address changes (parameters and
    their persistent copies never move; gradients usually come back at the same addresses: caching allocator)."""

    def __init__(self, direction: int):
        self.direction, self.key, self.table, self.first, self.n, self.chunks = direction, None, None, None, 0, 0
        self.captured, self.retired = False, []

    @staticmethod
    def ok(srcs, dsts, scales, direction) -> bool:
        sd, dd = (torch.float32, torch.bfloat16) if direction == 0 else (torch.bfloat16, torch.float32)
        if not srcs or len(srcs) != len(dsts) or len(srcs) != len(scales):
            return False
        dev = srcs[0].device
        for s, d, sc in zip(srcs, dsts, scales):
            if not (s.is_cuda and s.device == dev and d.device == dev and s.dtype == sd and d.dtype == dd
                    and s.shape == d.shape and s.stride() == d.stride() and s.numel() > 0
                    and (s.is_contiguous() or s.is_contiguous(memory_format=torch.channels_last))):
                return False
            if sc is not None and not (sc.device == dev and sc.dtype == torch.float32 and sc.is_contiguous()
                                       and sc.numel() > 0 and s.numel() % sc.numel() == 0):
                return False
        return True

    @staticmethod
    def _key(srcs, dsts, scales):
        return tuple((s.data_ptr(), d.data_ptr(), 0 if sc is None else sc.data_ptr(), s.numel(), 0 if sc is None else sc.numel())
                     for s, d, sc in zip(srcs, dsts, scales))

    def ready(self, srcs, dsts, scales) -> bool:
        """:meth:`ok`, and the launch can be issued NOW: a changed address means a new table upload (pinned allocation +
        host-to-device copy), which a stream capture forbids -- the caller then takes its per-tensor path for that capture."""
        if not MultiCast.ok(srcs, dsts, scales, self.direction):
            return False
        return not torch.cuda.is_current_stream_capturing() or self._key(srcs, dsts, scales) == self.key

    def run(self, srcs, dsts, scales):
        key = self._key(srcs, dsts, scales)
        dev = srcs[0].device
        if key != self.key:
            if torch.cuda.is_current_stream_capturing():
                raise NativeError("MultiCast.run: new addresses inside a stream capture (ask ready() first)")
            if self.captured:           # a captured launch reads this table on every replay: leave it as it is
                self.retired.append((self.table, self.first))
                self.table, self.captured = None, False
            chunk = int(load().dskd_cast_scale_chunk())
            rows, first = [], [0]
            for sp, dp, cp, n, rows_sc in key:
                rows += [sp, dp, cp, n, n // rows_sc if rows_sc else n]
                first.append(first[-1] + (n + chunk - 1) // chunk)
            # a NEW pinned source per upload (the host runs ahead of the GPU: see optim.FusedClipAdamW)
            t_host = torch.tensor(rows + first, dtype=torch.int64).pin_memory()
            if self.table is None or self.table.numel() != t_host.numel() or self.table.device != dev:
                self.table = torch.empty(t_host.numel(), dtype=torch.int64, device=dev)
                self.first = torch.empty(len(first), dtype=torch.int32, device=dev)
            self.table.copy_(t_host, non_blocking=True)
            self.first.copy_(self.table[len(rows):], non_blocking=True)         # int64 -> int32 on the device
            self.key, self.n, self.chunks = key, len(key), first[-1]
        if torch.cuda.is_current_stream_capturing():
            self.captured = True
        _check(load().dskd_cast_scale_many(self.table.data_ptr(), self.first.data_ptr(), self.n, self.chunks, self.direction,
                                           _stream(srcs[0])), "dskd_cast_scale_many")


class WeightTransposes:
    """``dskd_weight_t_many`` for a fixed list of convolution weights (one ResNet stage): the operands of the input-gradient
    launches -- ``w.t()`` of a 1x1 weight, ``w.flip(2, 3).transpose(0, 1)`` (channels_last) of a 3x3 one -- in ONE launch per
    step instead of a flip + a strided copy per convolution in every Bottleneck's backward.  The device table is rebuilt only
    when an address changes; not used inside a stream capture whose addresses it has not seen."""

    def __init__(self):
        self.key, self.table, self.first, self.n, self.blocks = None, None, None, 0, 0

    @staticmethod
    def eligible(w) -> bool:
        if not (w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 4 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
                and w.data_ptr() % 16 == 0):
            return False
        if tuple(w.shape[2:]) == (1, 1):
            return w.stride(1) == 1 and w.stride(0) == w.shape[1]
        return tuple(w.shape[2:]) == (3, 3) and w.is_contiguous(memory_format=torch.channels_last)

    def run(self, ws):
        """[transposed weight | None] per entry of ``ws``: [K, N] for 1x1, [K, N, 3, 3] channels_last for 3x3."""
        idx = [i for i, w in enumerate(ws) if self.eligible(w)]
        outs = [None] * len(ws)
        if not idx:
            return outs
        for i in idx:
            w = ws[i]
            N, K = w.shape[0], w.shape[1]
            outs[i] = torch.empty((K, N), dtype=w.dtype, device=w.device) if w.shape[2] == 1 else \
                torch.empty((K, N, 3, 3), dtype=w.dtype, device=w.device, memory_format=torch.channels_last)
        key = tuple((ws[i].data_ptr(), outs[i].data_ptr(), tuple(ws[i].shape)) for i in idx)
        if key != self.key:
            if torch.cuda.is_current_stream_capturing():
                return [None] * len(ws)               # the callers make their own copies (capturable)
            rows, first = [], [0]
            for i in idx:
                N, K, taps = ws[i].shape[0], ws[i].shape[1], ws[i].shape[2] * ws[i].shape[3]
                rows += [ws[i].data_ptr(), outs[i].data_ptr(), N, K, taps]
                first.append(first[-1] + taps * (N // 64) * (K // 64))
            t_host = torch.tensor(rows + first, dtype=torch.int64).pin_memory()
            dev = ws[idx[0]].device
            self.table = torch.empty(t_host.numel(), dtype=torch.int64, device=dev)
            self.first = torch.empty(len(first), dtype=torch.int32, device=dev)
            self.table.copy_(t_host, non_blocking=True)
            self.first.copy_(self.table[len(rows):], non_blocking=True)
            self.key, self.n, self.blocks = key, len(idx), first[-1]
        _check(load().dskd_weight_t_many(self.table.data_ptr(), self.first.data_ptr(), self.n, self.blocks, DTYPE_BF16,
                                         _stream(ws[idx[0]])), "dskd_weight_t_many")
        return outs


# --------------------------------------------------------------------------- a whole Bottleneck, backward fused
def gemm_nt_dx_raw(g, wt2d, res, gate, M, N, K, out):
    """``out[M, N] = (gate > 0) ? g[M, K] wt2d[N, K]^T + res : 0`` -- the input-gradient form of dskd_gemm_nt_ws."""
    return gemm_nt_raw(g, wt2d, None, res, M, N, K, False, out, gate=gate)


def conv3x3_dx_raw(g, wt, gate):
    """``(gate > 0) ? conv3x3(g, wt, stride 1, padding 1) : 0`` -- the input-gradient form of dskd_conv3x3_ws; wt
    [C_out_of_dx, C_in_of_dx, 3, 3] channels_last = the forward weight with the taps flipped and the channel roles swapped."""
    return conv3x3_raw(g, wt, None, None, False, 1, gate=gate)


def _rows(t):          # [B, C, H, W] channels_last -> its NHWC rows [B*H*W, C] (a view)
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def _dw1x1(g, x, w, stride, want_bias=False):
    """Weight gradient of a 1x1 convolution y = conv(x, w, stride) from g = dL/dy.  ``want_bias``: (dW, sum of g over batch
    and pixels) -- from the same launch pair where the split-K kernel runs."""
    N, K = w.shape[0], w.shape[1]
    g2, x2 = _rows(g), _rows(x)
    if stride == 1 and gemm_tn_ok(g2, x2):
        gb = None
        if w.dtype == torch.bfloat16 and want_bias:
            gw, gb = gemm_tn_bf16(g2, x2, want_bias=True)
        else:
            gw = gemm_tn_bf16(g2, x2) if w.dtype == torch.bfloat16 else gemm_tn(g2, x2).to(w.dtype)
        gw = gw.view(N, K, 1, 1)
        gw = gw.as_strided(w.shape, w.stride()) if w.stride() != gw.stride() else gw
        if want_bias:
            return gw, (gb if gb is not None else g.sum((0, 2, 3)))
        return gw
    gw = torch.ops.aten.convolution_backward(g, x, w, None, [stride] * 2, [0, 0], [1, 1], False, [0, 0], 1,
                                             [False, True, False])[1]
    return (gw, g.sum((0, 2, 3))) if want_bias else gw


class _BottleneckFunction(torch.autograd.Function):
    """conv1 (1x1) -> relu -> conv2 (3x3, stride s) -> relu -> conv3 (1x1) + identity -> relu of a ResNet Bottleneck with the
    BatchNorms folded (mmdet/models/backbones/resnet.py:271-303), forward on the kernels of conv1x1 / conv3x3 and the
    backward written out so that what autograd would run as separate passes over the activations rides in the epilogue of
    the input-gradient GEMMs: the ReLU masks (``threshold_backward`` of y1, y2 and -- for a block fed by another block --
    of the block input) and the sum of the identity path's gradient with conv1's.  The returned input gradient is then
    already masked by ``x > 0``; it is tagged with (pointer of ``x``, its own pointer, its version counter) so that the
    producing block (whose output IS x) skips its own mask.  Masking twice is the same as once, so a lost tag costs a pass;
    a tag that SURVIVES a sum would be wrong (autograd adds further consumers of x into the first-arrived gradient in
    place, keeping the Python object): the version counter in the tag catches that, the sum is masked again."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, wd, bd, stride, dstride, x_is_relu, wts=None):
        """``wts``: (w1^T, w2 flipped / channel-swapped, w3^T, wd^T) made for the whole stage by :class:`WeightTransposes`
        (entries may be None: the backward then makes its own copy)."""
        B, Cin, H, W = x.shape
        P, N = w1.shape[0], w3.shape[0]
        y1 = torch.empty((B, P, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        gemm_nt_raw(x, w1, b1, None, B * H * W, P, Cin, True, y1)
        y2 = conv3x3_raw(y1, w2, b2, None, True, stride)
        Ho, Wo = y2.shape[2], y2.shape[3]
        if wd is not None:
            idn = torch.empty((B, N, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            gemm_nt_raw(x, wd, bd, None, B * Ho * Wo, N, Cin, False, idn,
                        *((0, 0, 0, 0, 0) if dstride == 1 else (dstride, Ho, Wo, H, W)))
        else:
            idn = x
        y = torch.empty((B, N, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        gemm_nt_raw(y2, w3, b3, idn, B * Ho * Wo, N, P, True, y)
        ctx.stride, ctx.dstride, ctx.x_is_relu, ctx.has_down = stride, dstride, x_is_relu, wd is not None
        ctx.wts = tuple(wts) if wts is not None else (None, None, None, None)
        ctx.save_for_backward(x, y1, y2, y, w1, w2, w3, wd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y1, y2, y, w1, w2, w3, wd = ctx.saved_tensors
        w1t, w2t, w3t, wdt = ctx.wts
        need = ctx.needs_input_grad
        # the tag is honoured only on the very tensor it was put on, unmodified since: autograd sums several consumers of y
        # IN PLACE into the first gradient that arrived (InputBuffer: old_var.add_(var) keeps the Python object and its
        # attributes), which bumps the version counter -- a sum that contains an unmasked addend must be masked here
        tag = getattr(g, "_dskd_relu_masked", None)
        masked = tag is not None and tag == (y.data_ptr(), g.data_ptr(), g._version)
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        g3 = g if masked else torch.ops.aten.threshold_backward(g, y, 0)
        B, Cin, H, W = x.shape
        P, N = w1.shape[0], w3.shape[0]
        Ho, Wo = y.shape[2], y.shape[3]
        s, ds = ctx.stride, ctx.dstride
        gw1 = gw2 = gw3 = gwd = gb1 = gb2 = gb3 = gbd = gx = None
        # conv3: dW3 = g3^T y2; dY2 = (g3 W3) masked by y2 > 0
        if need[5] and need[6]:
            gw3, gb3 = _dw1x1(g3, y2, w3, 1, want_bias=True)       # the bias sums ride in the dW launch pair
        else:
            if need[5]:
                gw3 = _dw1x1(g3, y2, w3, 1)
            if need[6]:
                gb3 = g3.sum((0, 2, 3))
        g2 = torch.empty_like(y2)
        gemm_nt_dx_raw(g3, w3t if w3t is not None else w3.reshape(N, P).t().contiguous(), None, y2, B * Ho * Wo, P, N, g2)
        # conv2: dW2 from the library; dY1 = conv3x3(g2, W2') masked by y1 > 0 (stride 1), the library's data gradient else
        if need[3] and need[4]:
            gw2, gb2 = _dw3x3(g2, y1, w2, s, want_bias=True)
        else:
            if need[3]:
                gw2 = _dw3x3(g2, y1, w2, s)
            if need[4]:
                gb2 = g2.sum((0, 2, 3))
        if s == 1 and P in (64, 128, 256, 512, 1024):
            g1 = conv3x3_dx_raw(g2, w2t if w2t is not None else
                                w2.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last), y1)
        else:
            g1 = torch.ops.aten.convolution_backward(g2, y1, w2, None, [s] * 2, [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
            g1 = torch.ops.aten.threshold_backward(g1, y1, 0)
        if need[1] and need[2]:
            gw1, gb1 = _dw1x1(g1, x, w1, 1, want_bias=True)
        else:
            if need[1]:
                gw1 = _dw1x1(g1, x, w1, 1)
            if need[2]:
                gb1 = g1.sum((0, 2, 3))
        # the identity path: g3 itself, or through the downsample convolution
        gid = g3
        if ctx.has_down:
            if need[7]:
                gwd = _dw1x1(g3, x, wd, ds)
            if need[8]:       # the downsample branch's output gradient IS g3: the same sums as d(b3)
                gbd = gb3 if gb3 is not None else g3.sum((0, 2, 3))
            if need[0]:
                if ds == 1:
                    gid = torch.empty_like(x)
                    gemm_nt_dx_raw(g3, wdt if wdt is not None else wd.reshape(N, Cin).t().contiguous(), None, None, B * H * W,
                                   Cin, N, gid)
                else:
                    gid = torch.ops.aten.convolution_backward(g3, x, wd, None, [ds] * 2, [0, 0], [1, 1], False, [0, 0], 1,
                                                              [True, False, False])[0]
                    if not gid.is_contiguous(memory_format=torch.channels_last):
                        gid = gid.contiguous(memory_format=torch.channels_last)
        # conv1: dX = (g1 W1 + identity gradient) masked by x > 0 when x is a ReLU output whose producer masks anyway
        if need[0]:
            gx = torch.empty_like(x)
            gemm_nt_dx_raw(g1, w1t if w1t is not None else w1.reshape(P, Cin).t().contiguous(), gid,
                           x if ctx.x_is_relu else None, B * H * W, Cin, P, gx)
            if ctx.x_is_relu:
                gx._dskd_relu_masked = (x.data_ptr(), gx.data_ptr(), gx._version)
        return gx, gw1, gb1, gw2, gb2, gw3, gb3, gwd, gbd, None, None, None, None


def bottleneck_ok(x, w1, w2, w3, wd, conv1, conv2, conv3, down) -> bool:
    """Can the fused Bottleneck take this block: every convolution one that conv1x1_ok / conv3x3_ok accept, conv1 and conv3
    of stride 1 (the 'pytorch' style: the stride sits on conv2), channel counts the input-gradient launches can take?"""
    if not (conv1x1_ok(x, w1, conv1) and conv1.stride == (1, 1) and conv3.stride == (1, 1) and conv3.kernel_size == (1, 1)
            and w3.dtype == torch.bfloat16 and w1.shape[0] % 64 == 0 and w3.shape[0] % 64 == 0
            and w3.shape[1] == w1.shape[0] and w2.shape[0] == w2.shape[1] == w1.shape[0]
            and w1.shape[0] in (64, 128, 256, 512, 1024) and w2.dtype == torch.bfloat16
            and conv2.kernel_size == (3, 3) and conv2.stride in ((1, 1), (2, 2)) and conv2.padding == (1, 1)
            and conv2.dilation == (1, 1) and conv2.groups == 1 and w2.is_contiguous(memory_format=torch.channels_last)
            and w2.data_ptr() % 16 == 0 and w3.data_ptr() % 16 == 0 and (w3.stride(1) == 1 or w3.is_contiguous())
            and w3.stride(0) == w3.shape[1] and x.shape[2] * x.shape[3] * w1.shape[0] * 2 < 2 ** 31 - 1):
        return False
    if down is None:
        return conv2.stride == (1, 1) and w3.shape[0] == x.shape[1]
    return conv1x1_ok(x, wd, down) and down.stride == conv2.stride and wd.shape[0] == w3.shape[0]


def bottleneck(x, w1, b1, w2, b2, w3, b3, wd=None, bd=None, stride=1, x_is_relu=False, wts=None):
    """One Bottleneck (folded BatchNorms) as ONE autograd node: see :class:`_BottleneckFunction`.  ``wts``: the operands of
    its input-gradient launches when the caller has them (:class:`WeightTransposes`)."""
    def bf(b):
        return b if b is None or (b.dtype == torch.bfloat16 and b.is_contiguous()) else b.to(torch.bfloat16).contiguous()
    return _BottleneckFunction.apply(x, w1, bf(b1), w2, bf(b2), w3, bf(b3), wd, bf(bd), int(stride), int(stride),
                                     bool(x_is_relu), wts)


# --------------------------------------------------------------------------- Swin window attention (MFMA kernels)
WINATTN_TOKENS, WINATTN_HEAD_DIM, WINATTN_NEG = 49, 32, -30000.0


def window_attention_ok(qkv: torch.Tensor, num_heads: int, tokens: int, dropout_p: float) -> bool:
    """Can csrc/winattn.hip take this call: 49-token windows, head dimension 32, bf16 CUDA projection output, no
    attention dropout?"""
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and tokens == WINATTN_TOKENS
            and qkv.shape[1] == tokens and qkv.shape[2] == 3 * num_heads * WINATTN_HEAD_DIM and dropout_p == 0.0
            and qkv.is_contiguous() and qkv.data_ptr() % 16 == 0)


def window_attention_table(bias: torch.Tensor, mask_types: Optional[torch.Tensor]) -> torch.Tensor:
    """The additive table the kernels read: [types][heads][64 keys][64 queries] f32 = bias[h][q][k] + mask[type][q][k],
    -30000 on the padded keys.  bias [heads, 49, 49] (query, key); mask_types [types, 49, 49] or None."""
    nH, N, _ = bias.shape
    t = bias.detach().float().transpose(1, 2).unsqueeze(0)                       # [1, h, k, q]
    if mask_types is not None:
        t = t + mask_types.float().transpose(1, 2).unsqueeze(1)                  # [types, h, k, q]
    out = torch.zeros((t.shape[0], nH, 64, 64), dtype=torch.float32, device=bias.device)
    out[:, :, N:, :] = WINATTN_NEG
    out[:, :, :N, :N] = t
    return out


class _WindowAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias, mask_types, wtype, num_heads, scale):
        Bw, N, _ = qkv.shape
        table = window_attention_table(bias, mask_types)
        nW = 1 if wtype is None else wtype.numel()
        out = torch.empty((Bw, N, num_heads * WINATTN_HEAD_DIM), dtype=qkv.dtype, device=qkv.device)
        rc = load().dskd_winattn_fwd(qkv.data_ptr(), table.data_ptr(), None if wtype is None else wtype.data_ptr(),
                                     out.data_ptr(), Bw, num_heads, nW, N, WINATTN_HEAD_DIM, scale, DTYPE_BF16, _stream(qkv))
        _check(rc, "dskd_winattn_fwd")
        ctx.save_for_backward(qkv, table, wtype)
        ctx.meta = (num_heads, scale, nW, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, table, wtype = ctx.saved_tensors
        num_heads, scale, nW, bdtype = ctx.meta
        Bw, N, _ = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dtable = zeros((num_heads, 64, 64), torch.float32, qkv.device)
        rc = load().dskd_winattn_bwd(qkv.data_ptr(), table.data_ptr(), None if wtype is None else wtype.data_ptr(),
                                     dout.data_ptr(), dqkv.data_ptr(), dtable.data_ptr(), Bw, num_heads, nW, N,
                                     WINATTN_HEAD_DIM, scale, DTYPE_BF16, _stream(qkv))
        _check(rc, "dskd_winattn_bwd")
        dbias = dtable[:, :N, :N].transpose(1, 2).to(bdtype) if ctx.needs_input_grad[1] else None      # [h, q, k]
        return dqkv, dbias, None, None, None, None


def window_attention(qkv: torch.Tensor, bias: torch.Tensor, mask_types: Optional[torch.Tensor],
                     wtype: Optional[torch.Tensor], num_heads: int, scale: float) -> torch.Tensor:
    """``softmax(q k^T * scale + bias (+ mask)) v`` per window and head (reference: WindowMSA.forward,
    mmdet/models/backbones/swin.py:81-126).  qkv [windows, 49, 3 * heads * 32] = the qkv Linear's output; bias
    [heads, 49, 49] (differentiable); mask_types [types, 49, 49] with wtype [windows per image] int32 (type of each window
    of an image) or both None.  Returns [windows, 49, heads * 32], the input of the output projection."""
    return _WindowAttentionFunction.apply(qkv, bias, mask_types, wtype, int(num_heads), float(scale))


# --------------------------------------------------------------------------- self-attention of the decoder's queries
ATTN_HEAD_DIM, ATTN_MAX_TOKENS = 32, 320


def self_attention_ok(qk: torch.Tensor, v: torch.Tensor, num_heads: int) -> bool:
    """Can csrc/attn.hip take this call: bf16 CUDA, heads of 32 channels, up to 320 tokens, q | k side by side in ``qk``
    [.., L, 2 E] and v [.., L, E] as the projections wrote them?"""
    E = num_heads * ATTN_HEAD_DIM
    return (qk.is_cuda and qk.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and qk.dim() == 3 and v.dim() == 3
            and qk.shape[-1] == 2 * E and v.shape[-1] == E and qk.shape[:2] == v.shape[:2]
            and qk.is_contiguous() and v.is_contiguous() and qk.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0)


class _SelfAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qk, v, num_heads, scale, p, batch_first):
        E = num_heads * ATTN_HEAD_DIM
        B, L = (qk.shape[0], qk.shape[1]) if batch_first else (qk.shape[1], qk.shape[0])
        if L > ATTN_MAX_TOKENS:
            raise NativeError(f"self_attention: {L} tokens, built for up to {ATTN_MAX_TOKENS}")
        # (batch, row) strides in elements of q, k (inside qk), v and the output
        rows = (L, 1) if batch_first else (1, B)
        strides = [rows[0] * 2 * E, rows[1] * 2 * E] * 2 + [rows[0] * E, rows[1] * E] * 2
        st = (C.c_int64 * 8)(*strides)
        train = any(ctx.needs_input_grad[:2])
        out = torch.empty_like(v)
        stats = torch.empty((B, num_heads, L, 2), dtype=torch.float32, device=qk.device) if train else None
        seed, offset = _next_drop_key() if p > 0 else (0, 0)
        ep = dropout_epoch(qk.device).data_ptr() if p > 0 else None
        rc = load().dskd_attn_fwd(qk.data_ptr(), qk.data_ptr() + 2 * E, v.data_ptr(), out.data_ptr(),
                                  None if stats is None else stats.data_ptr(), B, num_heads, L, ATTN_HEAD_DIM, st, scale, p,
                                  seed, offset, ep, DTYPE_BF16, _stream(qk))
        _check(rc, "dskd_attn_fwd")
        if train:
            ctx.save_for_backward(qk, v, out, stats)
        ctx.meta = (num_heads, scale, p, seed, offset, B, L, strides)
        return out

    @staticmethod
    def backward(ctx, dout):
        qk, v, out, stats = ctx.saved_tensors
        num_heads, scale, p, seed, offset, B, L, strides = ctx.meta
        E = num_heads * ATTN_HEAD_DIM
        dout = dout.contiguous()
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        delta = torch.empty((B, num_heads, L), dtype=torch.float32, device=qk.device)
        st = (C.c_int64 * 8)(*strides)
        ep = dropout_epoch(qk.device).data_ptr() if p > 0 else None
        rc = load().dskd_attn_bwd(qk.data_ptr(), qk.data_ptr() + 2 * E, v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                  stats.data_ptr(), delta.data_ptr(), dqk.data_ptr(), dqk.data_ptr() + 2 * E, dv.data_ptr(),
                                  B, num_heads, L, ATTN_HEAD_DIM, st, scale, p, seed, offset, ep, DTYPE_BF16, _stream(qk))
        _check(rc, "dskd_attn_bwd")
        return dqk, dv, None, None, None, None


def self_attention(qk: torch.Tensor, v: torch.Tensor, num_heads: int, dropout_p: float = 0.0,
                   batch_first: bool = True) -> torch.Tensor:
    """``dropout(softmax(q k^T / sqrt(32))) v`` per (image, head): the core of ext-mmcv ``MultiheadAttention`` /
    ``nn.MultiheadAttention`` for the decoder's object queries (self-attention: configs/deformable_detr/*_il.py:82-87, run by
    mmdet/models/utils/transformer.py:639-709).  ``qk`` [B, L, 2 E] (or [L, B, 2 E]) = output of the joint q | k projection,
    ``v`` [B, L, E]; returns [B, L, E], the input of the output projection.  Gradients arrive as d(qk), d(v) in the same
    layouts: nothing is split, permuted or concatenated around the kernels (csrc/attn.hip)."""
    return _SelfAttentionFunction.apply(qk, v, int(num_heads), 1.0 / math.sqrt(ATTN_HEAD_DIM), float(dropout_p),
                                        bool(batch_first))


# --------------------------------------------------------------------------- GroupNorm of the neck
def _cl_rows(t: torch.Tensor):
    """Batch stride (elements) if ``t`` [B, C, H, W] is laid out as [b][h][w][c] rows with any batch stride, else None."""
    B, Cc, H, W = t.shape
    if t.stride(1) == 1 and t.stride(3) == Cc and t.stride(2) == W * Cc and t.stride(0) % 8 == 0 and \
            t.data_ptr() % 16 == 0:
        return t.stride(0)
    return None


class _GroupNormCLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[x.dtype]
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)                       # preserve_format: channels_last
        train = any(ctx.needs_input_grad[:3])
        sums = torch.empty(load().dskd_gn_workspace(B, H * W) // 8, dtype=torch.float64, device=x.device)
        stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        gamma_f, beta_f = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        rc = load().dskd_gn_fwd(x.data_ptr(), gamma_f.data_ptr(), beta_f.data_ptr(), y.data_ptr(), sums.data_ptr(),
                                stats.data_ptr(), B, H * W, Cc, groups, x.stride(0), y.stride(0), eps, 1 if relu else 0, dt,
                                _stream(x))
        _check(rc, "dskd_gn_fwd")
        if train:
            ctx.save_for_backward(x, stats, gamma_f, beta_f)
            ctx.meta = (dt, groups, gamma.dtype, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma_f, beta_f = ctx.saved_tensors
        dt, groups, gdtype, relu = ctx.meta
        B, Cc, H, W = x.shape
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy_bs = _cl_rows(dy)
        if dy_bs is None:
            dy = dy.contiguous(memory_format=torch.channels_last)
            dy_bs = dy.stride(0)
        dx = torch.empty_like(x)
        copies = _colsum_copies(B * H * W)
        sums = zeros((B, groups, 2), torch.float64, x.device)
        dgb = zeros((copies, 2, Cc), torch.float32, x.device)
        rc = load().dskd_gn_bwd(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma_f.data_ptr(), beta_f.data_ptr(),
                                dx.data_ptr(), sums.data_ptr(), dgb.data_ptr(), copies, B, H * W, Cc, groups, x.stride(0),
                                dy_bs, dx.stride(0), 1 if relu else 0, dt, _stream(x))
        _check(rc, "dskd_gn_bwd")
        dgb = dgb.sum(0) if copies > 1 else dgb[0]
        return dx, dgb[0].to(gdtype), dgb[1].to(gdtype), None, None, None


def nchw_f32(t: torch.Tensor) -> torch.Tensor:
    """``t.detach().contiguous().float()`` of a [B, C, H, W] map; one transposing launch when ``t`` is a 256-channel
    channels_last f32 | bf16 CUDA tensor (what the neck's GroupNorm kernel hands over), ATen otherwise."""
    t = t.detach()
    if t.is_cuda and t.dim() == 4 and t.shape[1] == 256 and t.dtype in (torch.float32, torch.bfloat16) and t.numel() > 0 \
            and not t.is_contiguous():
        bs = _cl_rows(t)
        if bs is not None:
            B, Cc, H, W = t.shape
            out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=t.device)
            dt = DTYPE_F32 if t.dtype == torch.float32 else DTYPE_BF16
            _check(load().dskd_nhwc_to_nchw_f32(t.data_ptr(), out.data_ptr(), B, H * W, Cc, bs, dt, _stream(t)),
                   "dskd_nhwc_to_nchw_f32")
            return out
    return t.contiguous().float()


def group_norm_cl_ok(x: torch.Tensor, gn: torch.nn.GroupNorm) -> bool:
    """Can csrc/gn.hip take this GroupNorm call (CUDA, channels_last rows, 256 channels in 32 groups, affine)?"""
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and gn.affine
            and gn.num_channels == 256 and gn.num_groups == 32 and x.shape[1] == 256 and x.numel() > 0
            and _cl_rows(x) == x.shape[1] * x.shape[2] * x.shape[3])


def group_norm_cl(x: torch.Tensor, gn: torch.nn.GroupNorm, relu: bool = False) -> torch.Tensor:
    """``gn(x)`` (``relu(gn(x))`` with ``relu``) for a channels_last [B, 256, H, W] activation in two streaming passes
    each way; the result has x's dtype and memory format (under autocast: bf16 in, bf16 out -- ATen would cast to f32,
    copy to NCHW and back)."""
    return _GroupNormCLFunction.apply(x, gn.weight, gn.bias, gn.num_groups, float(gn.eps), bool(relu))


# --------------------------------------------------------------------------- conv epilogue
class _BiasActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, identity, relu):
        dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[x.dtype]
        Cc = x.shape[1]
        rc = load().dskd_bias_act(x.data_ptr(), bias.data_ptr(), None if identity is None else identity.data_ptr(),
                                  x.numel(), Cc, 1 if relu else 0, dt, _stream(x))
        _check(rc, "dskd_bias_act")
        ctx.mark_dirty(x)
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, g):
        if ctx.relu:
            (y,) = ctx.saved_tensors
            g = torch.ops.aten.threshold_backward(g, y, 0)
        gb = g.sum((0, 2, 3)) if ctx.needs_input_grad[1] else None
        return g, gb, (g if ctx.needs_input_grad[2] else None), None


def bias_act(x: torch.Tensor, bias: torch.Tensor, identity: Optional[torch.Tensor] = None, relu: bool = False):
    """``act(x + bias[None, :, None, None] (+ identity))`` IN PLACE on ``x`` (a convolution
    output in channels_last memory): the tail of ResNet's conv -> folded BN -> (+ identity) ->
    ReLU (reference mmdet/models/backbones/resnet.py:271-303) as one pass instead of up to four."""
    f = _dispatch_cpu("bias_act", x)
    if f is not None:
        return f(x, bias, identity, relu)
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16) or x.shape[1] % vec != 0 \
            or not x.is_contiguous(memory_format=torch.channels_last) or bias.dtype != x.dtype \
            or (identity is not None and (identity.dtype != x.dtype or identity.shape != x.shape or
                                          not identity.is_contiguous(memory_format=torch.channels_last))):
        y = x + bias.to(x.dtype).view(1, -1, 1, 1)          # layouts the kernel does not take
        if identity is not None:
            y = y + identity
        return torch.relu_(y) if relu else y
    return _BiasActFunction.apply(x, bias.contiguous(), identity, relu)


def bias_relu_maxpool_ok(x: torch.Tensor, bias: torch.Tensor) -> bool:
    vec = 8 if x.dtype == torch.bfloat16 else 4
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % vec == 0
            and x.is_contiguous(memory_format=torch.channels_last) and bias.dtype == x.dtype and bias.numel() == x.shape[1]
            and not (torch.is_grad_enabled() and (x.requires_grad or bias.requires_grad)))


def bias_relu_maxpool(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``max_pool2d(relu(x + bias[None, :, None, None]), 3, 2, 1)`` in one pass for a convolution output that needs no
    gradient (the stem of the teacher and of a student with frozen_stages >= 0; resnet.py:633-640).  Raw op: no autograd."""
    _need_gpu(x, bias)
    B, Cc, H, W = x.shape
    y = torch.empty((B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = load().dskd_bias_relu_maxpool(x.data_ptr(), bias.contiguous().data_ptr(), y.data_ptr(), B, H, W, Cc,
                                       DTYPE_BF16 if x.dtype == torch.bfloat16 else DTYPE_F32, _stream(x))
    _check(rc, "dskd_bias_relu_maxpool")
    return y


# --------------------------------------------------------------------------- LSAP
def lsap_host(cost: torch.Tensor):
    """scipy-bit-exact linear_sum_assignment on a host float32 matrix (C-ABI host entry)."""
    cost = cost.detach().to("cpu", torch.float32).contiguous()
    nr, nc = cost.shape
    n = min(nr, nc)
    row = torch.empty(n, dtype=torch.int64)
    col = torch.empty(n, dtype=torch.int64)
    rc = load().dskd_lsap_host(cost.data_ptr(), nr, nc, row.data_ptr(), col.data_ptr())
    _check(rc, "dskd_lsap_host")
    return row, col


def lsap_batched(cost_flat: torch.Tensor, nr: Sequence[int], nc: Sequence[int], offsets: Sequence[int]):
    """Solve len(nr) problems stored back to back in ``cost_flat`` (device f32) in one launch.
    Returns (row, col, out_offsets, status) with row/col device int64 and status device int32."""
    _need_gpu(cost_flat)
    nprob = len(nr)
    outs, acc = [], 0
    for r, c in zip(nr, nc):
        outs.append(acc)
        acc += min(int(r), int(c))
    row = torch.empty(max(acc, 1), dtype=torch.int64, device=cost_flat.device)
    col = torch.empty(max(acc, 1), dtype=torch.int64, device=cost_flat.device)
    status = zeros(max(nprob, 1) * 4, torch.int32, cost_flat.device)[:max(nprob, 1)]
    rc = load().dskd_lsap_batched(cost_flat.data_ptr(), _host_i32(nr), _host_i32(nc), _host_i64(offsets), nprob,
                                  row.data_ptr(), col.data_ptr(), _host_i64(outs), status.data_ptr(),
                                  _stream(cost_flat))
    _check(rc, "dskd_lsap_batched")
    return row, col, outs, status


def raise_for_lsap_status(status: torch.Tensor) -> None:
    """Host check of the per-problem status words (one sync; call off the critical path)."""
    st = status.cpu()
    if (st == ERR_INVALID_COST).any():
        raise ValueError("matrix contains invalid numeric entries")
    if (st == ERR_INFEASIBLE).any():
        raise ValueError("cost matrix is infeasible")


# --------------------------------------------------------------------------- matching cost
def match_cost(bbox_pred: torch.Tensor, cls_pred: torch.Tensor, gt_bboxes: torch.Tensor,
               gt_labels: torch.Tensor, gt_start: Sequence[int], img_wh: Sequence[Tuple[float, float]],
               w_cls: float, w_reg: float, w_iou: float) -> torch.Tensor:
    """bbox_pred [P,Q,4], cls_pred [P,Q,C], gt_* concatenated over the P problems.
    Returns the flat cost buffer; problem p is [Q, G_p] at Q*gt_start[p]."""
    f = _dispatch_cpu("match_cost", bbox_pred)
    if f is not None:
        return f(bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_start, img_wh, w_cls, w_reg, w_iou)
    _need_gpu(cls_pred, gt_bboxes, gt_labels)
    P, Q, _ = bbox_pred.shape
    Cn = cls_pred.shape[-1]
    bbox_pred = bbox_pred.detach().contiguous().float()
    cls_pred = cls_pred.detach().contiguous().float()
    gt_bboxes = gt_bboxes.contiguous().float()
    gt_labels = gt_labels.contiguous().long()
    total = int(gt_start[-1])
    cost = torch.empty(max(Q * total, 1), dtype=torch.float32, device=bbox_pred.device)
    wh = [v for pair in img_wh for v in pair]
    rc = load().dskd_match_cost(bbox_pred.data_ptr(), cls_pred.data_ptr(), gt_bboxes.data_ptr(),
                                gt_labels.data_ptr(), _host_i64(gt_start), _host_f32(wh), cost.data_ptr(),
                                P, Q, Cn, w_cls, w_reg, w_iou, _stream(bbox_pred))
    _check(rc, "dskd_match_cost")
    return cost


# --------------------------------------------------------------------------- dense detection losses (csrc/denseloss.hip)
class _DenseLossFunction(torch.autograd.Function):
    """(loss_cls, loss_bbox, loss_iou, loss_dfl), each [nl], of ``GFLDeformableDETRHead_il.loss_layers_dense`` in two
    launches; backward in one (``dskd_dense_loss_fwd`` / ``_bwd``)."""

    @staticmethod
    def forward(ctx, cls, box, lrtb, labels, tgt, pos, factors, avg_pos, weights):
        nl, N, Cn = cls.shape
        R1 = lrtb.shape[-1] // 4
        dev = cls.device
        f32 = torch.float32
        losses = torch.empty((4, nl), dtype=f32, device=dev)
        row_loss = torch.empty((4, nl * N), dtype=f32, device=dev)
        d_cls = torch.empty((nl * N, Cn), dtype=f32, device=dev)
        d_box = torch.empty((3, nl * N, 4), dtype=f32, device=dev)
        d_lrtb = torch.empty((nl * N, 4 * R1), dtype=f32, device=dev)
        rc = load().dskd_dense_loss_fwd(cls.data_ptr(), box.data_ptr(), lrtb.data_ptr(), labels.data_ptr(), tgt.data_ptr(),
                                        pos.data_ptr(), factors.data_ptr(), avg_pos.data_ptr(), losses.data_ptr(),
                                        row_loss.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(), d_lrtb.data_ptr(), nl, N, Cn, R1,
                                        *weights, _stream(cls))
        _check(rc, "dskd_dense_loss_fwd")
        ctx.save_for_backward(d_cls, d_box, d_lrtb, avg_pos)
        ctx.meta = (nl, N, Cn, R1, weights)
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        d_cls, d_box, d_lrtb, avg_pos = ctx.saved_tensors
        nl, N, Cn, R1, weights = ctx.meta
        g = torch.stack([t if t is not None else torch.zeros(nl, device=d_cls.device) for t in (g0, g1, g2, g3)]).float().contiguous()
        g_cls = torch.empty((nl, N, Cn), dtype=torch.float32, device=d_cls.device)
        g_box = torch.empty((nl, N, 4), dtype=torch.float32, device=d_cls.device)
        g_lrtb = torch.empty((nl, N, 4 * R1), dtype=torch.float32, device=d_cls.device)
        rc = load().dskd_dense_loss_bwd(g.data_ptr(), avg_pos.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(), d_lrtb.data_ptr(),
                                        g_cls.data_ptr(), g_box.data_ptr(), g_lrtb.data_ptr(), nl, N, Cn, R1, *weights,
                                        _stream(d_cls))
        _check(rc, "dskd_dense_loss_bwd")
        return g_cls, g_box, g_lrtb, None, None, None, None, None, None


def dense_losses_ok(cls, box, lrtb, beta, eps_iou, reg_max1) -> bool:
    """Can csrc/denseloss.hip take these dense losses (CUDA f32 tensors, QFL beta 2, GIoU eps 1e-6, <= 128 classes)?"""
    return (cls.is_cuda and cls.dtype == box.dtype == lrtb.dtype == torch.float32 and cls.dim() == 3 and float(beta) == 2.0
            and abs(float(eps_iou) - 1e-6) < 1e-12 and cls.shape[-1] <= 128 and 2 <= reg_max1 <= 64
            and lrtb.shape[-1] == 4 * reg_max1)


def dense_losses(cls, box, lrtb, labels, tgt, pos, factors, avg_pos, weights):
    """cls [nl, N, C], box [nl, N, 4] cxcywh, lrtb [nl, N, 4 * (reg_max + 1)], labels [nl, N] int64, tgt [nl, N, 4],
    pos [nl, N] bool, factors [N, 4], avg_pos 0-dim f32 tensor, weights = (w_cls, w_bbox, w_iou, w_dfl).
    Returns (loss_cls, loss_bbox, loss_iou, loss_dfl), each [nl]; differentiable w.r.t. cls, box, lrtb."""
    _need_gpu(cls, box, lrtb, labels, tgt, pos, factors, avg_pos)
    return _DenseLossFunction.apply(cls.contiguous(), box.contiguous(), lrtb.contiguous(), labels.contiguous().long(),
                                    tgt.contiguous().float(), pos.contiguous(), factors.contiguous().float(),
                                    avg_pos.detach().reshape(1).float().contiguous(), tuple(float(w) for w in weights))


# --------------------------------------------------------------------------- DSKD loss 1
class _ScaledGrad(torch.autograd.Function):
    """loss (scalar, already computed) whose gradient w.r.t. ``x`` is the saved dense grad."""

    @staticmethod
    def forward(ctx, x, loss, grad):
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def proto_corr_loss(hs_s: torch.Tensor, labels_s: torch.Tensor, prev_mask: torch.Tensor, hs_t: torch.Tensor,
                    keepid_t: torch.Tensor, labels_t: torch.Tensor, L: int, loss_weight: float = 1.0):
    """DSKD loss 1 (gfl_deformable_detr_head_il.py:525-555, :1197-1222). hs_s [N,D] requires
    grad; returns a scalar tensor whose backward reaches hs_s."""
    f = _dispatch_cpu("proto_corr_loss", hs_s)
    if f is not None:
        return f(hs_s, labels_s, prev_mask, hs_t, keepid_t, labels_t, L, loss_weight)
    _need_gpu(labels_s, prev_mask, hs_t, keepid_t, labels_t)
    N, D = hs_s.shape
    Cn = prev_mask.numel()
    M = keepid_t.numel()
    x = hs_s.detach().contiguous().float()
    ht = hs_t.detach().contiguous().float()
    lib = load()
    ws = torch.empty(int(lib.dskd_proto_corr_workspace(L, D)), dtype=torch.uint8, device=x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x)
    ls_, pm_ = labels_s.contiguous().long(), prev_mask.contiguous().to(torch.uint8)
    kt_, lt_ = keepid_t.contiguous().long(), labels_t.contiguous().long()
    rc = lib.dskd_proto_corr_fwd(x.data_ptr(), ls_.data_ptr(), pm_.data_ptr(), ht.data_ptr(),
                                 kt_.data_ptr(), lt_.data_ptr(),
                                 N, D, Cn, M, L, loss_weight, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                                 _stream(x))
    _check(rc, "dskd_proto_corr_fwd")
    return _ScaledGrad.apply(hs_s, loss[0], grad.to(hs_s.dtype))


# --------------------------------------------------------------------------- DSKD loss 2
def fgkd_loss(feats_s: List[torch.Tensor], feats_t: List[torch.Tensor], boxes: List[torch.Tensor],
              img_shapes: Sequence[Tuple[int, int]], hs_t: torch.Tensor, keepid_t: torch.Tensor,
              hs_s: torch.Tensor, labels_s: torch.Tensor, prev_mask: torch.Tensor, T: float = 2.0,
              loss_weight: float = 1.0, return_status: bool = False):
    """DSKD loss 2, ``decode_v1`` (gfl_deformable_detr_head_il.py:664-718).
    feats_* : per level [B,C,H,W]; boxes: per image [n_i,4] pixel xyxy (teacher order);
    img_shapes: per image (h, w) un-padded.  Returns a scalar whose backward reaches hs_s."""
    f = _dispatch_cpu("fgkd_loss", hs_s)
    if f is not None:
        return f(feats_s, feats_t, boxes, img_shapes, hs_t, keepid_t, hs_s, labels_s, prev_mask, T, loss_weight)
    _need_gpu(hs_t, keepid_t, labels_s, prev_mask, *feats_s, *feats_t)
    lib = load()
    levels = len(feats_s)
    B, Cc = feats_s[0].shape[:2]
    N, D = hs_s.shape
    fs = [nchw_f32(t) for t in feats_s]
    ft = [nchw_f32(t) for t in feats_t]
    shapes = []
    for t in fs:
        shapes += [t.shape[2], t.shape[3]]
    starts = [0]
    for bx in boxes:
        starts.append(starts[-1] + int(bx.shape[0]))
    M = starts[-1]
    allb = (torch.cat([b.reshape(-1, 4) for b in boxes], 0) if M > 0 else hs_s.new_zeros((0, 4))).contiguous().float()
    hw = [v for s in img_shapes for v in (float(s[0]), float(s[1]))]
    sh = _host_i32(shapes)
    ws = torch.empty(int(lib.dskd_fgkd_workspace(B, Cc, levels, sh, M, N)), dtype=torch.uint8, device=hs_s.device)
    x = hs_s.detach().contiguous().float()
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x)
    status = zeros(4, torch.int32, x.device)[:1]
    ps = (C.c_void_p * levels)(*[t.data_ptr() for t in fs])
    pt = (C.c_void_p * levels)(*[t.data_ptr() for t in ft])
    ht_, kt_ = hs_t.detach().contiguous().float(), keepid_t.contiguous().long()
    ls_, pm_ = labels_s.contiguous().long(), prev_mask.contiguous().to(torch.uint8)
    rc = lib.dskd_fgkd_fwd(ps, pt, sh, levels, B, Cc, allb.data_ptr(), _host_i32(starts), _host_f32(hw),
                           ht_.data_ptr(), kt_.data_ptr(),
                           x.data_ptr(), ls_.data_ptr(),
                           pm_.data_ptr(), N, D, prev_mask.numel(), M,
                           float(T), float(loss_weight), loss.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                           status.data_ptr(), _stream(x))
    _check(rc, "dskd_fgkd_fwd")
    out = _ScaledGrad.apply(hs_s, loss[0], grad.to(hs_s.dtype))
    return (out, status) if return_status else out
