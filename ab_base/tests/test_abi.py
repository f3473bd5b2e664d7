"""The C-ABI library loads (no GPU needed: hipcc cross-compiles, libamdhip64 is in the image)
and exports every symbol include/dskd_hip.h declares; host-only entry points work."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from dskd_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dskd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dskd_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(native.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(native.lib_path())
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    assert native.load().dskd_abi_version() == 2


def test_cpu_tensors_fail_loudly_without_checker():
    """The product path has no CPU fallback."""
    native.install_cpu_checker(None)
    v = torch.randn(1, 6, 8, 32)
    loc = torch.rand(1, 2, 8, 1, 4, 2)
    attn = torch.rand(1, 2, 8, 1, 4)
    with pytest.raises(native.NativeError, match="no CPU fallback"):
        native.ms_deform_attn(v, [(2, 3)], loc, attn)
    with pytest.raises(native.NativeError):
        native.msda_forward_raw(v, [(2, 3)], loc, attn)
    with pytest.raises(native.NativeError):
        native.proto_corr_loss(torch.randn(4, 8), torch.zeros(4, dtype=torch.long), torch.ones(3, dtype=torch.bool),
                               torch.randn(4, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 2)


def test_argument_validation_returns_error_codes():
    lib = native.load()
    # unsupported head geometry -> DSKD_ERR_INVALID_ARG with a message, no launch attempted
    ss = (ctypes.c_int64 * 2)(2, 3)
    ls = (ctypes.c_int64 * 1)(0)
    rc = lib.dskd_msda_fwd(1, ss, ls, 1, 1, 1, 1, 6, 2, 4, 64, 1, 4, 0, None)
    assert rc == -1 and b"heads=8" in lib.dskd_last_error()
    rc = lib.dskd_lsap_batched(None, None, None, None, -1, None, None, None, None, None)
    assert rc == -1
    # the fused FFN is built for d_model 256 / hidden 1024 only; other sizes are refused before any launch
    assert lib.dskd_ffn_packed_bytes(256, 1024) == 32 * 32768
    assert lib.dskd_ffn_packed_bytes(256, 2048) == -1 and b"hidden 1024" in lib.dskd_last_error()
    assert lib.dskd_ffn_fwd(1, 1, 1, 1, None, 1, 4, 128, 1024, 0.0, 0, 0, None, 1, None) == -1
    assert lib.dskd_ffn_bwd(16, 16, 16, 16, 16, None, None, 1, 4, 256, 1024, 1.5, 1, None) == -1 and b"p=" in lib.dskd_last_error()


def test_gemm_tile_hook_and_scratch_size():
    """dskd_gemm_nt_tune refuses unknown configurations; the scratch bound covers the largest split (512 partial tiles of
    128 x 128 f32 or 256 of 256 x 128)."""
    lib = native.load()
    assert lib.dskd_gemm_nt_tune(10, 0) == -1 and b"cfg" in lib.dskd_last_error()
    for cfg in (-1, 0, 1, 6, 7, 8, -1):
        assert lib.dskd_gemm_nt_tune(cfg, 0) == 0
    assert lib.dskd_gemm_nt_scratch_bytes() >= 512 * 128 * 128 * 4
    # argument validation happens before any launch: misaligned scratch, negative size, bad N
    assert lib.dskd_gemm_nt_ws(16, 16, None, None, None, 16, 4, 64, 64, 0, 0, 0, 0, 0, 0, 1, 8, 64, None) == -1
    assert lib.dskd_gemm_nt_ws(16, 16, None, None, None, 16, 4, 64, 64, 0, 0, 0, 0, 0, 0, 1, 16, -1, None) == -1
    assert lib.dskd_gemm_nt_ws(16, 16, None, None, None, 16, 4, 96, 64, 0, 0, 0, 0, 0, 0, 1, None, 0, None) == -1
    assert lib.dskd_conv3x3_ws(16, 16, None, None, None, 16, 1, 4, 4, 48, 64, 1, 0, 1, None, 0, None) == -1


def test_lsap_host_entry_matches_scipy():
    from scipy.optimize import linear_sum_assignment as sp
    rng = np.random.default_rng(3)
    for shape in [(300, 17), (5, 9), (9, 5), (1, 1), (40, 40)]:
        for integer in (False, True):
            c = (rng.integers(0, 4, shape) if integer else rng.random(shape)).astype(np.float32)
            r, cc = native.lsap_host(torch.from_numpy(c))
            a = sp(c)
            assert np.array_equal(r.numpy(), a[0]) and np.array_equal(cc.numpy(), a[1])
    with pytest.raises(ValueError, match="invalid numeric"):
        native.lsap_host(torch.tensor([[1.0, float("nan")]]))
    with pytest.raises(ValueError, match="infeasible"):
        native.lsap_host(torch.tensor([[float("inf"), float("inf")], [1.0, 2.0]]))


def test_gemm_nt_argument_validation():
    """dskd_gemm_nt refuses what the kernel's tiling cannot take before any launch (no GPU needed)."""
    lib = native.load()
    ok_ptr = 4096
    assert lib.dskd_gemm_nt(ok_ptr, ok_ptr, None, None, ok_ptr, 128, 96, 64, 0, 0, 0, 0, 0, 0, 1, None) == -1
    assert b"multiples of 64" in lib.dskd_last_error()
    assert lib.dskd_gemm_nt(ok_ptr, ok_ptr, None, None, ok_ptr, 128, 64, 100, 0, 0, 0, 0, 0, 0, 1, None) == -1
    assert lib.dskd_gemm_nt(ok_ptr + 2, ok_ptr, None, None, ok_ptr, 128, 64, 64, 0, 0, 0, 0, 0, 0, 1, None) == -1
    assert b"aligned" in lib.dskd_last_error()
    assert lib.dskd_gemm_nt(ok_ptr, ok_ptr, None, None, ok_ptr, 128, 64, 64, 0, 0, 0, 0, 0, 0, 0, None) == -1      # f32: refused
    assert lib.dskd_gemm_nt(ok_ptr, ok_ptr, None, None, ok_ptr, 100, 64, 64, 0, 2, 5, 5, 9, 8, 1, None) == -1      # 2 * 4 >= Wi
    assert b"row map" in lib.dskd_last_error()
    assert lib.dskd_gemm_nt(ok_ptr, ok_ptr, None, None, ok_ptr, 0, 64, 64, 0, 0, 0, 0, 0, 0, 1, None) == 0         # nothing to do


def test_round3_late_entry_points_validate_their_arguments():
    """dskd_gemm_nt_dx / dskd_conv3x3_dx / dskd_gemm_tn_bf16 / dskd_colsum_short / dskd_sum_clear / dskd_fgkd_fwd refuse
    bad shapes, null and misaligned pointers and short scratch buffers BEFORE any launch (no GPU needed): a kernel that
    runs on operands it was not built for can fault the whole node."""
    lib = native.load()
    p = 4096
    bf16, f32 = native.DTYPE_BF16, native.DTYPE_F32
    assert lib.dskd_gemm_nt_dx(p, p, None, None, p, 128, 96, 64, bf16, None) == -1 and b"multiples of 64" in lib.dskd_last_error()
    assert lib.dskd_gemm_nt_dx(p, p, None, p + 8, p, 128, 64, 64, bf16, None) == -1 and b"aligned" in lib.dskd_last_error()
    assert lib.dskd_gemm_nt_dx(p, p, None, None, p, 128, 64, 64, f32, None) == -1
    assert lib.dskd_gemm_nt_dx(p, p, None, None, p, 0, 64, 64, bf16, None) == 0
    assert lib.dskd_conv3x3_dx(p, p, None, p, 1, 8, 8, 96, 64, bf16, None) == -1 and b"64 * 2^k" in lib.dskd_last_error()
    assert lib.dskd_conv3x3_dx(p, p, p + 2, p, 1, 8, 8, 64, 64, bf16, None) == -1 and b"aligned" in lib.dskd_last_error()
    assert lib.dskd_conv3x3_dx(p, p, None, p, 0, 8, 8, 64, 64, bf16, None) == 0
    # split-K scratch: the planner's answer, and a buffer one byte short of it
    need = lib.dskd_gemm_tn_scratch_bytes(88892, 256, 256)
    assert need > 0 and need % ((256 * 256 + 256) * 4) == 0 and need <= 64 << 20      # product planes + bias-gradient planes
    assert lib.dskd_gemm_tn_scratch_bytes(1000, 100, 256) == -1
    assert lib.dskd_gemm_tn_bf16(p, p, p, p, need - 1, 88892, 256, 256, 256, 256, bf16, None) == -1
    assert b"scratch" in lib.dskd_last_error()
    assert lib.dskd_gemm_tn_bf16(p, p, None, p, need, 88892, 256, 256, 256, 256, bf16, None) == -1
    assert lib.dskd_gemm_tn_bf16(p, p, p, p, need, 88892, 256, 256, 128, 256, bf16, None) == -1          # ldg < N
    assert lib.dskd_gemm_tn_bias_bf16(p, p, p, None, p, need, 88892, 256, 256, 256, 256, bf16, None) == -1    # no db output
    assert lib.dskd_gemm_tn_bias_bf16(p, p, p, p, p, need - 1, 88892, 256, 256, 256, 256, bf16, None) == -1
    assert lib.dskd_colsum_short(p, p, 70000, 256, bf16, None) == -1 and b"rows" in lib.dskd_last_error()
    assert lib.dskd_colsum_short(p, p, 100, 250, bf16, None) == -1
    assert lib.dskd_colsum_short(p + 2, p, 100, 256, bf16, None) == -1
    assert lib.dskd_sum_clear(None, 1, 1, 256, p, f32, None) == -1
    assert lib.dskd_sum_clear(p, 1, 0, 256, p, f32, None) == -1
    assert lib.dskd_sum_clear(p, 1, 1, 256, p, 7, None) == -1 and b"out_dtype" in lib.dskd_last_error()


def test_every_environment_switch_is_listed_and_unknown_ones_are_reported(monkeypatch):
    """an A/B script that sets a switch which no longer exists must not time the same code twice in silence.
    native.KNOWN_ENV lists every DSKD_* variable read anywhere (python: os.environ lookups; library: getenv), and
    native.unknown_env() -- which load() turns into a RuntimeWarning -- names the rest."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for f in glob.glob(root + "/dskd_amd/*.py") + glob.glob(root + "/tools/*.py") + [root + "/bench.py", root + "/__graft_entry__.py"]:
        read |= set(re.findall(r"environ(?:\.get|\.setdefault)?\(?\[?\s*[\"'](DSKD_[A-Z0-9_]+)[\"']", open(f).read()))
    for f in glob.glob(root + "/dskd_amd/csrc/*.hip") + glob.glob(root + "/dskd_amd/csrc/*.cpp") + glob.glob(root + "/dskd_amd/csrc/*.h"):
        read |= set(re.findall(r"getenv\(\"(DSKD_[A-Z0-9_]+)\"\)", open(f).read()))
    assert read and read <= native.KNOWN_ENV, sorted(read - native.KNOWN_ENV)
    monkeypatch.setenv("DSKD_MSDA_BWD", "r2")            # a switch round 3 removed
    monkeypatch.setenv("DSKD_NO_GRAPHS", "1")
    assert native.unknown_env() == ["DSKD_MSDA_BWD"]
