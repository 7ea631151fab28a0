"""CPU-only: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/lx.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from loongx_amd import _lib
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/lx.h but not exported by liblx_amd.so"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert _lib.lib.lx_version() == 403


def test_struct_layouts_match_header():
    """ctypes mirrors of lx_gemm_desc / lx_attn_desc must have the C sizes (checked against a host compile)."""
    import subprocess, tempfile
    from loongx_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        open(src, "w").write('#include "lx.h"\n#include <stdio.h>\nint main(){printf("%zu %zu\\n", sizeof(lx_gemm_desc), sizeof(lx_attn_desc));return 0;}\n')
        exe = os.path.join(d, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert ctypes.sizeof(_lib.GemmDesc) == a and ctypes.sizeof(_lib.AttnDesc) == b


def test_argument_validation_without_gpu():
    """Entry points validate before launching: errors surface as status + message, no GPU needed."""
    from loongx_amd import _lib
    d = _lib.GemmDesc()
    assert _lib.lib.lx_gemm_bf16(ctypes.byref(d), 0, None) == -1
    assert b"out of range" in _lib.lib.lx_last_error()
    assert _lib.lib.lx_s4_scan(None, None, None, None, None, 1, 1, 100, 1, None) == -1


def test_qkv_epilogue_and_attention_argument_validation_without_gpu():
    """LX_EPI_QKV / n_qseg are validated on the host before anything is launched (fake, aligned addresses: nothing dereferences them)."""
    from loongx_amd import _lib
    d = _lib.GemmDesc()
    d.A = d.W = d.C = d.bias = 0x10000
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = 96, 768, 64, 64, 64, 768
    d.rows_per_batch = 48                                  # not a multiple of 32
    d.epilogue = _lib.LX_EPI_STORE_BF16 | _lib.LX_EPI_QKV
    d.qkv_norm_q = d.qkv_norm_k = d.qkv_rope = d.qkv_vt = 0x20000
    d.qkv_d, d.qkv_vt_ld, d.qkv_vt_pos0 = 256, 64, 0
    assert _lib.lib.lx_gemm_bf16(ctypes.byref(d), 1, None) == -1 and b"rows_per_batch" in _lib.lib.lx_last_error()
    d.rows_per_batch = 96
    d.qkv_d = 128                                          # heads must come in pairs (256-column tiles of one kind)
    assert _lib.lib.lx_gemm_bf16(ctypes.byref(d), 1, None) == -1 and b"qkv_d" in _lib.lib.lx_last_error()
    d.qkv_d, d.qkv_rope = 256, 0                           # the table is mandatory
    assert _lib.lib.lx_gemm_bf16(ctypes.byref(d), 1, None) == -1 and b"qkv_rope" in _lib.lib.lx_last_error()
    a = _lib.AttnDesc()
    a.Q = a.K = a.VT = a.O = 0x10000
    a.ldq = a.ldk = a.ldo = 768
    a.vt_ld, a.B, a.H, a.n_seg = 64, 1, 2, 2
    a.seg_len[0] = a.seg_len[1] = 32
    a.n_qseg = 3                                           # more query segments than segments
    assert _lib.lib.lx_attn_fwd(ctypes.byref(a), None) == -1 and b"n_qseg" in _lib.lib.lx_last_error()
