"""CPU: the C-ABI library builds, loads, and exports every symbol include/cl3d.h declares; the Python
binding table covers the header; the product never imports the oracle; missing library fails loudly."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cl3d.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cl3d_[a-z0-9_]+)\s*\(", src)) - {"cl3d_stream_t"})


@pytest.fixture(scope="module")
def libpath():
    from closerlook3d_amd import build
    return build.build()


def test_header_symbols_exported(libpath):
    names = _declared()
    assert "cl3d_masked_ordered_ball_query" in names and len(names) >= 9
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath]).decode()
    exported = set(re.findall(r" T (cl3d_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in cl3d.h but not exported: {missing}"
    undeclared = sorted(exported - set(names))
    assert not undeclared, f"exported but not declared in cl3d.h: {undeclared}"


def test_library_loads_and_reports_version(libpath):
    import torch  # noqa: F401  (binds the HIP runtime first, as the product does)
    h = ctypes.CDLL(libpath)
    from closerlook3d_amd import _lib
    assert h.cl3d_abi_version() == _lib.ABI_VERSION == 5  # bumped with round 6's new entry points (include/cl3d.h)
    assert _lib.header_abi_version() == _lib.ABI_VERSION  # the package's constant is the header's
    h.cl3d_last_error_string.restype = ctypes.c_char_p
    assert isinstance(h.cl3d_last_error_string(), bytes)
    h.cl3d_workspace_bytes.restype = ctypes.c_size_t
    assert h.cl3d_workspace_bytes(1, 16, 4096, 4096, 32, 64) >= 0
    assert h.cl3d_d2_form() == 0  # the default library carries hipcc's contraction of the reference expression


def test_fused_capability_query(libpath):
    """cl3d_fused_supported answers without a GPU: every shipped (nsample, width) is covered, absurd nsample is not
    -- `impl='auto'` asks before it commits to the fused kernels."""
    import torch  # noqa: F401
    from closerlook3d_amd import _lib
    h = _lib.lib()
    for op in (7, 8, 9, 10, 13):
        for K in (16, 26, 32, 42):
            for C in (36, 64, 72, 144, 1152):
                assert h.cl3d_fused_supported(op, K, C) == 1, (op, K, C)
        assert h.cl3d_fused_supported(op, 5000, 64) == 0
    assert h.cl3d_fused_supported(1, 16, 64) == 0  # not an operator id


def test_invalid_arguments_return_codes_not_exit(libpath):
    """Bad sizes / null pointers come back as negative codes with a message (the reference exit(-1)s)."""
    import torch  # noqa: F401
    from closerlook3d_amd import _lib
    h = _lib.lib()
    rc = h.cl3d_group_points(None, None, 1, 4, 16, 8, 2, None, None)
    assert rc == -1 and b"null" in h.cl3d_last_error_string()
    rc = h.cl3d_masked_ordered_ball_query(None, None, None, None, 1, 8, 0, 0.1, 4, None, None, None, 0, None)
    assert rc == -1 and b"bad sizes" in h.cl3d_last_error_string()
    # empty problems are fine and touch nothing
    assert h.cl3d_group_points(None, None, 0, 4, 16, 8, 2, None, None) == 0


def test_python_binding_table_covers_header():
    from closerlook3d_amd import _lib
    names = set(_declared()) - {"cl3d_abi_version", "cl3d_last_error_string", "cl3d_workspace_bytes", "cl3d_d2_form",
                                "cl3d_pwmlp_pass", "cl3d_reduce_pass"}
    by_pointer = {"cl3d_pwmlp_train_forward", "cl3d_pwmlp_train_backward", "cl3d_reduce_train_forward",
                  "cl3d_reduce_train_backward"}  # (argument block by pointer: _lib._declare)
    assert names == set(_lib.SIGNATURES) | by_pointer, (names ^ (set(_lib.SIGNATURES) | by_pointer))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "closerlook3d_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "libcl3d_oracle" in text:
                    bad.append(os.path.join(dirpath, f))
    for f in ("drop_in/pt_utils.py", "drop_in/pt_custom_ops/_ext.py"):
        text = open(os.path.join(ROOT, f)).read()
        if re.search(r"\boracle\b", text):
            bad.append(f)
    # measurement helpers go through bench.py's cpu_baseline legs, never to the oracle themselves
    for sub in ("scripts", "examples"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".sh")):
                    text = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "libcl3d_oracle" in text:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, f"product files reference the oracle: {bad}"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from closerlook3d_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_PATH", str(tmp_path / "libcl3d.so"))
    with pytest.raises(ImportError, match="not built"):
        _lib.lib()


def test_cpu_tensors_rejected_like_the_reference():
    import torch
    from closerlook3d_amd import _ext
    x = torch.rand(1, 8, 3)
    m = torch.ones(1, 8, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.masked_ordered_ball_query(x, x, m, m, 0.1, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.group_points(torch.rand(1, 2, 8), torch.zeros(1, 2, 2, dtype=torch.int32))


@pytest.mark.parametrize("cname,mirror", [("cl3d_pwmlp_pass", "PwmlpPass"), ("cl3d_reduce_pass", "ReducePass")])
def test_pass_argument_block_matches_the_header(tmp_path, cname, mirror):
    """cl3d_pwmlp_pass / cl3d_reduce_pass (include/cl3d.h) against their ctypes mirrors (_lib.PwmlpPass / ReducePass): size
    and the offset of every field, as the C compiler lays the header's struct out."""
    from closerlook3d_amd import _lib
    cls = getattr(_lib, mirror)
    fields = [n for n, _ in cls._fields_]
    src = tmp_path / "offsets.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "cl3d.h"\nint main(void) {\n'
                   + f'  printf("%zu\\n", sizeof({cname}));\n'
                   + "".join(f'  printf("%zu\\n", offsetof({cname}, {n}));\n' for n in fields) + "  return 0;\n}\n")
    exe = tmp_path / "offsets"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert got[0] == ctypes.sizeof(cls)
    assert got[1:] == [getattr(cls, n).offset for n in fields]


def test_pass_calls_validate_their_argument_block_before_any_launch():
    """cl3d_pwmlp_train_forward / _backward (csrc/pass.hip): a null block, bad sizes and a block with null buffers are
    CL3D_E_INVALID with a message, not a fault -- checked without a GPU (nothing is launched before the checks pass)."""
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    assert lib.cl3d_pwmlp_train_forward(None, None) == -1
    assert b"null argument block" in lib.cl3d_last_error_string()
    assert lib.cl3d_pwmlp_train_backward(None, None) == -1
    p = _lib.PwmlpPass()
    assert lib.cl3d_pwmlp_train_forward(ctypes.byref(p), None) == -1  # all sizes zero
    assert b"bad sizes" in lib.cl3d_last_error_string()
    p.B, p.N, p.M, p.K, p.C, p.Co, p.radius = 2, 64, 64, 8, 16, 16, 0.1
    assert lib.cl3d_pwmlp_train_forward(ctypes.byref(p), None) == -1  # every buffer null
    assert b"null pointer" in lib.cl3d_last_error_string()
    assert lib.cl3d_pwmlp_train_backward(ctypes.byref(p), None) == -1
    assert b"null pointer" in lib.cl3d_last_error_string()
    # the same for the three gather-and-reduce operators' passes (round 6)
    assert lib.cl3d_reduce_train_forward(None, None) == -1 and b"null argument block" in lib.cl3d_last_error_string()
    assert lib.cl3d_reduce_train_backward(None, None) == -1
    r = _lib.ReducePass()
    assert lib.cl3d_reduce_train_forward(ctypes.byref(r), None) == -1 and b"bad sizes" in lib.cl3d_last_error_string()
    r.B, r.N, r.M, r.K, r.C, r.op, r.radius = 2, 64, 64, 8, 12, 0, 0.1
    assert lib.cl3d_reduce_train_forward(ctypes.byref(r), None) == -1 and b"null pointer" in lib.cl3d_last_error_string()
    assert lib.cl3d_reduce_train_backward(ctypes.byref(r), None) == -1 and b"null pointer" in lib.cl3d_last_error_string()
    r.op = 4
    assert lib.cl3d_reduce_train_forward(ctypes.byref(r), None) == -1 and b"bad sizes" in lib.cl3d_last_error_string()


def test_pass_arena_hands_out_aligned_disjoint_addresses():
    """pass_calls._Arena: the sub-buffers of a pass's one allocation are 256-byte aligned, disjoint, in order, and a
    zero-size request still gets an address of its own (CPU tensor: only the address arithmetic is exercised)."""
    import torch
    from closerlook3d_amd import _lib
    from closerlook3d_amd.pass_calls import _Arena
    p = _lib.PwmlpPass()
    a = _Arena(p)
    sizes = {"idx": 4 * 1000, "idx_mask": 1, "bq_ws": 0, "ght": 4 * 12345, "kstar": 777, "partial": 8 * 64 * 8}
    for k, v in sizes.items():
        a.add(k, v)
    buf = a.allocate(torch.device("cpu"))
    base, prev_end = buf.data_ptr(), buf.data_ptr()
    for k, v in sizes.items():
        addr = getattr(p, k)
        assert addr is not None and (addr - base) % 256 == 0 and addr >= prev_end
        prev_end = addr + max(v, 1)
    assert prev_end <= base + buf.numel()
