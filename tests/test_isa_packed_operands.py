"""The built library's packed-FP32 instructions, read back from its gfx950 code objects (no GPU needed).

Round 6 (DESIGN 6 "What may run beside a bf16 contraction", profiles/r06/session37-62): `v_pk_fma_f32` with `op_sel` set on
an operand -- the LOW result lane reading the HIGH dword of a register pair, which is how the compiler broadcasts a scalar
that sits in an odd register -- returned 0.0 to that lane whenever a dense bf16 MFMA contraction ran on the same CUs (the
engine's or torch.matmul, same process or another one): the PointWiseMLP gather pass and AdaptiveWeight's forward pass came
back with a few hundred wrong elements per launch.  The fix is in the source (csrc/fused_pwmlp.hip pk_low: the broadcast
value in the low dword of a pair of its own) and in the build (-fno-slp-vectorize: no compiler-made packed FP32); this test
keeps both from regressing: no kernel of the library may carry such an operand, except PseudoGrid's, listed below, whose
packed operands come out of VALU arithmetic (not out of ds_read_b128) and which stayed bit-exact beside bf16 contractions in
every survey of the round (scripts/micro/two_stream_survey.py, forward and backward, up to 400 launches each).  The TRAIN walk
and the ball query's distance chains, packed in rounds 4-6, are scalar code since sessions 65 / 66 (as fast or faster).
"""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "closerlook3d_amd", "libcl3d.so")
OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"

# kernels that may keep an op_sel'd packed operand (substring of the mangled name): measured exact beside bf16 contractions
ALLOWED = (
    "fused_reduce_fwd_kernelILi3E",   # PseudoGrid forward (influences computed in registers)
    "fused_reduce_bwd_kernelILi3E",   # PseudoGrid backward
    "pg_dkw_kernel",                  # PseudoGrid kernel-weight gradient
)


def gfx950_code_objects(path):
    """The device code objects of every clang offload bundle inside a host shared library."""
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    while pos != -1:
        num = struct.unpack_from("<Q", data, pos + 24)[0]
        p = pos + 32
        for _ in range(num):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size > 0:
                yield data[pos + off:pos + off + size]
        pos = data.find(magic, pos + 1)


def swizzled_packed_ops(lib, tmp_path):
    """{kernel: [instruction text, ...]} for every v_pk_* instruction with a 1 in its op_sel list."""
    found, kernels = {}, 0
    for n, blob in enumerate(gfx950_code_objects(lib)):
        co = os.path.join(str(tmp_path), f"obj{n}.co")
        with open(co, "wb") as fh:
            fh.write(blob)
        asm = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]{16} <([^>]+)>:", line)
            if m:
                cur = m.group(1)
                kernels += 1
                continue
            if "v_pk_" in line and re.search(r"op_sel:\[[01,]*1", line):
                found.setdefault(cur, []).append(line.split("//")[0].strip())
    return found, kernels


@pytest.mark.skipif(not os.path.exists(LIB), reason="libcl3d.so not built")
@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_no_packed_operand_reads_the_high_dword_of_a_pair(tmp_path):
    found, kernels = swizzled_packed_ops(LIB, tmp_path)
    assert kernels > 100, f"only {kernels} kernels disassembled: the code objects were not found"
    offenders = {k: v for k, v in found.items() if not any(a in k for a in ALLOWED)}
    assert not offenders, ("packed FP32 instructions with op_sel on an operand outside the kernels measured exact beside bf16 "
                           "contractions (DESIGN 6): " + "; ".join(f"{k}: {len(v)} x, e.g. {v[0]}" for k, v in offenders.items()))
    # the gather passes that WERE wrong: none at all, not even allowed ones
    for k in found:
        assert "pwmlp_query_kernel" not in k and "fused_reduce_fwd_kernelILi2E" not in k, k
