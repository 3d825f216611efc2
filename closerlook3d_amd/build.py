"""Build libcl3d.so (hand-written HIP for gfx950 behind the C ABI of include/cl3d.h).

hipcc cross-compiles without a GPU; the .so is written next to this file so it travels with the
source tree (it is git-ignored, not gpurun-ignored).

`CL3D_D2_FORM` (environment, or build(d2_form=...)): the operation order of the squared distance that decides
ball-query / nearest-query indices bit for bit (DESIGN.md "floating-point canon", csrc/cl3d_common.h:dist2,
mirrors the oracle's switch of the same name):
    0  fadd(fma(dy,dy, dx*dx), dz*dz)   what hipcc -O2 makes of the reference expression on gfx950 (default)
    1  no contraction: (dx*dx + dy*dy) + dz*dz
    2  full left-to-right fma chain: fma(dz,dz, fma(dy,dy, dx*dx))   (what nvcc normally emits)
Form 0 builds libcl3d.so; the others build libcl3d_d2form<N>.so, which `_lib` loads when the same variable is
set at import time -- for a maintainer comparing against indices produced by another compiler.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcl3d.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # every fused multiply-add in the engine is written as __builtin_fmaf; the compiler must not
    # add or remove any (bit-exact distances, see DESIGN.md "floating-point canon")
    "-ffp-contract=off",
    # native global_atomic_add_f32 for the large-N scatter fallback
    "-munsafe-fp-atomics",
    # no compiler-made packed FP32: the SLP vectoriser pairs scalar FMAs into v_pk_* instructions and, for an operand in
    # the high half of a register pair, sets op_sel on it -- the form that read zeros beside bf16 MFMA kernels (round 6,
    # DESIGN 6; csrc/fused_pwmlp.hip pk_low).  The packed code that is left is written out by hand and checked by
    # tests/test_isa_packed_operands.py
    "-fno-slp-vectorize",
    "-Wall", "-Wextra", "-Wno-unused-parameter",
]


def d2_form_from_env():
    form = int(os.environ.get("CL3D_D2_FORM", "0") or 0)
    if form not in (0, 1, 2):
        raise ValueError(f"CL3D_D2_FORM must be 0, 1 or 2, got {form}")
    return form


def lib_path(d2_form=0):
    return LIB if d2_form == 0 else os.path.join(HERE, f"libcl3d_d2form{d2_form}.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _common_deps():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cl3d.h"), __file__]


def _obj(src, d2_form):
    return src[:-4] + (".o" if d2_form == 0 else f".d2form{d2_form}.o")


def needs_build(d2_form=0):
    lib = lib_path(d2_form)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in sources() + _common_deps())


def build(force=False, verbose=False, d2_form=None):
    if d2_form is None:
        d2_form = d2_form_from_env()
    lib = lib_path(d2_form)
    if not force and not needs_build(d2_form):
        return lib
    hdr_time = max(os.path.getmtime(d) for d in _common_deps())
    objs = []
    procs = []
    for src in sources():
        obj = _obj(src, d2_form)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(src)):
            continue  # object is newer than its source and every header
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + [f"-DCL3D_D2_FORM={d2_form}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
