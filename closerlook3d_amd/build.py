"""Build libcl3d.so (hand-written HIP for gfx950 behind the C ABI of include/cl3d.h).

hipcc cross-compiles without a GPU; the .so is written next to this file so it travels with the
source tree (it is git-ignored, not gpurun-ignored).
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
    "-Wall", "-Wextra", "-Wno-unused-parameter",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cl3d.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
