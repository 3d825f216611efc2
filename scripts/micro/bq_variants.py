"""Build variants of the LDS-resident ball query (csrc/ball_query_lds.hip, its CL3D_TL_* tunables) as whole libraries and
time them: the kernel alone (scripts/bench_bq.py), its bit-exactness against the shipped library, and the replayed
step (bench.py) -- the kernel shares the chip with the per-point GEMM there, which decides more than its own time.

  python scripts/micro/bq_variants.py --build            (build container: hipcc cross-compiles; the .so files travel)
  python scripts/micro/bq_variants.py --run [--step]     (GPU box)
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "closerlook3d_amd", "csrc")
VAR = os.path.join(ROOT, "scripts", "micro", "var")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-DCL3D_D2_FORM=0"]

VARIANTS = {
    "rank1_qt4": [],
    "rank0_qt4": ["-DCL3D_TL_RANK=0"],
    "rank1_qt2": ["-DCL3D_TL_QT=2"],
    "rank0_qt2": ["-DCL3D_TL_RANK=0", "-DCL3D_TL_QT=2"],
    "rank1_qt4_flat3": ["-DCL3D_TL_FLAT=3"],
    "rank1_qt8": ["-DCL3D_TL_QT=8"],
    # timing experiments (results are wrong on purpose: a phase is cut)
    "rank1_qt4_phase1": ["-DCL3D_TL_PHASE=1"],
    "rank1_qt4_phase2": ["-DCL3D_TL_PHASE=2"],
}


def build(only=None):
    os.makedirs(VAR, exist_ok=True)
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "*.o"))) if ".d2form" not in o
              and os.path.basename(o) != "ball_query_lds.o"]
    procs = []
    for tag, defs in VARIANTS.items():
        if only and tag not in only:
            continue
        obj = os.path.join(VAR, f"bq_{tag}.o")
        procs.append((tag, obj, subprocess.Popen([HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, "ball_query_lds.hip"),
                                                                          "-o", obj], stderr=subprocess.DEVNULL)))
    for tag, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed on variant {tag}")
        lib = os.path.join(VAR, f"libcl3d_{tag}.so")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + others + [obj])
        os.remove(obj)
        print("built", lib)


def run(step, only=None):
    import numpy as np
    env0 = dict(os.environ)
    ref = None
    rows = []
    tags = ["shipped"] + [t for t in VARIANTS if os.path.exists(os.path.join(VAR, f"libcl3d_{t}.so"))]
    for tag in tags:
        if only and tag != "shipped" and tag not in only:
            continue
        env = dict(env0)
        if tag != "shipped":
            env["CL3D_LIB"] = os.path.join(VAR, f"libcl3d_{tag}.so")
        row = {"variant": tag}
        out = os.path.join("/tmp", f"bq_idx_{tag}.npy")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_bq.py"), "--dump", out], env=env,
                           capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            row["error"] = r.stderr[-300:]
            rows.append(row)
            print(json.dumps(row), flush=True)
            continue
        row["us_median"] = json.loads(r.stdout.strip().splitlines()[-1])["us_median"]
        got = np.load(out)
        if ref is None:
            ref = got
        row["bit_exact"] = bool(np.array_equal(ref, got))
        if step and "phase" not in tag:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-kernel-roofline"],
                               env=env, capture_output=True, text=True, timeout=600)
            if r.returncode == 0:
                row["step_ms"] = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
            else:
                row["step_error"] = r.stderr[-300:]
        rows.append(row)
        print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--step", action="store_true")
    ap.add_argument("--only", nargs="*")
    a = ap.parse_args()
    if a.build:
        build(a.only)
    if a.run:
        run(a.step, a.only)
