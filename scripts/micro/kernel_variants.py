"""Build variants of one kernel file (its compile-time tunables: CL3D_TL_* of csrc/ball_query_lds.hip, CL3D_SUP_* of the
support-major pass in csrc/fused_pwmlp.hip) as whole libraries and time them: the ball query alone
(scripts/bench_bq.py) with its bit-exactness against the shipped library, and the replayed step with its per-entry table
(bench.py) -- a kernel shares the chip with its neighbours there, which decides more than its own time.

  python scripts/micro/kernel_variants.py --build            (build container: hipcc cross-compiles; the .so files travel)
  python scripts/micro/kernel_variants.py --run [--step]     (GPU box)
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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fno-slp-vectorize",
         "-DCL3D_D2_FORM=0"]

VARIANTS = {  # tag: (source file, extra defines)
    "sup_sb4": ("fused_pwmlp.hip", []),
    "sup_sb4_late": ("fused_pwmlp.hip", ["-DCL3D_SUP_LATE=1"]),
    "sup_sb6_late": ("fused_pwmlp.hip", ["-DCL3D_SUP_LATE=1", "-DCL3D_SUP_SB=6"]),
    "sup_sb3_late": ("fused_pwmlp.hip", ["-DCL3D_SUP_LATE=1", "-DCL3D_SUP_SB=3"]),
    "sup_sb2_late_w5": ("fused_pwmlp.hip", ["-DCL3D_SUP_LATE=1", "-DCL3D_SUP_SB=2", "-DCL3D_SUP_WAVES=5"]),
    # round 6: grid subsampling's in-LDS sort as the bitonic network of round 5 (the shipped build: radix sort on the cell bits)
    "sub_bitonic": ("grid_subsample.hip", ["-DCL3D_SUB_SORT=0"]),
    # round 6: BatchNorm statistics followed by a finalize launch of their own (shipped: the last-arriving workgroup of the
    # statistics launch finishes the channel)
    "bn_finalize_launch": ("bn_relu.hip", ["-DCL3D_BN_FOLD=0"]),
    # round 6: the GEMM's tiles in plain order (shipped: the tiles that share the large operand strip on one XCD)
    "gemm_plain_order": ("mfma_gemm.hip", ["-DCL3D_GEMM_XCD=0"]),
    # round 6: the PointWiseMLP passes with the lanes per row taken from CL3D_LANES (how a 72- / 144-channel row is cut)
    "pw_lanes_env": ("fused_pwmlp.hip", ["-DCL3D_LANE_ENV"]),
    # round 6: the lane maps of round 1-5 (most queries per wave; shipped: fewer, wider row pieces per wave-load)
    "lane_rule_r5": (("fused_pwmlp.hip", "fused_reduce.hip", "fused_maxpool.hip"), ["-DCL3D_LANE_RULE=0"]),
    # round 6, session 67: PseudoGrid's channel pairs as scalar FMAs (shipped: packed -- scalar measured 0.8-1 % slower)
    "pg_scalar": ("fused_reduce.hip", ["-DCL3D_PG_PK=0"]),
    # round 6, session 66: the ball query's two distance chains as one packed chain (rounds 4-6; shipped since: scalar)
    "bq_packed": ("ball_query_lds.hip", ["-DCL3D_TL_PK=1"]),
    # round 6, session 65: the TRAIN walk on packed pairs (rounds 4-6; shipped since: scalar FMAs)
    "train_packed": ("fused_pwmlp.hip", ["-DCL3D_TRAIN_PK=1"]),
    # (round 6, session 61: "walk_rz_pair" = rel.z of the TRAIN walk in a register pair of its own was the A/B arm that removed
    #  the wrong elements beside bf16 contractions; shipped as pk_low() in csrc/fused_pwmlp.hip)
    # (round 6: "pg_nofork" = fused_reduce.hip with -DCL3D_PG_FORK=0 was the A/B arm of PseudoGrid's forked kernel-weight
    #  pass; the fork measured slower and was removed with its macro -- profiles/r06/session7_summary.txt, commit dbb7bed)
}


def build(only=None):
    os.makedirs(VAR, exist_ok=True)
    objs = [o for o in sorted(glob.glob(os.path.join(CSRC, "*.o"))) if ".d2form" not in o]
    procs = []
    for tag, (src, defs) in VARIANTS.items():
        if only and tag not in only:
            continue
        srcs = [src] if isinstance(src, str) else list(src)  # (a variant may rebuild several files with its defines)
        for one in srcs:
            obj = os.path.join(VAR, f"var_{tag}_{one[:-4]}.o")
            procs.append((tag, one, obj, subprocess.Popen([HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, one), "-o", obj],
                                                          stderr=subprocess.DEVNULL)))
    by_tag = {}
    for tag, one, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed on variant {tag} ({one})")
        by_tag.setdefault(tag, []).append((one, obj))
    for tag, pairs in by_tag.items():
        lib = os.path.join(VAR, f"libcl3d_{tag}.so")
        replaced = {one[:-4] + ".o" for one, _ in pairs}
        others = [o for o in objs if os.path.basename(o) not in replaced]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + others + [obj for _, obj in pairs])
        for _, obj in pairs:
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
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"],
                               env=env, capture_output=True, text=True, timeout=600)
            if r.returncode == 0:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                row["step_ms"] = d["ms_per_step"]
                for k in d["roofline"]["step"]["kernels"]:
                    if k["entry"] in ("cl3d_pwmlp_bwd_support", "cl3d_masked_ordered_ball_query", "cl3d_pwmlp_stats"):
                        row[k["entry"].replace("cl3d_", "") + "_us"] = k["us"]
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
