"""Fit of the bf16 launch-time model of csrc/mfma_gemm.hip (bf16_launch_us / slice_sum_us) to the sweeps of
scripts/micro/gemm_plan_sweep.py, and what the plans it picks cost against the best plan of every product.

    python scripts/micro/fit_gemm_plan.py profiles/r06/gemm_plan_sweep_bf16_xcd.jsonl [more sweep files]

Runs on the CPU (numpy + scipy).  Prints the fitted constants in the order they appear in the source, the rms of the
log residual, and three totals over the products: plans chosen by the CURRENT constants (the ones in this file, copied
from the source), plans chosen by the new fit, the best plan of each product.
"""
import json
import sys

import numpy as np
from scipy.optimize import least_squares

B = 16
KCUS = 256
RESIDENT = [4, 3, 3, 2]  # 64x64, 128x64, 64x128, 128x128
KC = 64
CURRENT = dict(T0RT=4.3 + 3.178, chunk=[1.171, 1.682, 1.644, 0.65], tail=[5.662, 4.682, 3.924, 5.453], cu=35.636,
               sum_a=1.546, sum_b=6.851, gain=1.0)


def dims(layer, product):
    C, n, Co = layer
    P = B * n
    return {"fwd": (Co, P, C), "bwd_data": (C, P, Co), "bwd_weight": (Co, C, P),
            "pt_fwd": (P, 2 * Co, C), "pt_bwd_data": (C, P, 2 * Co), "pt_bwd_weight": (2 * Co, C, P)}[product]


def ceil_div(a, b):
    return -(-a // b)


def real_plan(K, split):
    chunks = ceil_div(K, KC)
    split = min(split, chunks)
    cps = ceil_div(chunks, split)
    return ceil_div(chunks, cps), cps


def model(p, I, J, K, wi, wj, split, always_reduce):
    t = (1 if wi == 2 else 0) + (2 if wj == 2 else 0)
    real, cps = real_plan(K, split)
    ti, tj = ceil_div(I, 64 * wi), ceil_div(J, 64 * wj)
    wgs = ti * tj * real
    conc = min(max(wgs / KCUS, 1.0), RESIDENT[t])
    chunk_kb = (64.0 * wi + 64.0 * wj) * 64.0 * 4.0 / 1e3
    t_chunk = max(conc * chunk_kb / p["cu"], p["chunk"][t])
    gens = max(wgs / (KCUS * RESIDENT[t]), 1.0)
    us = p["T0RT"] + gens * ((cps - 1) * t_chunk + p["tail"][t])
    if real > 1 or always_reduce:
        two = p["sum_a"] + real * I * J * 4.0 * 2.0 / 1e6 / p["sum_b"]
        in_launch = (not always_reduce) and real <= 3 and wi == 1 and wj == 1
        us += two - p["gain"] if in_launch else two
    return us


def unpack(x):
    return dict(T0RT=x[0], chunk=list(x[1:5]), tail=list(x[5:9]), cu=x[9], sum_a=x[10], sum_b=x[11], gain=CURRENT["gain"])


def pack(p):
    return np.array([p["T0RT"]] + p["chunk"] + p["tail"] + [p["cu"], p["sum_a"], p["sum_b"]])


def splits_of(K, max_split):
    chunks = ceil_div(K, KC)
    out, s = [], 1
    while s <= max_split and s <= (chunks // 2 if chunks >= 2 else 1):
        out.append(s)
        s += 1 if s < 4 else s // 2
    return out or [1]


def choose(p, I, J, K, wgrad, always_reduce):
    best, best_cost = None, 1e300
    for wi, wj in ((2, 2), (2, 1), (1, 2), (1, 1)):
        for s in splits_of(K, 512 if wgrad else 16):
            c = model(p, I, J, K, wi, wj, s, always_reduce)
            if c < best_cost * 0.97:
                best_cost, best = c, (wi, wj, real_plan(K, s)[0])
    return best


def main(files):
    samples, tables = [], {}
    for f in files:
        for line in open(f):
            if not line.startswith("{"):
                continue
            r = json.loads(line)
            if r.get("prec") != "bf16" or "us_by_split" not in r:
                continue
            key = (tuple(r["layer"]), r["product"])
            I, J, K = dims(*key)
            wi, wj = r["tile"]
            for sp, us in r["us_by_split"].items():
                real = real_plan(K, int(sp))[0]
                samples.append((I, J, K, wi, wj, int(sp), r["product"] == "pt_bwd_weight", us))
                tables.setdefault(key, {})[(wi, wj, real)] = min(us, tables.get(key, {}).get((wi, wj, real), 1e9))

    def resid(x):
        p = unpack(x)
        return [np.log(model(p, I, J, K, wi, wj, sp, ar) / us) for I, J, K, wi, wj, sp, ar, us in samples if us < 200.0]

    x0 = pack(CURRENT)
    print("samples", len(samples), "rms(log) with the current constants %.3f" % np.sqrt(np.mean(np.square(resid(x0)))))
    fit = least_squares(resid, x0, bounds=(x0 * 0 + 1e-3, x0 * 0 + 100.0))
    p = unpack(fit.x)
    print("rms(log) after the fit %.3f" % np.sqrt(np.mean(np.square(fit.fun))))
    print("T0 + RT = %.3f" % p["T0RT"])
    print("kChunk  =", [round(v, 3) for v in p["chunk"]])
    print("kTail   =", [round(v, 3) for v in p["tail"]])
    print("cu_kb_per_us = %.3f   slice sum: %.3f + bytes / %.3f" % (p["cu"], p["sum_a"], p["sum_b"]))
    tot = {"current": 0.0, "fit": 0.0, "best": 0.0}
    for key, tab in tables.items():
        I, J, K = dims(*key)
        wgrad = key[1].endswith("bwd_weight")
        for name, par in (("current", CURRENT), ("fit", p)):
            plan = choose(par, I, J, K, wgrad, key[1] == "pt_bwd_weight")
            near = min(tab, key=lambda k: (k[0] != plan[0], k[1] != plan[1], abs(np.log(k[2] / plan[2]))))
            tot[name] += tab[near]
        tot["best"] += min(tab.values())
    print({k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main(sys.argv[1:])
