"""Static instruction account of one kernel from hipcc's gfx950 assembly (--save-temps): per loop (as the compiler
labels them: `Loop Header: Depth=n`) the number of instructions by class -- VALU (packed FP32 counted apart: two issue
slots each on gfx950, profiles/r04/micro_pk_rate.txt), SALU, LDS reads / writes, vector-memory loads / stores, scalar
loads, MFMA, s_waitcnt, barriers, branches.  Round 5, VERDICT r4 item 10: what the PointWiseMLP's TRAIN pass
(pwmlp_query_kernel<0,4,4,4,1,true>) spends per slot, next to the PMC counters of the same kernel
(profiles/r05/train_pass_isa.txt).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics --save-temps -c csrc/fused_pwmlp.hip
    python scripts/isa_account.py fused_pwmlp-hip-amdgcn-amd-amdhsa-gfx950.s 'pwmlp_query_kernelILi0ELi4ELi4ELi4ELi1ELb1E'
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "lds_read"
    if op.startswith("ds_"):
        return "lds_write_or_atomic"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
        return "vmem_store"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem_load"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_barrier":
        return "s_barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # blocks: label -> (depth, header label) from the compiler's loop comments
    blocks, cur, cur_info = [], None, (0, None)
    label_re = re.compile(r"^(\.LBB\d+_\d+):")
    info = {}
    pending = None
    for i, l in enumerate(body):
        m = label_re.match(l)
        if m:
            cur = m.group(1)
            pending = cur
            info.setdefault(cur, {"depth": 0, "header": None, "counts": collections.Counter(), "line": i})
            # comments on the label line and the following comment lines carry the loop nest
            j = i
            text = l
            while j + 1 < len(body) and body[j + 1].lstrip().startswith(";"):
                j += 1
                text += " " + body[j]
            mh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", text)
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", text)
            if mh:
                info[cur]["depth"], info[cur]["header"] = int(mh.group(1)), cur
            elif mi:
                info[cur]["depth"], info[cur]["header"] = int(mi.group(2)), ".L" + mi.group(1)
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or cur is None:
            continue
        op = s.split()[0]
        info[cur]["counts"][classify(op)] += 1
    total = collections.Counter()
    per_loop = collections.defaultdict(collections.Counter)
    depth_of = {}
    for lab, d in info.items():
        total.update(d["counts"])
        if d["header"] is not None:
            per_loop[d["header"]].update(d["counts"])
            depth_of[d["header"]] = max(depth_of.get(d["header"], 0), d["depth"])
    keys = ["valu", "valu_packed", "salu", "lds_read", "lds_write_or_atomic", "vmem_load", "vmem_store", "smem_load", "mfma",
            "s_waitcnt", "s_barrier", "branch", "s_nop", "other"]
    print("kernel:", body[0].split(":")[0])
    print("static instruction counts (whole kernel): " + ", ".join(f"{k} {total[k]}" for k in keys if total[k]))
    print("per loop (blocks whose innermost enclosing loop header is the label; straight-line count, not trip-weighted):")
    for lab in sorted(per_loop, key=lambda x: info[x]["line"]):
        c = per_loop[lab]
        issue = c["valu"] + 2 * c["valu_packed"] + c["salu"] + c["lds_read"] + c["lds_write_or_atomic"] + c["vmem_load"] + \
            c["vmem_store"] + c["smem_load"] + c["s_waitcnt"] + c["branch"] + c["s_nop"] + c["mfma"]
        print(f"  {lab:12s} depth {depth_of[lab]}  issue slots {issue:5d}  " + ", ".join(f"{k} {c[k]}" for k in keys if c[k]))


if __name__ == "__main__":
    main()
