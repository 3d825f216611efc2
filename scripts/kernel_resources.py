"""Static resource usage of every kernel in libcl3d (VGPRs, AGPRs, SGPRs, scratch, static LDS, the occupancy the
register budget allows), from hipcc's -Rpass-analysis=kernel-resource-usage with the library's own flags.
Needs no GPU.      python scripts/kernel_resources.py > profiles/r01/kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from closerlook3d_amd import build  # noqa: E402

FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "LDS Size [bytes/block]",
          "Occupancy [waves/SIMD]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    flags = [f for f in build.FLAGS if f != "-shared"]
    print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'spill':>5s} {'LDS(static)':>11s} {'waves/SIMD':>10s}")
    for src in build.sources():
        with tempfile.TemporaryDirectory() as tmp:
            r = subprocess.run([build.HIPCC] + flags + ["-c", src, "-o", os.path.join(tmp, "x.o"),
                                "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:], file=sys.stderr)
            raise SystemExit(f"hipcc failed on {src}")
        recs, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+(.*?): (.*) \[-Rpass-analysis", line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2).strip()
            if k == "Function Name":
                cur = {"name": v}
                recs.append(cur)
            elif cur is not None:
                cur[k] = v
        names = demangle([x["name"] for x in recs])
        print(f"-- {os.path.basename(src)}")
        for x, n in zip(recs, names):
            n = re.sub(r"\(.*", "", n).replace("void ", "")
            if not n.startswith("cl3d::"):  # rocPRIM's sort / scan kernels instantiated by the library
                continue
            print(f"{n[:78]:78s} " + " ".join(f"{x.get(f, '?'):>{w}s}" for f, w in zip(FIELDS, (5, 5, 5, 7, 5, 11, 10))))


if __name__ == "__main__":
    main()
