"""Timeline of ONE replay of the bench step from a rocprofv3 kernel trace: per kernel its start offset, duration and
queue, plus the idle time on the critical path (gaps where no kernel of the step is running).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o tl -- python bench.py --steps 20 --warmup 5 ...
    python scripts/step_timeline.py out/**/tl_kernel_trace.csv [marker_kernel_substring]

The step is delimited by consecutive launches of the marker kernel (default: the first kernel of the fused step,
bq_tile_kernel or bq_prep_kernel); the last complete replay in the trace is printed.
"""
import csv
import glob
import sys


def main():
    paths = [p for a in sys.argv[1:2] for p in glob.glob(a, recursive=True)]
    marker = sys.argv[2] if len(sys.argv) > 2 else None
    rows = []
    for p in paths:
        for r in csv.DictReader(open(p)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    if marker is None:  # the ball query opens the step: the LDS-resident kernel, or the cell grid's prep kernel
        marker = "bq_tile_kernel" if any("bq_tile_kernel" in r[2] for r in rows) else "bq_prep_kernel"
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        print("marker kernel not found often enough")
        return
    # the last complete REPLAY of the step.  The bench launches the marker kernel again after the timed region -- alone
    # (the boundary timings) and inside an eager, event-timed run of the step (`roofline.step`) -- so: among the marker
    # intervals that hold a whole step's kernels, the last one whose period is within 20 % of the shortest (the replays)
    spans = [(rows[marks[k]][0] - rows[marks[k - 1]][0], k) for k in range(1, len(marks)) if marks[k] - marks[k - 1] >= 8]
    lo, hi = marks[-3], marks[-2]
    if spans:
        shortest = min(sp for sp, _ in spans)
        k = max(k for sp, k in spans if sp <= 1.2 * shortest)
        lo, hi = marks[k - 1], marks[k]
    step = rows[lo:hi]
    t0 = step[0][0]
    busy_until = t0
    idle = 0
    print("step of %d kernels, %.1f us from first start to next step's first start" % (len(step), (rows[hi][0] - t0) / 1e3))
    for s, e, name, q in step:
        gap = s - busy_until
        if gap > 0:
            idle += gap
        print("%8.1f +%7.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name.split("(")[0].replace("void ", "")[:70],
                                            "   <-- idle %.1f us before" % (gap / 1e3) if gap > 1500 else ""))
        busy_until = max(busy_until, e)
    print("idle (no kernel running) inside the step: %.1f us; last end at %.1f us" % (idle / 1e3, (busy_until - t0) / 1e3))


if __name__ == "__main__":
    main()
