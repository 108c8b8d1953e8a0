"""From a rocprofv3 kernel trace (CSV) of a DistOpt training run: do the
communication kernels (RCCL collectives, bucket pack / unpack) of a replayed
step run while compute kernels of the backward pass are executing?

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py ...
    python tools/overlap_timeline.py OUT > profiles/overlap_timeline.md

For every communication kernel of the last complete step the table lists its
interval and the compute kernels whose intervals intersect it (different
queue, overlapping time).  Inside a replayed hipGraph a forked branch is only
concurrent if the runtime maps it to another hardware queue, which is what
this checks."""
import csv
import glob
import re
import sys

COMM = re.compile(r"nccl|rccl|xgmi|pack_kernel|unpack_kernel|sparse|scatter_add",
                  re.I)


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""),
                     r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    comm = [r for r in rows if COMM.search(r[2])]
    if not comm:
        print("no communication kernel in the trace")
        return 1
    # the last step: communication kernels after the last long gap
    per_step = len([r for r in comm if r[0] >= comm[-1][0] - 50_000_000])
    last = comm[-max(1, min(per_step, 64)):]
    t0 = rows[0][0]
    print("# Communication kernels of the last replayed step against the compute "
          "kernels running at the same time")
    print()
    print("| communication kernel | queue | start us | duration us | compute "
          "kernels in flight on other queues (overlap us) |")
    print("|---|---|---|---|---|")
    total = overlapped = 0.0
    for s, e, name, q in last:
        hits = []
        cover = 0
        for s2, e2, n2, q2 in rows:
            if e2 <= s or s2 >= e or q2 == q or COMM.search(n2):
                continue
            ov = min(e, e2) - max(s, s2)
            hits.append("%s (%.1f)" % (n2.replace("sa::", "")[:48], ov / 1e3))
            cover += ov
        total += e - s
        overlapped += min(cover, e - s)
        print("| `%s` | %s | %.1f | %.1f | %s |" % (
            name[:60], q, (s - t0) / 1e3, (e - s) / 1e3,
            "; ".join(hits[:4]) or "none"))
    print()
    print("communication kernel time in the step: %.1f us, of which %.1f us "
          "(%.0f %%) with a compute kernel in flight on another queue" % (
              total / 1e3, overlapped / 1e3,
              100.0 * overlapped / total if total else 0.0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
