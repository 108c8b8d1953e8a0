"""Idle time between the kernels of one training iteration, from a rocprofv3
kernel trace (see tools/step_breakdown.py for how to record one): number of
kernels, sum of kernel durations, wall time from the first start to the last
end, and the time no kernel was running.

    python tools/step_gaps.py DIR [iterations-to-average]
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    iters, cur = [], []
    for i, r in enumerate(rows):
        cur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        nxt = rows[i + 1]["Kernel_Name"] if i + 1 < len(rows) else ""
        if "sgd_multi" in r["Kernel_Name"] and "sgd_multi" not in nxt and \
                "cast" not in nxt:
            iters.append(cur)
            cur = []
    for it in iters[-keep:]:
        busy_end = it[0][0]
        idle = 0
        hist = {}
        for s, e in it:
            if s > busy_end:
                gap = s - busy_end
                idle += gap
                b = min(gap // 1000, 9)
                hist[b] = hist.get(b, 0) + 1
            busy_end = max(busy_end, e)
        wall = busy_end - it[0][0]
        print("kernels %d  durations %.3f ms  wall %.3f ms  idle %.3f ms "
              "(%.1f %%)  gaps by microsecond: %s" % (
                  len(it), sum(e - s for s, e in it) / 1e6, wall / 1e6,
                  idle / 1e6, 100.0 * idle / wall,
                  " ".join("%d:%d" % kv for kv in sorted(hist.items()))))


if __name__ == "__main__":
    main()
