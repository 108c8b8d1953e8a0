"""Per-kernel time of one training iteration from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py \
        --steps 5 --warmup 3 --no-graph
    python tools/step_breakdown.py DIR [iterations-to-average]

Iterations are cut at the last optimizer kernel of a step; the mean of the
last few iterations is printed as a markdown table (autotuning launches of the
first iterations are thereby left out).
"""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    iters, cur = [], []
    for i, r in enumerate(rows):
        cur.append(r)
        nxt = rows[i + 1]["Kernel_Name"] if i + 1 < len(rows) else ""
        if "sgd_multi" in r["Kernel_Name"] and "sgd_multi" not in nxt and \
                "cast" not in nxt:
            iters.append(cur)
            cur = []
    iters = iters[-keep:]
    tot = collections.Counter()
    cnt = collections.Counter()
    for it in iters:
        for r in it:
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            tot[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            cnt[name] += 1
    n = len(iters)
    total = sum(tot.values()) / n / 1e6
    print("kernel time %.3f ms per iteration (mean of the last %d)" % (total, n))
    print("| kernel | calls / iteration | ms / iteration | % |")
    print("|---|---|---|---|")
    for name, t in tot.most_common(40):
        ms = t / n / 1e6
        print("| `%s` | %.1f | %.3f | %.1f |" % (name, cnt[name] / n, ms,
                                                  100 * ms / total))


if __name__ == "__main__":
    main()
