"""Per-layer mean kernel time of the weight-gradient trace runs."""
import csv, glob, sys
layers = ["64>64 1x1@56", "256>64 1x1@56", "64>256 1x1@56", "64>64 3x3@56",
          "512>128 1x1@28", "128>128 3x3@28", "1024>256 1x1@14",
          "256>256 3x3@14", "256>1024 1x1@14"]
reps = 10
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*kernel_trace.csv")[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    main = [r for r in rows if "wgrad_h16" in r["Kernel_Name"] or "conv_igemm" in r["Kernel_Name"]]
    red = [r for r in rows if "wgrad_reduce" in r["Kernel_Name"]]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = []
    for i, name in enumerate(layers):
        m = [dur(r) for r in main[i * reps:(i + 1) * reps]][2:]
        rr = [dur(r) for r in red[i * reps:(i + 1) * reps]][2:]
        g = main[i * reps]["Grid_Size_X"] if "Grid_Size_X" in main[0] else main[i * reps].get("Grid_Size", "0")
        out.append("%s: %.1f+%.1f (%d wg)" % (name, sum(m) / len(m), sum(rr) / max(1, len(rr)), int(g) // 256))
    print(d.split("/")[-1], " | ".join(out))
