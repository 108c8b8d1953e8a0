"""Per-kernel hardware-counter table from three rocprofv3 --pmc passes.

    python tools/pmc_table.py PASS_A PASS_B PASS_C [f16|f32]

PASS_A: SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_{F16,F32}
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
PASS_B: FETCH_SIZE TCC_HIT_sum        PASS_C: WRITE_SIZE TCC_MISS_sum
"""
import collections
import csv
import glob
import re
import sys


def load(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per, calls, dur


def main():
    a, ca, da = load(sys.argv[1])
    b, _, _ = load(sys.argv[2])
    c, _, _ = load(sys.argv[3])
    kind = sys.argv[4] if len(sys.argv) > 4 else "f16"
    mops = "SQ_INSTS_VALU_MFMA_MOPS_" + kind.upper()
    print("| kernel | calls | MFMA util % | MFMA TF/s | LDS conflict % | "
          "fetch MB/call | write MB/call | L2 hit % |")
    print("|---|---|---|---|---|---|---|---|")
    order = sorted(a, key=lambda k: -da[k])
    for k in order[:24]:
        n = len(ca[k])
        v = a[k]
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        util = 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / \
            (gui / 8 * 1024) if gui else 0.0
        tf = v.get(mops, 0.0) * 512 / da[k] / 1e3 if da[k] else 0.0
        idx = v.get("SQ_LDS_IDX_ACTIVE", 0.0)
        conf = 100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else 0.0
        fetch = b.get(k, {}).get("FETCH_SIZE", 0.0) / 1024 / max(n, 1)
        write = c.get(k, {}).get("WRITE_SIZE", 0.0) / 1024 / max(n, 1)
        hit = b.get(k, {}).get("TCC_HIT_sum", 0.0)
        miss = c.get(k, {}).get("TCC_MISS_sum", 0.0)
        l2 = 100.0 * hit / (hit + miss) if hit + miss else 0.0
        print("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.0f |" %
              (k, n, util, tf, conf, fetch, write, l2))


if __name__ == "__main__":
    main()
