"""Aggregate rocprofv3 --pmc counter CSVs per kernel.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [more dirs...] > summary.md

Counters of all given runs are merged by kernel name (one --pmc pass can hold only a few counters per
hardware block, so a profile is several passes). Derived columns: MFMA busy % (share of the SIMD-cycles of the
kernel in which the matrix pipe was busy: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip,
SQ_BUSY_CYCLES over its 32 shader engines, so their plain ratio overstates the figure 32-fold), bf16 MFMA TFLOP/s (512 flops per MOP unit), LDS bank-conflict % (conflict cycles over LDS
active cycles), HBM read GB and GB/s (FETCH_SIZE is in KiB)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")  # before cutting at the parameter list's parenthesis
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(Cijk_\w{4}_\w{4}).*?(MT\d+x\d+x\d+)", name)
    if m:
        return f"hipBLASLt {m.group(1)} {m.group(2)}"
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)(I|P|v|K)", name)
    return (m.group(1) if m else name)[:48]


N_SIMD = 256 * 4  # MI355X: 256 CUs x 4 SIMDs, each with its own MFMA busy counter
N_SE = 32         # shader engines (4 per XCD), each with its own SQ busy counter


def main(dirs):
    ctr = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    dur = defaultdict(lambda: defaultdict(float))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    c = r["Counter_Name"]
                    ctr[k][c] += float(r["Counter_Value"])
                    disp = (f, r["Dispatch_Id"])
                    if disp not in calls[k][c]:
                        calls[k][c].add(disp)
                        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                            dur[k][c] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = []
    for k, c in ctr.items():
        any_c = next(iter(c))
        n = len(calls[k][any_c])
        t_ns = dur[k][any_c]
        g = lambda name: c.get(name)  # noqa: E731
        mfma_busy = 100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES") * N_SE / N_SIMD if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None else None
        t_m = dur[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        tfl = g("SQ_INSTS_VALU_MFMA_MOPS_BF16") * 512 / t_m / 1e3 if t_m and g("SQ_INSTS_VALU_MFMA_MOPS_BF16") is not None else None
        lds = 100.0 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None else None
        t_f = dur[k].get("FETCH_SIZE", 0.0)
        rd_gb = g("FETCH_SIZE") * 1024 / 1e9 if g("FETCH_SIZE") is not None else None
        rd_bw = rd_gb / (t_f / 1e9) if rd_gb is not None and t_f else None
        t_w = dur[k].get("WRITE_SIZE", 0.0)
        wr_gb = g("WRITE_SIZE") * 1024 / 1e9 if g("WRITE_SIZE") is not None else None
        wr_bw = wr_gb / (t_w / 1e9) if wr_gb is not None and t_w else None
        rows.append((t_ns, k, n, mfma_busy, tfl, lds, rd_gb, rd_bw, wr_gb, wr_bw))
    if os.environ.get("PMC_RAW"):
        names = sorted({c for k in ctr for c in ctr[k]})
        print("| kernel | calls | " + " | ".join(names) + " |")
        print("|---|---|" + "---|" * len(names))
        for k, c in sorted(ctr.items(), key=lambda kv: -max(dur[kv[0]].values() or [0])):
            n = max(len(v) for v in calls[k].values())
            print(f"| {k} | {n} | " + " | ".join(f"{c.get(x, 0) / max(1, n):.3g}" for x in names) + " |")
        print()
    rows.sort(reverse=True)
    f = lambda v, p=1: "-" if v is None else f"{v:.{p}f}"  # noqa: E731
    print("| kernel | calls | avg us (under pmc) | MFMA busy % | bf16 MFMA TFLOP/s | LDS bank conflict % | HBM read GB/call | read GB/s | write GB/call | write GB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for t_ns, k, n, mb, tfl, lds, rd, rdbw, wr, wrbw in rows[:40]:
        print(f"| {k} | {n} | {f(t_ns / max(1, n) / 1e3)} | {f(mb)} | {f(tfl, 0)} | {f(lds)} | "
              f"{f(rd / max(1, n) if rd is not None else None, 3)} | {f(rdbw, 0)} | "
              f"{f(wr / max(1, n) if wr is not None else None, 3)} | {f(wrbw, 0)} |")


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
