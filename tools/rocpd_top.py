"""Per-kernel summary of a rocprofv3 run stored as a rocpd SQLite database (the default output format).

    python tools/rocpd_top.py gpurun_out/prof/run_results.db [--top 30] > profiles/summary.csv

Prints CSV: kernel name (template arguments and tile-configuration suffixes shortened), calls, total ms, mean us,
share of the summed kernel time."""
import argparse
import re
import sqlite3


def short(name: str, width: int = 110) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"_SN_LDSB.*$", "", name)  # hipBLASLt: keep the macro-tile, drop the tuning suffix
    name = re.sub(r"at::native::", "", name)
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print("kernel,calls,total_ms,mean_us,percent")
    for name, calls, dur, avg, pct in rows[:a.top]:
        print(f'"{short(name)}",{calls},{dur / 1e3:.3f},{avg:.2f},{pct:.2f}')  # the view reports microseconds
    print(f'"(all {len(rows)} kernels)",{sum(r[1] for r in rows)},{total / 1e3:.3f},,100.00')


if __name__ == "__main__":
    main()
