"""Summarise a rocprofv3 rocpd SQLite result (`--kernel-trace --stats`, default output format) into the same
per-kernel CSV rocprofv3 prints with `--output-format csv`:  python tools/rocprof_db_to_csv.py in.db out.csv"""
import csv
import sqlite3
import statistics
import sys


def main(db: str, out: str) -> None:
    con = sqlite3.connect(db)
    rows = {}
    for name, dur in con.execute("select name, duration from kernels"):
        rows.setdefault(name, []).append(dur)
    total = sum(sum(v) for v in rows.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 4), min(v), max(v),
                        round(statistics.pstdev(v), 3)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
