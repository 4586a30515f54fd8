"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd SQLite result (`--pmc ... --kernel-trace`):
    python tools/rocprof_pmc_summary.py in.db out.csv
Writes one row per (kernel, counter): calls, total, per-launch average, plus the kernel's average duration."""
import csv
import sqlite3
import sys


def main(db: str, out: str) -> None:
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('pmc_events')")]
    if not cols:
        raise SystemExit("no pmc_events view in " + db)
    name_col = "counter_name" if "counter_name" in cols else [c for c in cols if "name" in c][0]
    val_col = "counter_value" if "counter_value" in cols else [c for c in cols if "value" in c][0]
    key = "dispatch_id" if "dispatch_id" in cols else "event_id"
    kcols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    kkey = key if key in kcols else "dispatch_id"
    durs = {}
    for did, name, dur in con.execute(f"select {kkey}, name, duration from kernels"):
        durs[did] = (name, dur)
    agg = {}
    for did, cname, val in con.execute(f"select {key}, {name_col}, {val_col} from pmc_events"):
        if did not in durs:
            continue
        k = (durs[did][0], cname)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(val)
        a[2] += durs[did][1]
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Kernel", "Counter", "Launches", "Total", "PerLaunch", "AvgDurationNs"])
        for (kn, cn), (n, tot, dur) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            w.writerow([kn[:120], cn, n, tot, tot / n, dur / n])
    print("columns in pmc_events:", cols)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
