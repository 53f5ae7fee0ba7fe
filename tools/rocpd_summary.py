"""Summarise rocprofv3's rocpd (SQLite) output: per-kernel time statistics and PMC sums.

    python tools/rocpd_summary.py stats <results.db>        # CSV like `rocprofv3 --stats`
    python tools/rocpd_summary.py pmc   <results.db>        # counter sums per kernel
"""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for n, c, t, a, lo, hi in rows:
        print(f"\"{short(n)}\",{c},{t},{a:.1f},{lo},{hi},{100.0 * t / total:.2f}")


def pmc(path):
    db = sqlite3.connect(path)
    # a dispatch holds one row per counter (and per counter dimension): time comes from distinct dispatches
    times = {n: (k, d) for n, k, d in db.execute(
        "select name, count(*), sum(duration) from (select distinct name, dispatch_id, duration from pmc_events) "
        "group by name")}
    rows = db.execute("select name, counter_name, sum(counter_value) "
                      "from pmc_events group by name, counter_name order by name, counter_name").fetchall()
    last = None
    for n, c, v in rows:
        if n != last:
            last = n
            print(f"{short(n)}  dispatches {times[n][0]}  time_ms {times[n][1] / 1e6:.3f}")
        print(f"    {c:<28s} {v:.6g}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
