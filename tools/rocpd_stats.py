#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace ... -o name` writes name_results.db).

    python tools/rocpd_stats.py gpurun_out/train_prof/train_results.db [top_n] [--csv out.csv]
"""
import sqlite3
import sys


def main() -> int:
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 30
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id "
                       "group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    lines = ["name,calls,total_ms,avg_us,percent"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2] / 1e6:.3f},{r[3] / 1e3:.2f},{100 * r[2] / tot:.2f}")
    for r in rows[:top]:
        print(f"{r[2] / 1e6:9.2f} ms {100 * r[2] / tot:5.1f}% n={r[1]:5d} avg={r[3] / 1e3:9.1f}us  {r[0][:120]}")
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
