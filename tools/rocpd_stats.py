#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite database (ROCm 7.2 default output
of `rocprofv3 --kernel-trace --stats`). Usage: tools/rocpd_stats.py <results.db> [steps_in_trace] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                           "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace --stats; {steps:g} train steps in the trace)")
    print(f"# total kernel time {tot / 1e6:.3f} ms = {tot / 1e6 / steps:.3f} ms per step")
    print(f"{'total_ms':>10} {'%':>6} {'calls':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel")
    for n, c, s, a, mn, mx, vg, ag, lds in rows:
        print(f"{s / 1e6:10.3f} {100 * s / tot:6.2f} {c:6d} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {vg or 0:5d} {ag or 0:5d} {lds or 0:7d}  {n[:140]}")


if __name__ == "__main__":
    main()
