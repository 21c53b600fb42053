#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite database (ROCm 7.2 default output
of `rocprofv3 --kernel-trace --stats`). Usage: tools/rocpd_stats.py <results.db> [steps_in_trace] [--by-grid] > profiles/<name>.txt
--by-grid additionally splits every kernel by its launch grid (= by layer shape) for the kernels matching k_ig / k_wgrad / k_norm."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1.0
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                           "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace --stats; {steps:g} train steps in the trace)")
    print(f"# total kernel time {tot / 1e6:.3f} ms = {tot / 1e6 / steps:.3f} ms per step")
    print(f"{'total_ms':>10} {'%':>6} {'calls':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel")
    for n, c, s, a, mn, mx, vg, ag, lds in rows:
        print(f"{s / 1e6:10.3f} {100 * s / tot:6.2f} {c:6d} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {vg or 0:5d} {ag or 0:5d} {lds or 0:7d}  {n[:140]}")
    if "--by-grid" in sys.argv:
        print("\n# split by launch grid (grid_x/256 workgroups, grid_y, grid_z) -- per step")
        rows = list(db.execute("select name, grid_x / workgroup_x, grid_y, grid_z, lds_size, count(*), sum(end-start), avg(end-start) from kernels "
                               "where name like '%k_ig%' or name like '%k_wgrad%' or name like '%k_norm%' or name like '%k_stem%' "
                               "group by name, grid_x, grid_y, grid_z, lds_size order by 7 desc"))
        print(f"{'ms/step':>9} {'calls/step':>10} {'avg_us':>9} {'grid':>18} {'lds':>7}  kernel")
        for n, gx, gy, gz, lds, c, sm, a in rows[:80]:
            print(f"{sm / 1e6 / steps:9.3f} {c / steps:10.1f} {a / 1e3:9.2f} {f'{gx}x{gy}x{gz}':>18} {lds or 0:7d}  {n[:90]}")


def timeline(path, steps, out):
    """--timeline <file>: kernels of the LAST step of the trace in start order (start offset, duration, queue, grid, name) plus the
    union-busy time and the time with >= 2 kernels in flight (the detection-head levels run on side streams)."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(db.execute(f"select start, end, name, grid_x / workgroup_x, grid_y, grid_z, {qcol} from kernels order by start"))
    # one step = from one launch of the stem forward kernel (first kernel of a forward pass) to the next
    # (fused stem block: its statistics pass k_stem_fwd3<T, 1> opens the forward pass; the other instantiations also serve the
    # recompute pass and the rank-1 backward of decoder.out.P0)
    marks = [i for i, r in enumerate(rows) if "k_stem_fwd3<" in r[2] and ", 1>(" in r[2]]
    if len(marks) < 2:
        marks = [i for i, r in enumerate(rows) if "k_stem_fwd" in r[2]]
    if len(marks) >= 2:
        rows = rows[marks[-2]:marks[-1]]
    per = len(rows)
    t0 = rows[0][0]
    ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
    busy = multi = 0
    depth = 0
    last = ev[0][0]
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: multi += t - last
        depth += d
        last = t
    with open(out, "w") as f:
        f.write(f"# one step (stem forward to the next stem forward): {per} kernels, span {(max(r[1] for r in rows) - t0) / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, "
                f">= 2 kernels in flight {multi / 1e6:.3f} ms, sum of durations {sum(r[1] - r[0] for r in rows) / 1e6:.3f} ms\n")
        gaps = []
        cur_end = rows[0][1]
        for st, en, *_ in rows[1:]:
            if st > cur_end: gaps.append(st - cur_end)
            cur_end = max(cur_end, en)
        f.write(f"# idle gaps: {len(gaps)} totalling {sum(gaps) / 1e6:.3f} ms; > 20 us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.3f} ms\n")
        f.write("# start_us   dur_us  queue  grid  kernel\n")
        for st, en, name, gx, gy, gz, q in rows:
            f.write(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {q!s:>5} {f'{gx}x{gy}x{gz}':>14}  {name[:70]}\n")


if __name__ == "__main__":
    if "--timeline" in sys.argv:
        i = sys.argv.index("--timeline")
        timeline(sys.argv[1], float(sys.argv[2]), sys.argv[i + 1])
    main()
