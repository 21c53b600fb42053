#!/usr/bin/env python
"""Sum the PMC counter values per kernel name from one or more rocprofv3 rocpd databases (--pmc passes).
Usage: tools/rocpd_pmc.py <db> [<db> ...]"""
import collections
import sqlite3
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            print("# no counters_collection view in", path, [t for t in tabs if "pmc" in t or "counter" in t][:8])
            continue
        cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
        kcol = next((c for c in ("kernel_name", "name") if c in cols), None)
        ccol = next((c for c in ("counter_name",) if c in cols), None)
        vcol = next((c for c in ("value", "counter_value") if c in cols), None)
        dcol = next((c for c in ("dispatch_id", "id") if c in cols), None)
        if not (kcol and ccol and vcol and dcol):
            print("# unexpected columns in", path, cols)
            continue
        for k, c, v, did in db.execute(f"select {kcol}, {ccol}, {vcol}, {dcol} from counters_collection"):
            agg[k][c] += float(v)
            disp[(k, c)].add((path, did))
    for k in sorted(agg, key=lambda kk: -sum(agg[kk].values())):
        print(k[:120])
        for c, v in sorted(agg[k].items()):
            n = max(1, len(disp[(k, c)]))
            print(f"    {c:30s} total {v:22.0f}   dispatches {n:5d}   per dispatch {v / n:18.1f}")
        per = {c: v / max(1, len(disp[(k, c)])) for c, v in agg[k].items()}
        # derived (per dispatch; the counters come from separate passes over the same launches). SQ_LEVEL_WAVES reads 0 on this stack, so the
        # occupancy is taken from the wave-cycle integral: SQ_WAVE_CYCLES (sum over waves of their resident cycles, 4-cycle quanta like
        # SQ_BUSY_CU_CYCLES) / SQ_BUSY_CU_CYCLES (sum over CUs of their busy cycles) = average waves resident on a busy CU
        if per.get("SQ_WAVE_CYCLES") and per.get("SQ_BUSY_CU_CYCLES"):
            print(f"    {'derived: waves per busy CU':30s} {per['SQ_WAVE_CYCLES'] / per['SQ_BUSY_CU_CYCLES']:10.2f}   (SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES; 32 = full)")
        if per.get("SQ_WAVE_CYCLES") and per.get("SQ_WAIT_INST_ANY") is not None and "SQ_WAIT_INST_ANY" in per:
            print(f"    {'derived: wave cycles waiting':30s} {per['SQ_WAIT_INST_ANY'] / per['SQ_WAVE_CYCLES']:10.2f}   (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)")
        if per.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_ANY" in per:
            print(f"    {'derived: wave cycles issuing':30s} {per['SQ_ACTIVE_INST_ANY'] / per['SQ_WAVE_CYCLES']:10.2f}   (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)")
        if per.get("SQ_WAVES") and "SQ_INSTS_VALU" in per:
            print(f"    {'derived: VALU insts per wave':30s} {per['SQ_INSTS_VALU'] / per['SQ_WAVES']:10.1f}")


if __name__ == "__main__":
    main()
