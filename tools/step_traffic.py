#!/usr/bin/env python
"""HBM traffic of WHOLE training steps from the PMC counters.

Two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not share a pass, MI355X_MICROARCH.md "Counter slots") over
`bench.py --steps S --warmup W --no-extras`; every dispatch of the run is tallied, the totals are divided by S + W steps
(the first step's one-time work -- anchor grids, workspace fills -- is a few MB). Units and the gfx950 correction as the guide's
HBM / rocprofv3 section prescribes: both counters in KB, FETCH_SIZE x 2 (wide coalesced reads are tallied at half their bytes).

    python tools/step_traffic.py [--steps 5] [--warmup 3] [--out gpurun_out/step_traffic.json] [--table gpurun_out/step_traffic.txt]

`measure(...)` is what bench.py calls for `step_roofline.traffic`; the per-kernel-family table goes to profiles/.
"""
import argparse
import collections
import glob
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAMILIES = (            # first match wins
    ("weight gradient", r"k_wgrad|k_stem_wgrad|k_colsum"),
    ("norm", r"k_norm"),
    ("conv 3x3x3 persistent (k_ig3r)", r"k_ig3r"),
    ("conv 3x3x3 stride 1 (k_ig3)", r"k_ig3"),
    ("strided data gradient (k_dgs)", r"k_dgs"),
    ("conv generic / strided (k_igemm)", r"k_igemm|k_ig_splitk"),
    ("1x1x1 / transposed (k_pw)", r"k_pw"),
    ("stem block", r"k_stem"),
    ("segmentation branch / loss", r"k_segbranch|k_seg"),
    ("head io / sparse outputs", r"k_head_gather|k_ho_"),
    ("target assignment / sampler / loss", r"k_atss|k_sp_|k_hnm|k_detloss|k_sigmoid|k_anchor"),
    ("weight pack / optimizer", r"k_pack|k_sgd|k_fused_sgd|multi_tensor|fused_sgd"),
    ("torch / rocPRIM / rocBLAS glue", r"."),
)


def _family(kernel):
    for name, pat in FAMILIES:
        if re.search(pat, kernel):
            return name
    return FAMILIES[-1][0]


def _collect(db_path, counter):
    """-> {kernel name: (sum of counter values, dispatches)}"""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else "id"
    agg = collections.defaultdict(lambda: [0.0, set()])
    for k, c, v, d in db.execute(f"select {kcol}, counter_name, {vcol}, {dcol} from counters_collection"):
        if c == counter:
            agg[k][0] += float(v)
            agg[k][1].add(d)
    return {k: (v[0], len(v[1])) for k, v in agg.items()}


def measure(steps=5, warmup=3, bench_args=(), timeout_s=240, keep_table=None):
    """-> dict(hbm_bytes_per_step, read / write split, per-family table) or {"error": ...} / None without rocprofv3."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.isfile("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="nndet_steppmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", NNDET_BENCH_PMC="0")
    per = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup),
                   "--no-extras"] + list(bench_args)
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            tot = collections.defaultdict(lambda: [0.0, 0])
            for dbp in glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True):
                for k, (v, n) in _collect(dbp, ctr).items():
                    tot[k][0] += v
                    tot[k][1] += n
            if not tot:
                return {"error": "no %s values in the rocprofv3 output" % ctr}
            per[ctr] = tot
        nstep = float(steps + warmup)
        fam = collections.defaultdict(lambda: {"read": 0.0, "write": 0.0, "launches": 0.0})
        kern = collections.defaultdict(lambda: {"read": 0.0, "write": 0.0, "launches": 0.0})
        for k, (v, n) in per["FETCH_SIZE"].items():
            b = v * 1024.0 * 2.0 / nstep
            fam[_family(k)]["read"] += b
            fam[_family(k)]["launches"] += n / nstep
            kern[k]["read"] += b
            kern[k]["launches"] += n / nstep
        for k, (v, n) in per["WRITE_SIZE"].items():
            b = v * 1024.0 / nstep
            fam[_family(k)]["write"] += b
            kern[k]["write"] += b
        rd = sum(f["read"] for f in fam.values())
        wr = sum(f["write"] for f in fam.values())
        res = {"hbm_bytes_per_step": int(rd + wr), "hbm_read_bytes_per_step": int(rd), "hbm_write_bytes_per_step": int(wr),
               "launches_per_step": round(sum(f["launches"] for f in fam.values()), 1), "steps_profiled": int(nstep),
               "families": {k: {"read_MB": round(v["read"] / 1e6, 1), "write_MB": round(v["write"] / 1e6, 1),
                                "launches_per_step": round(v["launches"], 1)} for k, v in sorted(fam.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"]))},
               "kernels": {k[:160]: {"read_MB": round(v["read"] / 1e6, 1), "write_MB": round(v["write"] / 1e6, 1), "launches_per_step": round(v["launches"], 2)}
                           for k, v in sorted(kern.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"]))[:24]},
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps %d --warmup %d --no-extras`, all dispatches / %d steps; "
                         "KB units, FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md)" % (steps, warmup, int(nstep))}
        if keep_table:
            with open(keep_table, "w") as f:
                f.write("# HBM traffic per training step by kernel family (MB), %s\n" % res["source"])
                f.write("%-45s %10s %10s %10s %9s\n" % ("family", "read", "write", "total", "launches"))
                for k, v in res["families"].items():
                    f.write("%-45s %10.1f %10.1f %10.1f %9.1f\n" % (k, v["read_MB"], v["write_MB"], v["read_MB"] + v["write_MB"], v["launches_per_step"]))
                f.write("%-45s %10.1f %10.1f %10.1f %9.1f\n\n" % ("TOTAL", rd / 1e6, wr / 1e6, (rd + wr) / 1e6, res["launches_per_step"]))
                f.write("# per kernel (MB per step), largest first\n")
                for k, v in sorted(kern.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"]))[:70]:
                    f.write("%9.1f %9.1f %7.1f  %s\n" % (v["read"] / 1e6, v["write"] / 1e6, v["launches"], k[:150]))
        return res
    except Exception as e:                                        # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "step_traffic.json"))
    ap.add_argument("--table", default=os.path.join(ROOT, "gpurun_out", "step_traffic.txt"))
    ap.add_argument("bench_args", nargs="*")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    r = measure(a.steps, a.warmup, a.bench_args, keep_table=a.table)
    with open(a.out, "w") as f:
        json.dump(r, f, indent=1)
    print(json.dumps({k: v for k, v in (r or {}).items() if k not in ("families", "kernels")}))
