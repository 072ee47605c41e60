#!/usr/bin/env python3
"""Build profiles/<tag>_traffic.json and profiles/<tag>_kernel_trace.md from three rocprofv3 runs of
bench.py (one --kernel-trace --stats, one --pmc FETCH_SIZE, one --pmc WRITE_SIZE; separate passes as
MI355X_MICROARCH.md prescribes).  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the
x2 is the gfx950 FETCH_SIZE correction (128-B requests tallied at 64 B), confirmed on kernels whose
reads are exactly one mesh.
usage: pmc_traffic.py <trace.db> <fetch.db> <write.db> <tag> <nmesh> <particles> <precision> [outdir [gradient]]"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = n.replace("void ", "")
    return (n[: n.index("(")] if "(" in n and len(n) > 100 else n)[:100]


def main():
    trace, fdb, wdb, tag, nmesh, npart, prec = sys.argv[1:8]
    outdir = sys.argv[8] if len(sys.argv) > 8 else os.path.join(ROOT, "profiles")
    q = lambda db, sql: list(sqlite3.connect(db).execute(sql))
    fetch = {k: v for k, v in q(fdb, "select kernel_name, avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name")}
    write = {k: v for k, v in q(wdb, "select kernel_name, avg(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name")}
    top = q(trace, "select name,total_calls,total_duration,average,percentage from top_kernels")
    kern = {}
    for name, calls, tot, avg, pct in top:
        if pct < 0.5:
            continue
        f, w = fetch.get(name, 0.0), write.get(name, 0.0)
        kern[short(name)] = {"calls": calls, "avg_us": avg, "pct": pct, "fetch_kib_raw": f, "write_kib": w,
                             "hbm_bytes": (2 * f + w) * 1024}

    def hbm(sub, must=None):
        for n, v in kern.items():
            if sub in n and (must is None or must in n):
                return v["hbm_bytes"]
        return None

    def add(*xs):
        return None if any(x is None for x in xs) else sum(xs)

    N = nmesh
    stage = {
        "sort": hbm("bin_scatter_wave_kernel") or hbm("bin_scatter_kernel<true, false>") or add(hbm("bin_kernel<false>"), hbm("bin_kernel<true>")),
        "paint": hbm("paint_march") or hbm("paint_strips") or hbm("paint_tiles"),
        "readout": hbm("readout_march") or hbm("readout_strips") or hbm("readout1of3_tiles") or hbm("readout3_tiles") or hbm("readout_grad_tiles") or hbm("readout_kernel") or hbm("readout_grad_kernel"),
        "xback3": hbm("xback3"),
        "k_yback2": hbm("yback2"),
        "k_colfft": hbm("colfft_kernel<%s" % N),
        "k_rowfft": hbm("rowfft_r2c"),
        "k_zc2r": hbm("rowfft_c2r") or hbm("C2R"),
        "transfer": hbm("transfer_kernel"),
    }
    gradient = sys.argv[9] if len(sys.argv) > 9 else "kspace"
    out = {"config": {"nmesh": int(nmesh), "particles": int(npart), "precision": int(prec), "n_gpus": 1,
                      "fft": "hand-written row + column passes", "gradient": gradient},
           "method": __doc__.split("usage")[0].strip(),
           "hbm_bytes_per_launch_by_stage": {k: v for k, v in stage.items() if v is not None},
           "kernels": kern}
    json.dump(out, open(os.path.join(outdir, tag + "_traffic.json"), "w"), indent=1)
    with open(os.path.join(outdir, tag + "_kernel_trace.md"), "w") as f:
        f.write("# %s: rocprofv3 --kernel-trace --stats of `python bench.py --steps 10 --warmup 2` "
                "(%s^3 mesh fp%s, %s particles, 1 MI355X), PMC columns from separate --pmc passes\n\n" % (tag, nmesh, prec, npart))
        f.write("| kernel | calls | avg_us | % | FETCH_SIZE KiB raw | WRITE_SIZE KiB | HBM bytes/launch |\n|---|---|---|---|---|---|---|\n")
        for n, v in kern.items():
            f.write("| `%s` | %d | %.1f | %.1f | %.4g | %.4g | %.4g |\n" % (n, v["calls"], v["avg_us"], v["pct"], v["fetch_kib_raw"], v["write_kib"], v["hbm_bytes"]))
    print(json.dumps(out["hbm_bytes_per_launch_by_stage"]))


if __name__ == "__main__":
    main()
