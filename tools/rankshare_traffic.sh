#!/bin/bash
# tools/rankshare_traffic.sh "<rank_share_bench args>" <tag>: HBM bytes per launch of the marching kernels of one rank's share,
# from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md; bytes = (2 x FETCH + WRITE) x 1024,
# the gfx950 FETCH_SIZE correction of tools/pmc_traffic.py) -> gpurun_out/<tag>_traffic.md
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; TAG=$2
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prs_f /tmp/prs_w
rocprofv3 --pmc FETCH_SIZE -d /tmp/prs_f -o f -- python $REPO/tools/rank_share_bench.py $ARGS > /tmp/prs_f.json 2>/tmp/prs.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/prs_w -o w -- python $REPO/tools/rank_share_bench.py $ARGS > /tmp/prs_w.json 2>>/tmp/prs.err
F=$(find /tmp/prs_f -name '*.db' | head -1); W=$(find /tmp/prs_w -name '*.db' | head -1)
python - $F $W /tmp/prs_f.json "$ARGS" > $OUT/${TAG}_traffic.md <<'P'
import json, sqlite3, sys
fdb, wdb, js, args = sys.argv[1:5]
q = lambda db, sql: list(sqlite3.connect(db).execute(sql))
fetch = {k: (v, n) for k, v, n in q(fdb, "select kernel_name, avg(value), count(*) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name")}
write = {k: v for k, v in q(wdb, "select kernel_name, avg(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name")}
d = json.loads(open(js).read().strip().splitlines()[-1])
alg = {k: v.get("alg_GB") for k, v in d["kernels"].items()}
print("# HBM bytes per launch, rank_share_bench.py %s (%s)\n" % (args, d["workload"]))
print("| kernel | launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM GB = (2 F + W) x 1024 | algorithmic GB | ratio |\n|---|---|---|---|---|---|---|")
for name, (f, n) in sorted(fetch.items(), key=lambda kv: -kv[1][0]):
    if not name.startswith("void fpm::") and not name.startswith("fpm::"):
        continue
    w = write.get(name, 0.0)
    gb = (2 * f + w) * 1024 / 1e9
    if gb < 0.5:
        continue
    a = None
    for key, sub in (("readout", "readout_"), ("paint", "paint_"), ("xback3", "xback3"), ("k_yback2", "yback2"), ("k_colfft", "colfft_kernel"), ("sort", "bin_scatter")):
        if sub in name:
            a = alg.get(key)
    short = name.replace("void ", "")
    short = short[: short.index("(")] if "(" in short else short
    print("| `%s` | %d | %.0f | %.0f | %.2f | %s | %s |" % (short[:90], n, f, w, gb, ("%.2f" % a) if a else "", ("%.2f" % (gb / a)) if a else ""))
P
cat $OUT/${TAG}_traffic.md; tail -2 /tmp/prs.err
