#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass (8 SQ counters, no trace domains) of bench.py per precision; per-kernel
# table (markdown) of where the waves' cycles go.  Units: SQ_* count quad-cycles summed over waves
# (MI355X_MICROARCH.md); VALU time = SQ_ACTIVE_INST_VALU * 4 cycles / 1024 SIMDs / clock.
#   tools/sq_probe.sh <out.md>
set -u
OUT=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for PR in 64 32; do
  rm -rf /tmp/sq_probe
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    -d /tmp/sq_probe -o p -- python $REPO/bench.py --precision $PR --no-cpu-baseline --no-alt --steps 4 --warmup 1 > /dev/null 2>/tmp/sq_probe.err
  DB=$(find /tmp/sq_probe -name '*.db' | head -1)
  python - "$DB" $PR >> $OUT <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"))
k = {}
for name, cn, v in rows:
    if "fpm::" in name:
        n = name.replace("void ", "")
        n = n[: n.index("(")] if "(" in n else n
        k.setdefault(n[:70], {})[cn] = v
print("\n### fp%s meshes, configs[1] sizes\n" % sys.argv[2])
print("| kernel | wave cycles (quad) | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | of which VALU | LDS | LDS issue stall | LDS bank conflict cycles | VALU time at 2.4 GHz |")
print("|---|---|---|---|---|---|---|---|---|---|")
for n, v in sorted(k.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = v.get("SQ_WAVE_CYCLES", 0)
    if w < 1e7:
        continue
    pct = lambda key: "%.0f %%" % (100 * v.get(key, 0) / w)
    print("| `%s` | %.3g | %s | %s | %s | %s | %s | %s | %.3g | %.2f ms |" % (
        n, w, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"),
        pct("SQ_ACTIVE_INST_LDS"), pct("SQ_WAIT_INST_LDS"), v.get("SQ_LDS_BANK_CONFLICT", 0),
        v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / 2.4e9 * 1e3))
PY
done
tail -1 /tmp/sq_probe.err
