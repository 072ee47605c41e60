#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass of bench.py with the given counters, per-kernel averages to stdout.
#   tools/pmc_probe.sh "<bench args>" COUNTER [COUNTER ...]
set -u
ARGS=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_probe
rocprofv3 --pmc "$@" -d /tmp/pmc_probe -o p -- python $REPO/bench.py $ARGS --no-cpu-baseline --no-alt --steps 4 --warmup 1 > /dev/null 2>/tmp/pmc_probe.err
DB=$(find /tmp/pmc_probe -name '*.db' | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
for name, cn, v, n in sorted(rows):
    if "fpm::" in name or "fft_rtc" in name:
        print("%-60s %-22s %12.4g  (%d)" % (name.replace("void ", "")[:60], cn, v, n))
PY
tail -2 /tmp/pmc_probe.err
