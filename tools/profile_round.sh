#!/bin/bash
# Run ON THE GPU BOX (gpurun): three rocprofv3 passes of bench.py -- kernel trace + stats, FETCH_SIZE,
# WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes -- with the raw databases kept in /tmp
# (they exceed gpurun_out's 64 MiB) and only the summaries written to gpurun_out/.
#   tools/profile_round.sh <tag> <gradient>      e.g.  tools/profile_round.sh r01_e kspace
#   NC=512 NMESH=1024 tools/profile_round.sh r01_e_1024 kspace      (another single-GPU workload)
set -u
TAG=$1; GRAD=$2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
NC=${NC:-256}; NMESH=${NMESH:-512}
B="python $REPO/bench.py --gradient $GRAD --no-cpu-baseline --no-alt --no-secondary --steps 10 --warmup 2 --nc $NC --nmesh $NMESH"
rm -rf /tmp/prof_$GRAD
rocprofv3 --kernel-trace --stats -d /tmp/prof_$GRAD/trace -o t -- $B > $OUT/${TAG}_${GRAD}_bench_under_rocprof.json 2>/tmp/prof_$GRAD.err
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_$GRAD/fetch -o f -- $B > /dev/null 2>>/tmp/prof_$GRAD.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_$GRAD/write -o w -- $B > /dev/null 2>>/tmp/prof_$GRAD.err
T=$(find /tmp/prof_$GRAD/trace -name '*.db' | head -1); F=$(find /tmp/prof_$GRAD/fetch -name '*.db' | head -1); W=$(find /tmp/prof_$GRAD/write -name '*.db' | head -1)
python $REPO/tools/rocprof_summary.py $T $OUT/${TAG}_rocprof_stats_${GRAD}.md
python $REPO/tools/pmc_traffic.py $T $F $W ${TAG}_${GRAD} $NMESH $((NC * NC * NC)) 64 $OUT $GRAD
tail -3 /tmp/prof_$GRAD.err
