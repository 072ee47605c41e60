#!/bin/bash
# Round 6, after the several-waves-per-row readouts: the rank-share profiles they change, under rocprofv3 --kernel-trace --stats
# (gpurun -- 'bash tools/profile_r06_split.sh'; summaries land in gpurun_out/, copy the ones to keep into profiles/).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "2048 32" "3072 32 128" "2048 32 0 0 pencil" "3072 32 256 0 pencil" "3072 32 256 2 pencil"; do
  set -- $cfg
  tag=$1_$2; [ $# -ge 3 ] && [ "$3" != "0" ] && tag=${tag}_$3
  [ $# -ge 5 ] && tag=pencil_$tag
  [ $# -ge 4 ] && [ "$4" = "2" ] && tag=${tag}_boxes
  rm -rf /tmp/prof_rs
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rs -o t -- python $REPO/tools/rank_share_bench.py $cfg > $OUT/r06_rankshare_$tag.json 2>/tmp/prof_rs.err
  T=$(find /tmp/prof_rs -name '*.db' | head -1)
  python $REPO/tools/rocprof_summary.py $T $OUT/r06_rankshare_${tag}_rocprof_stats.md
  python -c "
import json; d=json.load(open('$OUT/r06_rankshare_$tag.json')); print('$tag', d['workload'], round(d.get('per_rank_compute_ms_per_step', 0),1), d['parity_vs_small_cube']); print({k:(round(v['ms_per_launch'],2),v['launches'],round(v.get('frac_of_8TBps',0),3)) for k,v in d['kernels'].items()})"
done
