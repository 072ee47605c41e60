#!/bin/bash
# one rank of the 4 x 2 pencil mesh at 3072^3 fp32 with configs[4]'s LOAD (134 M particles per rank:
# ncube = 256 on the 768-cell cube, B = 3), strip tiles vs box tiles, under rocprofv3 --kernel-trace --stats.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "3072 32 256 0 pencil" "3072 32 256 2 pencil"; do
  set -- $cfg
  tag=pencil_$1_$2_$3; [ "$4" = "2" ] && tag=${tag}_boxes
  rm -rf /tmp/prof_rs
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rs -o t -- python $REPO/tools/rank_share_bench.py $cfg > $OUT/r06_rankshare_$tag.json 2>/tmp/prof_rs.err
  T=$(find /tmp/prof_rs -name '*.db' | head -1)
  python $REPO/tools/rocprof_summary.py $T $OUT/r06_rankshare_${tag}_rocprof_stats.md
  tail -2 /tmp/prof_rs.err
  python -c "
import json; d=json.load(open('$OUT/r06_rankshare_$tag.json')); print(d['workload'], round(d['per_rank_compute_ms_per_step'],1), d['parity_vs_small_cube']); print({k:(round(v['ms_per_launch'],2),v['launches'],round(v.get('frac_of_8TBps',0),3)) for k,v in d['kernels'].items()})"
done
