#!/bin/bash
# tools/ab_rankshare_libs.sh "<N precision>" <repeats> lib1.so lib2.so ...: one rank's share (tools/rank_share_bench.py) with
# several builds of the library on one box (fastpm_amd/libfastpm_hip.so itself is variant 0)
A=$1; REP=$2; shift 2
cp fastpm_amd/libfastpm_hip.so /tmp/ab_0.so
n=0; for l in "$@"; do n=$((n+1)); cp $l /tmp/ab_$n.so; done
for i in $(seq $REP); do for v in $(seq 0 $n); do
  cp /tmp/ab_$v.so fastpm_amd/libfastpm_hip.so
  python tools/rank_share_bench.py $A 2>/dev/null | tail -1 > /tmp/ab_o.json
  python -c "
import json; d=json.loads(open('/tmp/ab_o.json').read()); k=d['kernels']
print('v$v', '$A', 'step %.2f' % d['per_rank_compute_ms_per_step'], ' '.join('%s %.2f' % (n, k[n]['ms_per_launch']) for n in ('sort', 'paint', 'readout') if n in k), 'err %.1e' % d['parity_vs_small_cube'])"
done; done
cp /tmp/ab_0.so fastpm_amd/libfastpm_hip.so
