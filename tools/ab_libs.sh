#!/bin/bash
# Same-box comparison of several builds of the library: tools/ab_libs.sh "<bench args>" <repeats> lib1.so lib2.so ...
# (fastpm_amd/libfastpm_hip.so itself is variant 0)
ARGS=$1; REP=$2; shift 2
cp fastpm_amd/libfastpm_hip.so /tmp/ab_0.so
n=0; for l in "$@"; do n=$((n+1)); cp $l /tmp/ab_$n.so; done
for i in $(seq $REP); do
  for v in $(seq 0 $n); do
    cp /tmp/ab_$v.so fastpm_amd/libfastpm_hip.so
    python bench.py $ARGS --no-cpu-baseline --no-alt --no-secondary --steps 20 2>/dev/null > /tmp/ab_o.json
    python -c "
import json; d=json.loads(open('/tmp/ab_o.json').read().strip().split(chr(10))[-1]); print('v$v', '$ARGS', round(d['ms_per_step'],3), {k: round(v['avg_ms'],3) for k, v in d['stages'].items() if k in ('sort','paint','readout')})"
  done
done
cp /tmp/ab_0.so fastpm_amd/libfastpm_hip.so
