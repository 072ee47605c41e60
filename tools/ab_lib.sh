#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <other.so> "<bench args>" [repeats]
# runs bench.py alternately with fastpm_amd/libfastpm_hip.so (A) and <other.so> (B) swapped into its place.
OTHER=$1; ARGS=$2; REP=${3:-2}
cp fastpm_amd/libfastpm_hip.so /tmp/ab_A.so; cp $OTHER /tmp/ab_B.so
for i in $(seq $REP); do
  for v in A B; do
    cp /tmp/ab_$v.so fastpm_amd/libfastpm_hip.so
    python bench.py $ARGS --no-cpu-baseline --no-alt --steps 20 2>/dev/null > /tmp/ab_o.json
    python -c "
import json; d=json.loads(open('/tmp/ab_o.json').read()); print('$v', '$ARGS', round(d['ms_per_step'],3), {k: v['avg_ms'] for k, v in d['stages'].items()})"
  done
done
cp /tmp/ab_A.so fastpm_amd/libfastpm_hip.so
