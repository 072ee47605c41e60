#!/bin/bash
# A/B of the readout kernels on the three synthetic loads: FPMHIP_READOUT = 0 direct gather, 1 LDS-staged with
# three meshes per workgroup, 2 (default) LDS-staged with one (tile, component) per workgroup.
for load in a b c; do
  for m in ${MODES:-0 1 2}; do
    r=$(FPMHIP_READOUT=$m python bench.py --load $load --no-cpu-baseline --no-alt --steps 10 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['stages']['readout']['avg_ms'])")
    echo "load=$load FPMHIP_READOUT=$m $BENCH_ARGS: ms/step readout = $r"
  done
done
