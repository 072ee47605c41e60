#!/bin/bash
# A/B of the readout kernels (direct gather vs LDS-staged) on the three synthetic loads, both gradient modes.
for load in a b c; do
  for env in "FPMHIP_READOUT=0 FPMHIP_READOUT_GRAD=1" "FPMHIP_READOUT=1 FPMHIP_READOUT_GRAD=2"; do
    for grad in kspace real; do
      r=$(env $env python bench.py --load $load --gradient $grad --no-cpu-baseline --no-alt --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['stages']['readout']['avg_ms'], d['stages']['sort']['avg_ms'], d['stages']['paint']['avg_ms'])")
      echo "load=$load $env gradient=$grad : ms/step readout sort paint = $r"
    done
  done
done
