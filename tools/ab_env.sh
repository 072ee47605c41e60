#!/bin/bash
# Same-box comparison of builds AND environments: tools/ab_env.sh "<bench args>" <repeats> "<ENV=..> lib.so" ...
# ("-" as the library = fastpm_amd/libfastpm_hip.so itself)
ARGS=$1; REP=$2; shift 2
cp fastpm_amd/libfastpm_hip.so /tmp/abe_base.so
for i in $(seq $REP); do
  for spec in "$@"; do
    lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
    [ "$lib" = "-" ] && lib=/tmp/abe_base.so
    cp $lib fastpm_amd/libfastpm_hip.so
    env $envs python bench.py $ARGS --no-cpu-baseline --no-alt --no-secondary --steps 20 2>/dev/null > /tmp/abe_o.json
    python -c "
import json; d=json.loads(open('/tmp/abe_o.json').read().strip().split(chr(10))[-1]); print('$spec', '|', round(d['ms_per_step'],3), {k: round(v['avg_ms'],3) for k, v in d['stages'].items() if k in ('sort','paint','readout','xback3','k_colfft','k_yback2')})"
  done
done
cp /tmp/abe_base.so fastpm_amd/libfastpm_hip.so
