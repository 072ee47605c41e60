#!/bin/bash
# Same-box comparison of builds / environments on a rank share: tools/ab_share.sh "<rank_share_bench args>" "<ENV=..> lib.so" ...
# ("-" = fastpm_amd/libfastpm_hip.so itself)
ARGS=$1; shift
cp fastpm_amd/libfastpm_hip.so /tmp/abs_base.so
i=0
for spec in "$@"; do
  i=$((i+1))
  lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
  [ "$lib" = "-" ] && lib=/tmp/abs_base.so
  cp $lib fastpm_amd/libfastpm_hip.so
  env $envs timeout 900 python tools/rank_share_bench.py $ARGS 2>/dev/null > gpurun_out/abs_$i.json
  echo -n "$spec | "; python tools/rs_print.py abs_$i
done
cp /tmp/abs_base.so fastpm_amd/libfastpm_hip.so
