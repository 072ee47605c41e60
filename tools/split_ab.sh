#!/bin/bash
# tools/split_ab.sh [N precision ncube]: the several-waves-per-row readout of the long rows (readout_split_kernel /
# readout_split3_kernel, FPMHIP_RO_SPLIT = 1 | 2 | 3; readout_split_ws_kernel, = 5) against the one-wave-per-row shapes (= 0), one rank of eight
# (tools/rank_share_bench.py), same box.
N=${1:-2048}; PREC=${2:-32}; NCUBE=${3:-0}
mkdir -p gpurun_out/split
for v in ${SPLIT_VARIANTS:-0 1 2 3}; do
  f=gpurun_out/split/rs_${N}_${PREC}_v$v
  FPMHIP_RO_SPLIT=$v timeout 900 python tools/rank_share_bench.py $N $PREC $NCUBE > $f.json 2> $f.err
  echo "N $N prec $PREC split $v rc $?"
  python - $f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels"]
    print("  parity %.3g  readout %.3f ms (frac %.3f)  paint %.3f" % (d["parity_vs_small_cube"], k["readout"]["ms_per_launch"], k["readout"].get("frac_of_8TBps",0), k["paint"]["ms_per_launch"]))
except Exception as e:
    print("  parse failed", e)
P
done
