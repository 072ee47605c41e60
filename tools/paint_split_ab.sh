#!/bin/bash
# tools/paint_split_ab.sh: the several-waves-per-row paint of the long rows (paint_split_kernel, FPMHIP_PT_SPLIT = 1) against
# the workgroup-barrier shape (= 0), one rank of eight (tools/rank_share_bench.py), same box.
mkdir -p gpurun_out/split
for cfg in "2048 32 0" "3072 32 128" "2048 64 0"; do
  set -- $cfg
  for v in 0 1; do
    f=gpurun_out/split/pt_$1_$2_v$v
    FPMHIP_PT_SPLIT=$v timeout 900 python tools/rank_share_bench.py $1 $2 $3 > $f.json 2> $f.err
    echo "N $1 prec $2 paint split $v rc $?"
    python - $f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels"]
    print("  parity %.3g  paint %.3f ms (frac %.3f)  readout %.3f" % (d["parity_vs_small_cube"], k["paint"]["ms_per_launch"], k["paint"].get("frac_of_8TBps",0), k["readout"]["ms_per_launch"]))
except Exception as e:
    print("  parse failed", e)
P
  done
done
