#!/bin/bash
# Run ON THE GPU BOX: the C multi-rank sequence on virtual ranks over the asynchronous loopback under rocprofv3's kernel +
# memory-copy trace; tools/overlap_analyze.py turns the trace into "how much of the exchange copies' time ran beside kernels".
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CH in 4 -1; do
  rm -rf /tmp/prof_ov
  rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_ov -o t -- python $REPO/tools/overlap_run.py 512 2 $CH 4 > $OUT/r05_overlap_run_$CH.txt 2>/tmp/prof_ov.err
  T=$(find /tmp/prof_ov -name '*.db' | head -1)
  python $REPO/tools/overlap_analyze.py $T > $OUT/r05_overlap_chunks_$CH.md 2>&1
  tail -2 $OUT/r05_overlap_run_$CH.txt
done
cat $OUT/r05_overlap_chunks_4.md | head -60
