#!/bin/bash
# tools/env_sweep.sh VAR "v1 v2 ..." "<N precision>" ...: one rank's share (tools/rank_share_bench.py) per value of an A/B switch
VAR=$1; VALS=$2; shift 2
for a in "$@"; do for v in $VALS; do
  env $VAR=$v python tools/rank_share_bench.py $a 2>/dev/null | tail -1 > /tmp/ab_o.json
  python -c "
import json; d=json.loads(open('/tmp/ab_o.json').read()); k=d['kernels']
print('$VAR=$v', '$a', 'step %.2f' % d['per_rank_compute_ms_per_step'], ' '.join('%s %.2f' % (n, k[n]['ms_per_launch']) for n in ('k_colfft', 'k_yback2', 'xback3', 'k_rowfft', 'k_zc2r') if n in k), 'err %.1e' % d['parity_vs_small_cube'])"
done; done
