#!/bin/bash
# Same-box A/B of two builds of the library on one rank's share of the 8-GPU configs:
#   tools/ab_rankshare.sh <other.so> "<N precision>" ["<N precision>" ...]
# runs tools/rank_share_bench.py alternately with fastpm_amd/libfastpm_hip.so (A) and <other.so> (B) swapped into its place.
OTHER=$1; shift
cp fastpm_amd/libfastpm_hip.so /tmp/ab_A.so; cp $OTHER /tmp/ab_B.so
for a in "$@"; do
  for v in A B; do
    cp /tmp/ab_$v.so fastpm_amd/libfastpm_hip.so
    python tools/rank_share_bench.py $a 2>/dev/null | tail -1 > /tmp/ab_o.json
    python -c "
import json; d=json.loads(open('/tmp/ab_o.json').read()); k=d['kernels']
print('$v', '$a', 'step %.2f' % d['per_rank_compute_ms_per_step'], ' '.join('%s %.2f' % (n, k[n]['ms_per_launch']) for n in ('k_colfft', 'k_yback2', 'xback3', 'k_rowfft', 'k_zc2r', 'sort', 'paint', 'readout') if n in k), 'err %.1e' % d['parity_vs_small_cube'])"
  done
done
cp /tmp/ab_A.so fastpm_amd/libfastpm_hip.so
