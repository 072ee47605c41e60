set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
NC=512 NMESH=1024 bash tools/profile_round.sh r06_1024 kspace > gpurun_out/profile_round_1024.log 2>&1
cp gpurun_out/r06_1024_kspace_traffic.json gpurun_out/r06_1024_kspace_kernel_trace.md profiles/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2>gpurun_out/r06_bench.err
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 64" "2048 64"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf /tmp/prof_rs
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rs -o t -- python $REPO/tools/rank_share_bench.py $cfg > $REPO/gpurun_out/r06_rankshare_$tag.json 2>/tmp/prof_rs.err
  T=$(find /tmp/prof_rs -name '*.db' | head -1)
  python $REPO/tools/rocprof_summary.py $T $REPO/gpurun_out/r06_rankshare_${tag}_rocprof_stats.md
  python -c "
import json; d=json.load(open('$REPO/gpurun_out/r06_rankshare_$tag.json')); print('$tag', round(d.get('per_rank_compute_ms_per_step', 0),2), d['parity_vs_small_cube'], {k:(round(v['ms_per_launch'],2), round(v.get('frac_of_8TBps',0),3)) for k,v in d['kernels'].items() if k in ('readout','paint','xback3')})"
done
cd $REPO
python -c "
import json; d=json.load(open('gpurun_out/r06_bench.json')); m=d['secondary']['mesh1024']; print('bench', d['ms_per_step'], d['roofline']['frac'], 'mesh1024', m['ms_per_step'], m['step_frac'], m['kernel_fracs'], m['roofline'].get('frac'), m['roofline'].get('frac_rocprof'))"
