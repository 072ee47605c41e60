#!/bin/bash
# Round-6 profile set, run ON THE GPU BOX: gpurun -- 'bash tools/profile_r06.sh'.  Summaries land in gpurun_out/r06/
# (copy the ones to keep into profiles/); the raw rocprofv3 databases stay in /tmp.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
# 1. configs[1] and the 1024^3 mesh: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (three separate passes each)
bash tools/profile_round.sh r06 kspace
NC=512 NMESH=1024 bash tools/profile_round.sh r06_1024 kspace
# 2. where the waves' cycles go (one --pmc pass of SQ counters per precision)
bash tools/sq_probe.sh $OUT/r06_sq_counters.md
# 3. ONE rank's share of the 8-GPU configs under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 64" "2048 64" "2048 32" "3072 32 128" "3072 64 128"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf /tmp/prof_rs
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rs -o t -- python $REPO/tools/rank_share_bench.py $cfg > $OUT/r06_rankshare_$tag.json 2>/tmp/prof_rs.err
  T=$(find /tmp/prof_rs -name '*.db' | head -1)
  python $REPO/tools/rocprof_summary.py $T $OUT/r06_rankshare_${tag}_rocprof_stats.md
done
# 3a'. the same slab rank at 1024^3 with FPMHIP_GRADIENT_XSTENCIL (one mesh back through the transpose; the stencil pass timed)
python $REPO/tools/rank_share_bench.py 1024 64 0 0 xstencil > $OUT/r06_rankshare_1024_64_xstencil.json 2>/dev/null
# 3b. ONE rank of the reference's 4 x 2 PENCIL mesh (tests/rank_share.py: ReplicatedPencilForce): strip tiles at 1024^3
#     (the marching kernels on the exchange chunks), strip tiles at 2048^3 fp64 too (M = 1024: one workgroup per CU), and the
#     1024^3 share on box tiles for the A/B
for cfg in "1024 64 0 0 pencil" "1024 64 0 2 pencil" "2048 64 0 0 pencil"; do
  set -- $cfg
  tag=pencil_$1_$2; [ "$4" = "2" ] && tag=${tag}_boxes
  rm -rf /tmp/prof_rs
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rs -o t -- python $REPO/tools/rank_share_bench.py $cfg > $OUT/r06_rankshare_$tag.json 2>/tmp/prof_rs.err
  T=$(find /tmp/prof_rs -name '*.db' | head -1)
  python $REPO/tools/rocprof_summary.py $T $OUT/r06_rankshare_${tag}_rocprof_stats.md
done
# 4. the bench lines (default, fp32, clustered / adversarial loads, box tiles)
cd $REPO
python bench.py --steps 20 --warmup 5 > $OUT/r06_bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --precision 32 --no-cpu-baseline > $OUT/r06_bench_fp32.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --load b --no-cpu-baseline --no-secondary > $OUT/r06_bench_load_b.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --load c --no-cpu-baseline --no-secondary > $OUT/r06_bench_load_c.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --paint-mode 2 --no-cpu-baseline --no-secondary > $OUT/r06_boxes_bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --static --no-cpu-baseline --no-secondary > $OUT/r06_bench_static.json 2>/dev/null
python tools/bench_rows.py > $OUT/r06_rows.json 2>/dev/null
# 5. the N > 1 code path on this one GPU (ranks share it, exchanges staged through the host: a dry run, never a measurement):
#    plain `python3 bench.py --gpus 2` launches itself; both legs (c_dropin, python_mirror) in the line
python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/r06_bench_gpus2_dryrun.json 2>/dev/null
python bench.py --gpus 4 --nprocy 2 --steps 3 --warmup 1 --nc 128 --nmesh 256 > $OUT/r06_bench_gpus4_pencil_dryrun.json 2>/dev/null
ls -la $OUT | tail -30
