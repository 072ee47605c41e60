python -m pytest "tests/test_gpu_strips.py::test_largest_strip_meshes_agree_with_box_tiles" -q -m gpu -k "2048-32 or 1536-32" > gpurun_out/r05_rows2.log 2>&1; tail -4 gpurun_out/r05_rows2.log
run() { tag=$1; shift; "$@" > gpurun_out/r05_rs_$tag.json 2>>gpurun_out/r05_rs.err; }
run 2048_32_rows2 python tools/rank_share_bench.py 2048 32
FPMHIP_RO_ROWS2=0 python tools/rank_share_bench.py 2048 32 > gpurun_out/r05_rs_2048_32_rows1.json 2>>gpurun_out/r05_rs.err
cp fastpm_amd/libfastpm_hip.so /tmp/keep.so
cp build/variants/lib_pf3w2.so fastpm_amd/libfastpm_hip.so; run 2048_32_pf3w2 python tools/rank_share_bench.py 2048 32
cp build/variants/lib_pf4w2.so fastpm_amd/libfastpm_hip.so; run 2048_32_pf4w2 python tools/rank_share_bench.py 2048 32
cp /tmp/keep.so fastpm_amd/libfastpm_hip.so
run 3072_32_rows2 python tools/rank_share_bench.py 3072 32 128
FPMHIP_RO_ROWS2=0 python tools/rank_share_bench.py 3072 32 128 > gpurun_out/r05_rs_3072_32_rows1.json 2>>gpurun_out/r05_rs.err
python tools/rs_print.py r05_rs_2048_32_rows2 r05_rs_2048_32_rows1 r05_rs_2048_32_pf3w2 r05_rs_2048_32_pf4w2 r05_rs_3072_32_rows2 r05_rs_3072_32_rows1
tail -3 gpurun_out/r05_rs.err
