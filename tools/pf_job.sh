run() { tag=$1; shift; "$@" > gpurun_out/r05_pf_$tag.json 2>>gpurun_out/r05_pf.err; }
cp fastpm_amd/libfastpm_hip.so /tmp/keep.so
run base32 python tools/rank_share_bench.py 2048 32
run base64 python tools/rank_share_bench.py 2048 64
cp build/variants/lib_f32pf6.so fastpm_amd/libfastpm_hip.so; run f32pf6 python tools/rank_share_bench.py 2048 32
cp build/variants/lib_f64pf6.so fastpm_amd/libfastpm_hip.so; run f64pf6 python tools/rank_share_bench.py 2048 64
cp /tmp/keep.so fastpm_amd/libfastpm_hip.so
python tools/rs_print.py r05_pf_base32 r05_pf_f32pf6 r05_pf_base64 r05_pf_f64pf6
