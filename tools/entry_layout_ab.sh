#!/bin/bash
# Round-6 A/B of the strip-entry layout, run ON THE GPU BOX: gpurun -- 'bash tools/entry_layout_ab.sh'.
#   variant 0 = fastpm_amd/libfastpm_hip.so       : four arrays of 8-byte values (sx, sy, sz, scell)
#   variant 1 = build/libfastpm_hip_aos.so        : one 32-byte record per entry (-DFPM_ENTRY_AOS=1, fpm_internal.h); build it
#               first (no GPU needed): compile fpm_plan / fpm_particles / fpm_strips / fpm_step / fpm_force.hip with the
#               Makefile's flags + -DFPM_ENTRY_AOS=1 into a scratch directory and link them with the other objects of
#               fastpm_amd/csrc into build/libfastpm_hip_aos.so (build/ is git-ignored and travels with gpurun)
# (1) the record build must pass the strip / force parity tests, bit for bit; (2) timing side by side on configs[1] and the
# 1024^3 mesh; (3) kernel trace + PMC traffic of both.  Output: gpurun_out/r06_entry_layout_*.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
cp fastpm_amd/libfastpm_hip.so /tmp/soa.so
cp build/libfastpm_hip_aos.so fastpm_amd/libfastpm_hip.so
python -m pytest tests/test_gpu_strips.py tests/test_gpu_force.py tests/test_gpu_step.py -x -q > $OUT/r06_entry_layout_aos_tests.log 2>&1
echo "aos parity tests rc=$?" | tee $OUT/r06_entry_layout_ab.txt
grep -E "passed|failed" $OUT/r06_entry_layout_aos_tests.log | tail -2 | tee -a $OUT/r06_entry_layout_ab.txt
cp /tmp/soa.so fastpm_amd/libfastpm_hip.so
bash tools/ab_libs.sh "" 3 build/libfastpm_hip_aos.so 2>&1 | tee -a $OUT/r06_entry_layout_ab.txt
bash tools/ab_libs.sh "--nc 512 --nmesh 1024" 2 build/libfastpm_hip_aos.so 2>&1 | tee -a $OUT/r06_entry_layout_ab.txt
bash tools/ab_libs.sh "--load c" 2 build/libfastpm_hip_aos.so 2>&1 | tee -a $OUT/r06_entry_layout_ab.txt
bash tools/profile_round.sh r06_entry_soa kspace
cp build/libfastpm_hip_aos.so fastpm_amd/libfastpm_hip.so
bash tools/profile_round.sh r06_entry_aos kspace
cp /tmp/soa.so fastpm_amd/libfastpm_hip.so
python - <<'PY' | tee -a $OUT/r06_entry_layout_ab.txt
import json
for tag in ("soa", "aos"):
    t = json.load(open("gpurun_out/r06_entry_%s_kspace_traffic.json" % tag))
    for n, k in t["kernels"].items():
        if any(s in n for s in ("bin_scatter_wave", "paint_march", "readout_march3")):
            print(tag, n[:70], "avg_us %.1f" % k["avg_us"], "hbm GB %.3f" % (k["hbm_bytes"] / 1e9))
PY
