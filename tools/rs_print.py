import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open("gpurun_out/%s.json"%f))
        print(f, round(d["per_rank_compute_ms_per_step"],2), "%.2e"%d["parity_vs_small_cube"], {k:(round(v["ms_per_launch"],2),v["launches"]) for k,v in d["kernels"].items() if k in ("sort","paint","readout","k_zc2r","k_rowfft","xback3","k_colfft","k_yback2")})
    except Exception as e: print(f, "failed", e)
