#!/bin/bash
# Copy what tools/r03_profiles.sh / r03_benches.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call) and print the headline numbers.
for t in mx valu stress10k stress10k_valu textured; do
  [ -d gpurun_out/prof_$t ] || continue
  cp gpurun_out/prof_$t/kernel_stats.csv profiles/r03_${t}_kernel_stats.csv
  cp gpurun_out/prof_$t/sq_pmc.txt profiles/r03_${t}_sq_pmc.txt
  grep -v fma_peak_kernel gpurun_out/prof_$t/hbm.txt > profiles/r03_${t}_hbm_traffic.txt
  cp gpurun_out/prof_$t/phase_split.txt profiles/r03_${t}_phase_split.txt
done
[ -f gpurun_out/t_final.log ] && cp gpurun_out/t_final.log profiles/r03_gpu_tests.log
for f in r03_mx_bench r03_stress10k_bench r03_textured_bench r03_node_bench; do [ -f gpurun_out/$f.json ] && cp gpurun_out/$f.json profiles/; done
python3 - <<'PY'
import json,re
for f in ("r03_mx_bench","r03_stress10k_bench","r03_textured_bench","r03_node_bench"):
    d=json.loads(open(f"profiles/{f}.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], r.get("kernel"), r.get("kernel_ms_avg"), "frac", r.get("frac"), "bound_ms", r.get("bound_ms_per_launch"), "peak", r.get("peak"), "ach", r.get("achieved"), "vsvalu", r.get("vs_fp32_valu_peak"), "meas", r.get("peak_fp32_valu_measured"))
    for k in ("valu_broad","frames_in_flight_3","accel_bvh","f64_parity_mode","cpu_baseline","cpu_baseline_c1","multi_gpu"):
        if k in d and d[k]: print("  ",k, {a:b for a,b in d[k].items() if a in("value","kernel_ms","frac","frac_of_measured_peak","ms_per_frame","cores","gpu_same_frame","gather_ms","band_kernel_ms_per_device","call_ms")})
for t in ("mx","valu","stress10k","stress10k_valu","textured"):
    s=open(f"profiles/r03_{t}_sq_pmc.txt").read()
    print(t, re.findall(r"clock ~.*|MFMA wave-instructions.*|avg waves.*", s))
    print("   ", open(f"profiles/r03_{t}_phase_split.txt").read().splitlines()[:1])
PY
