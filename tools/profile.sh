#!/bin/bash
# One-stop profile of bench.py on the GPU box: kernel-trace stats (the driver's command: 20 steps after 5 warm-up steps, so that the few
# slower launches on a cold chip weigh as little in the average as in bench.py's own), SQ counters (3 passes), HBM bytes (2 passes).
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv,sq_pmc.txt,hbm.txt}
tag=$1; shift
BARGS="$*"
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_$tag; mkdir -p $out
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $BARGS > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cp $out/kt/k_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
tools/pmc.sh $tag $BARGS > $out/sq_pmc.txt 2>&1 || { tail -5 $out/sq_pmc.txt; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $out/hbm_$c -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BARGS > $out/hbm_$c.log 2>&1 || { tail -5 $out/hbm_$c.log; exit 1; }
done
python3 - $out <<'PY' > $out/hbm.txt
import csv, glob, sys
out = sys.argv[1]
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1; raw unit KiB.")
print("# gfx950 note (MI355X_MICROARCH.md): FETCH_SIZE under-reports wide coalesced reads by 2x; the raw value is right for 4-byte strided reads.")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/hbm_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "fma_peak_kernel" not in r["Kernel_Name"]:  # the bench's FMA-peak probe: 41 launches, no traffic
                v = float(r["Counter_Value"])
                print(f"{r['Kernel_Name'][:60]:60s} {c:10s} raw {v:12.1f} KiB = {v/1024:9.2f} MiB (x2 if wide reads: {v/512:9.2f} MiB)")
PY
cat $out/kernel_stats.csv | cut -c1-200; tail -25 $out/sq_pmc.txt; cat $out/hbm.txt
