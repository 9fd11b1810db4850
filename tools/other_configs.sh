for cfg in "--workload textured --width 3840 --spp 256" "--workload stress31623 --width 3840 --spp 16" "--workload stress31623 --width 3840 --spp 16 --accel bvh" "--workload stress10000 --width 3840 --spp 16 --accel bvh" "--workload mars --width 400 --spp 100"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 $cfg | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'], '|', d['roofline']['kernel'], d['value'],'Mrays/s', d['ms_per_step'],'ms frac', d['roofline']['frac'], d['diagnostics']['primary_samples_per_s'])" || exit 1
done
