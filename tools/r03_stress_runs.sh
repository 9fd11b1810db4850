set -o pipefail
r() { tag=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 "$@" 2>gpurun_out/sr_$tag.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'kernel', r['kernel'], 'frac', r.get('frac'), 'vs_valu', r.get('vs_fp32_valu_peak'), 'seg', d['diagnostics']['segments_per_step'])" || { tail -3 gpurun_out/sr_$tag.err; exit 1; }; }
r s10k_d1_mx --workload stress10000 --width 3840 --spp 16 --depth 1
r s10k_d1_valu --workload stress10000 --width 3840 --spp 16 --depth 1 --accel valu
r s31k_mx --workload stress31623 --width 3840 --spp 16
r s31k_bvh --workload stress31623 --width 3840 --spp 16 --accel bvh
r s100k_bvh --workload stress100000 --width 3840 --spp 16 --accel bvh
r s100k_mx --workload stress100000 --width 3840 --spp 16 --steps 1 --warmup 0
