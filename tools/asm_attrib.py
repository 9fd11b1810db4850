#!/usr/bin/env python3
"""Static instruction counts of one kernel per source function, from an assembly listing compiled with -gline-tables-only
(`.loc file line` directives attribute every instruction to the source line it was inlined from).
usage: tools/asm_attrib.py LISTING.s KERNEL_SUBSTRING raytracer_amd/csrc/mirt_kernels.h raytracer_amd/csrc/mirt_device.h"""
import re
import sys
from collections import Counter

lst, kern, kfile, dfile = sys.argv[1:5]


def functions(path):
    """(first line, last line, name) of every function-like definition, by a crude scan of 'name(' at low indentation."""
    out, name, start = [], None, 0
    for n, line in enumerate(open(path), 1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:__host__ |__device__ |__global__ |static |inline |__launch_bounds__\([^)]*\) )*[\w:<>\*& ,]*?\b(\w+)\s*\([^;]*$", line)
        if m and not line.startswith((" ", "\t", "#", "/", "*")) and m.group(1) not in ("if", "for", "while", "switch", "defined"):
            if name:
                out.append((start, n - 1, name))
            name, start = m.group(1), n
    if name:
        out.append((start, 10 ** 9, name))
    return out


fmap = {1: functions(kfile), 2: functions(dfile)}


def func_of(fid, line):
    for a, b, nm in fmap.get(fid, []):
        if a <= line <= b:
            return nm
    return {3: "ocml(math.h)", 4: "warp_functions", 5: "device_functions", 6: "hip_runtime", 7: "hip_atomic"}.get(fid, f"file{fid}")


inside, cur = False, (0, 0)
valu, other, quarter = Counter(), Counter(), Counter()
QUARTER = ("v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag")
lines_by = Counter()
for line in open(lst):
    if not inside:
        if line.startswith("_ZN") and kern in line and line.rstrip().endswith(tuple(":")) or (line.startswith("_ZN") and kern in line and ":" in line):
            inside = True
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.strip().startswith("s_endpgm") and False:
        break
    if re.match(r"^\.Lfunc_end", line):
        break
    m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)", line)
    if not m:
        continue
    op = m.group(1)
    f = func_of(*cur)
    if op.startswith("v_"):
        valu[f] += 1
        if op.startswith(QUARTER):
            quarter[f] += 1
        lines_by[(f, cur[1])] += 1
    else:
        other[(f, op.split("_")[0] if not op.startswith("s_waitcnt") else "s_waitcnt")] += 1
tot = sum(valu.values())
print(f"kernel {kern}: {tot} VALU instructions (static, both scene-residency copies of the loop)")
for f, n in valu.most_common():
    o = {k[1]: v for k, v in other.items() if k[0] == f}
    print(f"  {f:28s} VALU {n:5d} ({100.0 * n / tot:4.1f} %)  quarter-rate {quarter[f]:4d}   other: " + " ".join(f"{k}={v}" for k, v in sorted(o.items())))
if "--lines" in sys.argv:
    for (f, ln), n in sorted(lines_by.items(), key=lambda x: -x[1])[:60]:
        print(f"    {f}:{ln}  {n}")
