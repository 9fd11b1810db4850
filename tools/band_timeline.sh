#!/bin/bash
# Kernel timeline of one rank's band of an 8-way split, frames in flight (tools/inflight_probe.py under rocprofv3 --kernel-trace).
root=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp && cd $root
export PYTHONPATH=$root
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/band_tl -o t -- python3 tools/inflight_probe.py ${1:-8} ${2:-12} > gpurun_out/band_tl.log 2>&1 || { tail -5 gpurun_out/band_tl.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/band_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"] or "resolve_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# print a window in the middle of the 3-in-flight phase of the first band
n = len(rows)
for r in rows[n // 6: n // 6 + 24]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{r['Kernel_Name'][:28]:28s} start {s/1e3:10.1f} us  end {e/1e3:10.1f} us  dur {(e-s)/1e3:8.1f} us  queue {r.get('Queue_Id','?')}")
PY
