#!/bin/bash
# kernel trace of two bench steps; prints what runs around a named kernel of the last step.  usage: bench_trace_window.sh <kernel substring> [before_us] [after_us]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/bt
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o p -- python3 bench.py --steps 2 --warmup 1 --end-to-end off --global-probe off --cpu-baseline off > $R/gpurun_out/bench_trace_line.txt 2>&1) || { tail -5 $R/gpurun_out/bench_trace_line.txt; exit 1; }
python3 - "$1" "${2:-300}" "${3:-3000}" <<'PY' | tee $R/gpurun_out/bench_trace_window.txt
import csv, sys
name, before, after = sys.argv[1], float(sys.argv[2]) * 1e3, float(sys.argv[3]) * 1e3
rows = sorted(csv.DictReader(open("/tmp/bt/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
t0 = int(rows[hits[-1]]["Start_Timestamp"])
import collections
busy = collections.Counter()
for r in rows:  # kernels per millisecond in the 20 ms before the named kernel (start times), copy kernels apart
    s = int(r["Start_Timestamp"])
    if t0 - 20e6 <= s < t0:
        busy[(int((s - t0) // 1e6), "copy/fill" if "rocclr" in r["Kernel_Name"] else "kernel")] += 1
print("launches per ms before it:", ", ".join(f"{k[0]} ms {k[1]}: {v}" for k, v in sorted(busy.items())))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e >= t0 - before and s <= t0 + after:
        print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q {r.get("Queue_Id","?"):>3s} grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]} wg {r["Workgroup_Size_X"]}  {r["Kernel_Name"].split("(")[0][:60]}')
PY
