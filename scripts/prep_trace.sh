#!/bin/bash
# kernel durations of the matcher's preparation (scripts/probe/probe_prep.py) from a rocprofv3 kernel trace, last call only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pp
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 scripts/probe/probe_prep.py $1 > $R/gpurun_out/prep_trace.txt 2>&1) || { tail -5 $R/gpurun_out/prep_trace.txt; exit 1; }
grep "^call" $R/gpurun_out/prep_trace.txt
python3 - <<'PY' | tee -a $R/gpurun_out/prep_trace.txt
import csv
rows = sorted(csv.DictReader(open("/tmp/pp/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last absmax launch on
last = max(i for i, r in enumerate(rows) if "absmax" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]} wg {r["Workgroup_Size_X"]} vgpr {r.get("VGPR_Count","?")} lds {r.get("LDS_Block_Size","?")}  {r["Kernel_Name"].split("(")[0][:70]}')
PY
