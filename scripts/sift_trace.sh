#!/bin/bash
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/st
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o p -- python3 scripts/probe/probe_sift_trace.py > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("/tmp/st/p_kernel_trace.csv")))
# last third = the third extraction
rows=[r for r in rows if "synth" not in r["Kernel_Name"]]
n=len(rows)//3
rows=rows[2*n:]
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for r in rows:
    k=r["Kernel_Name"].split("(")[0]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    tot[k]+=d; cnt[k]+=1
print("per view (us):")
for k,v in sorted(tot.items(), key=lambda kv:-kv[1])[:14]: print(f"{v:9.1f} us {cnt[k]:4d} launches  {k[:80]}")
print("total", sum(tot.values()))
print("largest dispatches:")
big=sorted(rows, key=lambda r:-(int(r["End_Timestamp"])-int(r["Start_Timestamp"])))[:24]
for r in big:
    print(f'{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us grid {r.get("Grid_Size_X","?")}x{r.get("Grid_Size_Y","?")} wg {r.get("Workgroup_Size_X","?")} lds {r.get("LDS_Block_Size","?")} vgpr {r.get("VGPR_Count","?")} {r["Kernel_Name"].split("(")[0][:60]}')
t0=int(rows[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in rows)
print("span of the extraction (us):", (t1-t0)/1e3)
PY
