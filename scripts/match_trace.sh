cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/mp
rocprofv3 --kernel-trace --output-format csv -d /tmp/mp -o p -- python3 scripts/probe/probe_match.py 8 19800 > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("/tmp/mp/p_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:14]:
    v=sorted(v); print(f"{k[:60]:60s} n={len(v):4d} min={v[0]:8.1f} med={v[len(v)//2]:8.1f} max={v[-1]:9.1f} us")
PY
