#!/bin/bash
# HBM-side traffic per kernel of one bench step, from the L2's memory-side request counters, in a --pmc pass of its own
# (never combined with sys/hip traces).  Use the RAW counters: the derived FETCH_SIZE/WRITE_SIZE did not finish within
# 40 minutes on this pool.  Output: gpurun_out/<tag>_hbm_traffic.txt and .json (bytes per launch of each kernel).
# Calibration on this workload: blur_kernel reads one plane and writes one plane per launch and reports
# RDREQ*64 B == WRREQ*64 B == the plane bytes; extrema_kernel reports exactly its 7 planes.  So requests x 64 B is
# taken at face value here (the 2x correction MI355X_MICROARCH.md gives for FETCH_SIZE on wide reads would double-count).
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rm -rf /tmp/pt
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d /tmp/pt -o p -- \
    python3 bench.py --cpu-baseline off --end-to-end off --global-probe off --steps 1 --warmup 0 > /tmp/pt.json 2> /tmp/pt.err
python3 - "$TAG" <<'PY'
import csv, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open("/tmp/pt/p_counter_collection.csv")):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_EA0_RDREQ_sum": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -(kv[1]["TCC_EA0_RDREQ_sum"] + kv[1]["TCC_EA0_WRREQ_sum"]))
out = ["# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum (own pass), bench.py --steps 1 --warmup 0: ONE step of 64 x 4K",
       "# bytes = requests x 64 B (calibrated on blur_kernel/extrema_kernel, see scripts/hbm_traffic.sh)",
       "%-72s %8s %10s %10s %14s" % ("kernel", "launches", "read GB", "write GB", "GB per launch")]
sys.path.insert(0, ".")
import bench
js = {"_meta": {"csrc_sha256": bench.csrc_sha256(), "how": "scripts/hbm_traffic.sh: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum, one bench step"}}
for k, v in rows:
    rd, wr = v["TCC_EA0_RDREQ_sum"] * 64 / 1e9, v["TCC_EA0_WRREQ_sum"] * 64 / 1e9
    js[k] = {"launches_per_step": n[k], "read_bytes_per_step": rd * 1e9, "write_bytes_per_step": wr * 1e9}
    if rd + wr > 0.05: out.append("%-72s %8d %10.3f %10.3f %14.5f" % (k[:72], n[k], rd, wr, (rd + wr) / max(n[k], 1)))
open(f"gpurun_out/{tag}_hbm_traffic.txt", "w").write("\n".join(out) + "\n")
json.dump(js, open(f"gpurun_out/{tag}_hbm_traffic.json", "w"), indent=0)
print("\n".join(out[:14]))
PY

# Second pass (its own --pmc run): vector instructions per kernel, for the entries of the bench line that are bound by vector
# issue rather than by bytes (bench.py: valu_frac = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz / kernel time).
rm -rf /tmp/pv
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/pv -o p -- \
    python3 bench.py --cpu-baseline off --end-to-end off --global-probe off --steps 1 --warmup 0 > /tmp/pv.json 2> /tmp/pv.err
python3 - "$TAG" <<'PY'
import csv, collections, json, sys
sys.path.insert(0, ".")
import bench
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open("/tmp/pv/p_counter_collection.csv")):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
js = {"_meta": {"csrc_sha256": bench.csrc_sha256(), "how": "scripts/hbm_traffic.sh, second pass: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES, one bench step"}}
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])
out = ["%-72s %8s %14s %14s %12s" % ("kernel", "launches", "INSTS_VALU", "INSTS_SALU", "VALU/wave")]
for k, v in rows:
    js[k] = {"launches_per_step": n[k], "insts_valu_per_step": v["SQ_INSTS_VALU"], "insts_salu_per_step": v["SQ_INSTS_SALU"], "waves_per_step": v["SQ_WAVES"]}
    if v["SQ_INSTS_VALU"] > 1e7:
        out.append("%-72s %8d %14.4g %14.4g %12.1f" % (k[:72], n[k], v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"], v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)))
open(f"gpurun_out/{tag}_pmc_valu.txt", "w").write("\n".join(out) + "\n")
json.dump(js, open(f"gpurun_out/{tag}_pmc_valu.json", "w"), indent=0)
print("\n".join(out[:12]))
PY
