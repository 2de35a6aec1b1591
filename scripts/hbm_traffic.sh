#!/bin/bash
# HBM-side traffic per kernel of one bench step, from the L2's memory-side request counters, in a --pmc pass of its own
# (never combined with sys/hip traces).  Use the RAW counters: the derived FETCH_SIZE/WRITE_SIZE did not finish within
# 40 minutes on this pool.  Output: gpurun_out/<tag>_hbm_traffic.txt and .json (bytes per launch of each kernel).
# Bytes (round 5): the expression rocprofv3 itself lists for FETCH_SIZE on gfx950 (`rocprofv3 -L`):
#   read  = TCC_BUBBLE x 128 B + (TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B) x 64 B + TCC_EA0_RDREQ_32B x 32 B
#   write = TCC_EA0_WRREQ_64B x 64 B + (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B) x 32 B
# Rounds 1-4 took RDREQ x 64 B at face value; that halves the reads of kernels whose waves read 16 B per lane from
# consecutive addresses (128-B requests, counted in TCC_BUBBLE): the extrema sweep reads its seven planes, 1.24 GB per 4K
# view, and was reported at 0.73 GB (profiles/r05c_extrema_wave.txt).  The three TCC read counters and the two write
# counters do not fit one pass (4 TCC slots), so reads and writes are collected in separate passes.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rm -rf /tmp/pt /tmp/pw
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d /tmp/pt -o p -- \
    python3 bench.py --cpu-baseline off --end-to-end off --global-probe off --with-gain off --pipeline off --steps 1 --warmup 0 > /tmp/pt.json 2> /tmp/pt.err
timeout 900 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d /tmp/pw -o p -- \
    python3 bench.py --cpu-baseline off --end-to-end off --global-probe off --with-gain off --pipeline off --steps 1 --warmup 0 > /tmp/pw.json 2> /tmp/pw.err
python3 - "$TAG" <<'PY'
import csv, collections, json, os, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for d in ("/tmp/pt", "/tmp/pw"):
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_EA0_RDREQ_sum": n[k] += 1
def rd_bytes(v):
    return v["TCC_BUBBLE_sum"] * 128 + (v["TCC_EA0_RDREQ_sum"] - v["TCC_BUBBLE_sum"] - v["TCC_EA0_RDREQ_32B_sum"]) * 64 + v["TCC_EA0_RDREQ_32B_sum"] * 32
def wr_bytes(v):
    return v["TCC_EA0_WRREQ_64B_sum"] * 64 + (v["TCC_EA0_WRREQ_sum"] - v["TCC_EA0_WRREQ_64B_sum"]) * 32
rows = sorted(acc.items(), key=lambda kv: -(rd_bytes(kv[1]) + wr_bytes(kv[1])))
out = ["# rocprofv3 --pmc, two passes of bench.py --steps 1 --warmup 0 (ONE step of 64 x 4K): TCC_EA0_RDREQ / TCC_BUBBLE / TCC_EA0_RDREQ_32B, then TCC_EA0_WRREQ / _64B",
       "# read = BUBBLE x 128 + (RDREQ - BUBBLE - RDREQ_32B) x 64 + RDREQ_32B x 32 B; write = WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32 B (rocprofv3's own FETCH_SIZE / WRITE_SIZE terms)",
       "%-72s %8s %10s %10s %14s %10s" % ("kernel", "launches", "read GB", "write GB", "GB per launch", "128B share")]
sys.path.insert(0, ".")
import bench
js = {"_meta": {"csrc_sha256": bench.csrc_sha256(), "env": {k: v for k, v in os.environ.items() if k.startswith("APS_")},
                "how": "scripts/hbm_traffic.sh: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum | TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum, one bench step each"}}
for k, v in rows:
    rd, wr = rd_bytes(v) / 1e9, wr_bytes(v) / 1e9
    js[k] = {"launches_per_step": n[k], "read_bytes_per_step": rd * 1e9, "write_bytes_per_step": wr * 1e9}
    if rd + wr > 0.05:
        out.append("%-72s %8d %10.3f %10.3f %14.5f %10.2f" % (k[:72], n[k], rd, wr, (rd + wr) / max(n[k], 1), v["TCC_BUBBLE_sum"] * 128 / max(rd * 1e9, 1)))
open(f"gpurun_out/{tag}_hbm_traffic.txt", "w").write("\n".join(out) + "\n")
json.dump(js, open(f"gpurun_out/{tag}_hbm_traffic.json", "w"), indent=0)
print("\n".join(out[:14]))
PY

# Last pass (its own --pmc run): vector instructions per kernel, for the entries of the bench line that are bound by vector
# issue rather than by bytes (bench.py: valu_frac = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz / kernel time).
rm -rf /tmp/pv
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/pv -o p -- \
    python3 bench.py --cpu-baseline off --end-to-end off --global-probe off --with-gain off --pipeline off --steps 1 --warmup 0 > /tmp/pv.json 2> /tmp/pv.err
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
import os
js = {"_meta": {"csrc_sha256": bench.csrc_sha256(), "env": {k: v for k, v in os.environ.items() if k.startswith("APS_")}, "how": "scripts/hbm_traffic.sh, last pass: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES, one bench step"}}
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
