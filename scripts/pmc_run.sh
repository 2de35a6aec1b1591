#!/bin/bash
# usage: scripts/pmc_run.sh <kernel-substring> <out.txt> -- <python script args...>
# Runs separate rocprofv3 --pmc passes (never combined with sys/hip traces) and prints per-kernel counter sums.
KERN="$1"; OUT="$2"; shift 2; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT" ;; esac   # (the passes run from /tmp: a relative path is meant from the repo root)
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
i=0
for line in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
            ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  ( cd "$ROOT" && rocprofv3 --pmc $line --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python3 "$@" > /tmp/pmc_$i.log 2>&1 )
  python3 - "$KERN" /tmp/pmc_$i >> "$OUT" <<'PY'
import csv, glob, sys, collections
kern, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(f"{k:36s} {acc[k]/max(n[k],1):.6g}  (per dispatch, {n[k]} dispatches)")
PY
done
cat "$OUT"
