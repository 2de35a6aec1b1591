#!/bin/bash
# Produces the judged artefacts of a round on the GPU box (run through gpurun):
#   gpurun_out/<tag>_bench.json                 the bench line (un-profiled run)
#   gpurun_out/<tag>_bench_under_rocprofv3.json the bench line of the --kernel-trace --stats run
#   gpurun_out/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats summary of that same command
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
# a fresh box spends its first ten seconds or so on page-ins and allocator growth (the host side of a step is ~15 ms slower
# there): one throw-away run first, so that the judged runs measure the steady state the driver's 25-step run also reaches
python3 bench.py --steps 3 --warmup 1 --cpu-baseline off --end-to-end off --with-gain off --global-probe off > /dev/null 2>&1
python3 bench.py --steps 3 --warmup 1 --cpu-baseline off --end-to-end off --with-gain off --global-probe off > /dev/null 2>&1
python3 bench.py --steps 6 --warmup 2 > gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || { tail -5 /tmp/bench.err >&2; exit 1; }
# ONE profiled run, no retry (a GPU step that failed is not repeated in the same call): a missing or empty summary
# fails the script, so a broken profile never becomes a judged artefact.  (Round 2 saw rocprofv3 die with a SIGSEGV in
# the runtime's launch path when the SIFT pool's threads issued their first launches at once; the pool now brings its
# workers' contexts up one after the other - pipeline._init_worker.)
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python3 bench.py --cpu-baseline off --steps 6 --warmup 2 \
    > gpurun_out/${TAG}_bench_under_rocprofv3.json 2> /tmp/prof_stats.err
rc=$?
if [ $rc -ne 0 ] || [ ! -s /tmp/prof_stats/p_kernel_stats.csv ]; then
    echo "rocprofv3 run failed (rc=$rc) or produced no kernel summary" >&2
    tail -20 /tmp/prof_stats.err >&2
    exit 1
fi
cp /tmp/prof_stats/p_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
# HBM traffic (FETCH_SIZE / WRITE_SIZE): NOT collected here.  Round 1 tried twice (on bench.py and on the small matching
# probe, each in its own --pmc pass): the TCC-derived counters did not finish within 40 and 10 minutes on this pool, while
# SQ_* counters on the same probe take seconds (scripts/pmc_run.sh).  bench.py therefore reports "traffic": null.
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print(d["value"], d["unit"], d["ms_per_step"], d["roofline"], d["cpu_baseline"])
PY
