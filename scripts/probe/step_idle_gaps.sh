#!/bin/bash
# kernel trace of a short bench run; for the LAST step: total time no kernel is running, and the idle gaps above 40 us with the
# kernels on either side (where a step still waits on the host).  usage: bash scripts/probe/step_idle_gaps.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/bg
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/bg -o p -- python3 bench.py --steps 3 --warmup 1 --end-to-end off --global-probe off --cpu-baseline off --with-gain off > /tmp/bg_line.txt 2>&1) || { tail -5 /tmp/bg_line.txt; exit 1; }
python3 - <<'PY'
import csv
rows = sorted(csv.DictReader(open("/tmp/bg/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "synth" not in r["Kernel_Name"]]
# the steps are delimited by the single launch of the screening kernel: last step = from the previous screen's end + ... use extrema launches instead
scr = [i for i, r in enumerate(rows) if "match_screen_i8x16_kernel<false>" in r["Kernel_Name"]]
# a step starts with the first SIFT kernel after the previous step's crop; take the window between the last two crop_bbox launches
crop = [i for i, r in enumerate(rows) if "crop_bbox" in r["Kernel_Name"]]
a, b = crop[-2] + 1, crop[-1] + 1
win = rows[a:b]
t0, t1 = int(win[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in win)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]) for r in win)
idle, cur_end, last = 0, ev[0][0], ev[0][2]
gaps = []
for s, e, nme in ev:
    if s > cur_end:
        idle += s - cur_end
        if s - cur_end > 40000: gaps.append(((cur_end - t0) / 1e6, (s - cur_end) / 1e3, last, nme))
    if e > cur_end:
        cur_end, last = e, nme
print(f"last step: {(t1 - t0) / 1e6:.2f} ms from its first kernel to its last, {idle / 1e6:.2f} ms with no kernel running, {len(win)} launches")
for at, g, p, n in gaps:
    print(f"  at {at:8.2f} ms: {g:7.1f} us idle between {p} and {n}")
PY
