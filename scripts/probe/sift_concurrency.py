"""Reads a rocprofv3 --kernel-trace csv of bench.py and reports, for the LAST feature stage, how many kernels were in
flight over time (time-weighted histogram), the share of the window with no kernel running, and the largest idle gaps.
usage: python3 scripts/probe/sift_concurrency.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
screens = [e for e in ev if "match_screen_i8" in e[2]]
end = screens[-1][0]                      # the last screen kernel starts after the last feature stage
prev_end = max(e[1] for e in ev if e[1] < end and ("rw_" in e[2] or "crop" in e[2] or "mb_" in e[2]) ) if len(screens) > 1 else ev[0][0]
win = [e for e in ev if e[0] >= prev_end and e[1] <= end and "aps::" in e[2] or ("rocprim" in e[2] and e[0] >= prev_end and e[1] <= end)]
t0, t1 = min(e[0] for e in win), max(e[1] for e in win)
pts = sorted([(s, 1) for s, e, _ in win] + [(e, -1) for s, e, _ in win])
hist, cur, last, gaps = {}, 0, t0, []
for t, d in pts:
    if t > last:
        hist[cur] = hist.get(cur, 0) + (t - last)
        if cur == 0:
            gaps.append((t - last, last - t0))
    cur += d
    last = t
tot = t1 - t0
print("window %.2f ms, %d kernels" % (tot / 1e6, len(win)))
for k in sorted(hist):
    print("  %2d in flight: %5.1f %%" % (k, 100.0 * hist[k] / tot))
print("mean in flight %.2f" % (sum(k * v for k, v in hist.items()) / tot))
print("largest idle gaps (us @ ms):", [(round(g / 1e3, 1), round(at / 1e6, 2)) for g, at in sorted(gaps, reverse=True)[:8]])
