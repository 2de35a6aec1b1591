"""cProfile of bench steps (host-side overheads): python scripts/probe/pyprofile_step.py - the package's own functions by cumulative
time per step, then everything by internal time."""
import cProfile
import pstats
import sys

STEPS = 8
sys.argv = ["bench.py", "--steps", str(STEPS - 2), "--warmup", "2", "--cpu-baseline", "off", "--end-to-end", "off", "--global-probe", "off"]
sys.path.insert(0, ".")
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if "autopanostitch-matlab_amd" in fn and nc >= STEPS - 2:
        rows.append((ct / STEPS * 1e3, tt / STEPS * 1e3, nc, fn.rsplit("/", 1)[-1], line, name))
print("package functions, ms per step (cumulative, own), calls:")
for ct, tt, nc, fn, line, name in sorted(rows, reverse=True)[:40]:
    print(f"  {ct:8.3f} {tt:8.3f} {nc:6d}  {fn}:{line}({name})")
st.sort_stats("tottime").print_stats(16)
