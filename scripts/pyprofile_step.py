"""cProfile of one bench step (host-side overheads): python scripts/pyprofile_step.py"""
import cProfile
import pstats
import sys

sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--cpu-baseline", "off", "--end-to-end", "off", "--global-probe", "off"]
sys.path.insert(0, ".")
import bench  # noqa: E402

pr = cProfile.Profile()
_orig = bench.time.perf_counter
started = {"n": 0}
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
