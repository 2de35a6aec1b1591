"""Round 6 experiment: how does the screening kernel's time depend on the MAGNITUDES of its int8 operands?  (The chip is
power-limited under a dense int8 MFMA stream: 1.98 GHz on random operands, 2.39 GHz on zeros - scripts/probe/mfma_i8_shapes.hip.)
Needs the timing build (APS_LIB_PATH=.../libaps_hip_timing.so): APS_Q8_CENTER=c writes the exact codes as clamp(u - c), which
gives WRONG match lists for c != 128 - only the kernel's time is read here.  Also prints the floor of the survivor share: the
rows that pass the ratio / threshold filter (Unique = False)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs)]
order = fm.pair_order(len(imgs))
d0 = descs[0][:2000].cpu().numpy()
t = 1.0 / np.array([r[r > 0].min() for r in d0])[:, None]
uu = np.rint(d0 * t)
print("u percentiles 50/90/99/99.9/max:", [float(np.percentile(uu, q)) for q in (50, 90, 99, 99.9, 100)], "share > 127: %.5f" % (uu > 127).mean(),
      "rows with some u > 127: %.4f" % (uu > 127).any(1).mean(), "rows with some u > 191: %.4f" % (uu > 191).any(1).mean(), flush=True)

def run(tag, reps=3):
    best = None
    for r in range(reps):
        capi.profile_enable(1); capi.profile_reset()
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True)
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        prof = capi.profile_all()
        scr = prof["match_screen_i8"][0]
        best = scr if best is None else min(best, scr)
    import ctypes
    rows, surv = ctypes.c_int64(), ctypes.c_int64()
    capi.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv))
    print(f"{tag:>16}: screen {best:.2f} ms (last call: wall {wall:.2f}, cand {prof.get('match_cand_f16', (0.0, 0))[0]:.2f} list {prof.get('match_list_i8', (0.0, 0))[0]:.2f} rescore {prof.get('match_rescore', (0.0, 0))[0]:.2f}, prep {prof['match_prep'][0]:.2f}), survivors {100.0 * surv.value / max(rows.value, 1):.2f} %", flush=True)

run("exact c=128")
os.environ["APS_MATCH_NO_EXACT"] = "1"
run("general codes")
del os.environ["APS_MATCH_NO_EXACT"]
if "timing" in os.environ.get("APS_LIB_PATH", ""):
    for c in (96, 64, 48, 32, 16, -1, 128):
        os.environ["APS_Q8_CENTER"] = str(c)
        run(f"clamp(u - {max(c, 0)})")
    del os.environ["APS_Q8_CENTER"]
out = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], False, device_out=True)
n_rows = sum(int(descs[i].shape[0]) * (len(descs) - 1 - i) for i in range(len(descs)))
print("rows that pass ratio + threshold (Unique = False): %d of %d = %.3f %%" % (int(out[0][-1]), n_rows, 100.0 * int(out[0][-1]) / n_rows))
