"""How well does a dense f16-MFMA kernel with few registers co-run with SIFT?  (APS_DBG build)
Prints: SIFT alone, co-runner alone (ms per launch), both together (SIFT time, co-runner launches completed meanwhile)."""
import sys, ctypes, threading, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
lib = apsamd._capi.lib
lib.aps_dbg_corun.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
imgs, cams = synth.make_scene(8, 4, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NWG, SPIN = 8192, 2000
def sift():
    t0 = time.perf_counter(); pl.sift_many(inp, imgs); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
def corun():
    t0 = time.perf_counter(); apsamd._capi.check(lib.aps_dbg_corun(mode, NWG, SPIN)); return 1e3 * (time.perf_counter() - t0)
sift(); corun()
ts = [sift() for _ in range(3)]; tc = [corun() for _ in range(3)]
print(f"SIFT alone {min(ts):.1f} ms (32 views); co-runner alone {min(tc):.2f} ms per launch", flush=True)
for rep in range(3):
    done = []
    stop = False
    def bg():
        while not stop:
            done.append(corun())
    th = threading.Thread(target=bg); th.start()
    t = sift()
    stop = True; th.join()
    work = (len(done) - 1 + 0.5) * min(tc)  # co-runner work finished during the SIFT window, in stand-alone milliseconds
    print(f"together: SIFT {t:.1f} ms, co-runner launches {len(done)} (~{work:.1f} ms of stand-alone MFMA work inside the window)", flush=True)
