"""What ONE rank of eight pays per step on the 64 x 4K job, measured on one GPU: its 8 views through SIFT
(by worker count), its share of the pairs (1/8 of the 2016 by weight) through the matcher, its share of the candidate pairs
through RANSAC, the replicated host part, its 1/8 of the tiles through the renderer.  The exchanges themselves need the node.
usage: probe_rank_costs.py [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")
fm = import_module(apsamd.__name__ + ".featureMatching")
im = import_module(apsamd.__name__ + ".imageMatching")
rp = import_module(apsamd.__name__ + ".renderPanorama")
capi = apsamd._capi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W, H, f, world = 3840, 2160, 8000.0, 8
imgs, cams = synth.make_scene(8, 8, W, H, f, 0.4, device="cuda", finest_px=16.0)
n = len(imgs)
inp = pl.default_input(bands=5)
Ks = [c["K"] for c in cams]


def sync():
    capi.check(capi.lib.aps_synchronize())
    torch.cuda.synchronize()


def timed(fn, reps=reps):
    fn()
    sync()
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        out = fn()
        sync()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), 1e3 * float(np.median(ts)), out


mine = [i for i in range(n) if i % world == 0]
for workers in (4, 6, 8, 10):
    lo, med, _ = timed(lambda: pl.sift_many(inp, [imgs[i] for i in mine], workers=workers))
    print(f"SIFT of this rank's {len(mine)} views, {workers} workers: min {lo:.2f} ms, median {med:.2f} ms ({lo / len(mine):.2f} ms per view)")
lo, med, feats = timed(lambda: pl.sift_many(inp, imgs), reps=2)
print(f"SIFT of all {n} views (one rank of one): min {lo:.2f} ms ({lo / n:.2f} ms per view)")
descs = [d for d, _ in feats]
kps = [torch.from_numpy(p).cuda() for _, p in feats]
counts = [int(d.shape[0]) for d in descs]
order = fm.pair_order(n)
wts = [float(counts[i]) * float(counts[j]) for (i, j) in order]
pown = par.partition_pairs_blocked(order, wts, n, world)
my = [p for p in range(len(order)) if pown[p] == 0]
lo, med, out = timed(lambda: fm.match_pairs_csr(descs, [order[p] for p in my], inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True))
capi.profile_enable(1)
capi.profile_reset()
fm.match_pairs_csr(descs, [order[p] for p in my], inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True)
sync()
print("   kernels of one call: " + ", ".join(f"{k} {v[0]:.3f}" for k, v in capi.profile_all().items() if v[0] > 0.01))
capi.profile_enable(False)
full = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=False)
nm_all = np.diff(np.asarray(full[0]))
for scatter in (False, True):
    ow = par.partition_pairs_blocked(order, wts, n, world, scatter=scatter)
    line = []
    for r in range(world):
        sel = [order[p] for p in range(len(order)) if ow[p] == r]
        lo_r, _, _ = timed(lambda: fm.match_pairs_csr(descs, sel, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True), reps=3)
        line.append((lo_r, len(sel), len({v for q in sel for v in q}), int((nm_all[ow == r] >= 20).sum())))
    print(f"matching per rank, groups {'scattered' if scatter else 'by index range'}: " +
          ", ".join(f"{t:.2f} ms ({k} pairs, {s_} sets, {o} overlapping)" for t, k, s_, o in line) + f"; max {max(t for t, *_ in line):.2f} ms")
print(f"matching of this rank's {len(my)} pairs (touching {len({v for p in my for v in order[p]})} descriptor sets): min {lo:.2f} ms, median {med:.2f} ms")
lo, med, out_all = timed(lambda: fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True), reps=2)
print(f"matching of all {len(order)} pairs: min {lo:.2f} ms")
pp, ia_d, ib_d, _ = out_all
n_match = np.diff(pp)
oi, oj = par._pair_index_arrays(n)
put = np.zeros((n, n), np.int64)
put[oi, oj] = n_match
sym = put + put.T
srt = np.argsort(-sym, axis=1, kind="stable")[:, : min(int(inp["mBrownLowe"]), n - 1)]
cand = np.zeros((n, n), bool)
cand[np.repeat(np.arange(n), srt.shape[1]), srt.reshape(-1)] = True
cand = np.triu(cand | cand.T, 1)
cj, ci = np.nonzero(cand.T)
pw = cj * (cj - 1) // 2 + ci
work = pw[n_match[pw] >= 4].tolist()
gpos = pp[:-1].astype(np.int64)


def ransac(sel):
    cnts = [int(n_match[p]) for p in sel]
    wptr = np.concatenate([[0], np.cumsum(cnts)]).astype(np.int64)
    dst, src = im.gather_match_points(kps, ia_d, ib_d, gpos[sel], wptr, [order[p][0] for p in sel], [order[p][1] for p in sel])
    return im.ransac_batch_drawn(src, dst, wptr, cnts, inp, 0, keys=sel)


for label, sel in (("this rank's", work[0::world]), ("all", work)):
    capi.profile_enable(1)
    capi.profile_reset()
    lo, med, res = timed(lambda: ransac(sel))
    prof = capi.profile_all()
    capi.profile_enable(False)
    print(f"RANSAC (gather + batch) of {label} {len(sel)} candidate pairs: min {lo:.2f} ms, median {med:.2f} ms; kernels per call: " +
          ", ".join(f"{k} {v[0] / (reps + 1):.3f}" for k, v in prof.items() if k.startswith("ransac") or k.startswith("gather")))
models, mask, found, ninl = res
pairs, models_l, num_matches = [], [], np.zeros((n, n))
for k, p in enumerate(work):
    if found[k] and ninl[k] > 8 + 0.3 * n_match[p]:
        i, j = order[p]
        pairs.append((i, j))
        models_l.append(models[k])
        num_matches[i, j] = ninl[k]


def host():
    ncomp, labels = pl.connected_components(num_matches)
    return pl.recognize_panoramas(n, pairs, models_l, num_matches, Ks, labels, None)


host()  # (the first call pays imports)
t0 = time.perf_counter()
for _ in range(reps):
    comps = host()
print(f"host graph + cameras (replicated on every rank): {1e3 * (time.perf_counter() - t0) / reps:.2f} ms, {len(pairs)} verified pairs")
comp = max(comps, key=lambda c: len(c["members"]))
cams_e = comp["cameras"]
sizes = [(H, W, 3)] * n
opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": False}
geo = None
for label, sub in (("this rank's 1/8 of the tiles", (0, world)), ("all tiles", None)):
    lo, med, _ = timed(lambda: rp.renderPanorama(inp, imgs, sizes, cams_e, "spherical", comp["ref"], opts, device_out=True, tile_subset=sub))
    print(f"render of {label}: min {lo:.2f} ms, median {med:.2f} ms")
# round 5: contiguous, area-balanced tile runs per rank (parallel.tile_ranges) against the t % world deal, every rank's share
full_, _, _, geo_ = rp.renderPanorama(inp, imgs, sizes, cams_e, "spherical", comp["ref"], opts, device_out=True, return_covered=True)
ranges = par.tile_ranges(int(geo_["H"]), int(geo_["W"]), (2048, 2048), world)
for label, subs in (("t % 8", [(r, world) for r in range(world)]), ("contiguous runs", [("range",) + ranges[r] for r in range(world)])):
    ts_ = []
    for sub in subs:
        lo, med, _ = timed(lambda: rp.renderPanorama(inp, imgs, sizes, cams_e, "spherical", comp["ref"], opts, device_out=True, tile_subset=sub), reps=3)
        ts_.append(lo)
    print(f"render per rank, tiles dealt {label}: " + ", ".join(f"{t:.2f}" for t in ts_) + f" ms; max {max(ts_):.2f} ms  (ranges {ranges if label != 't % 8' else ''})")

# round 6: the same rank's step with its pieces strung together - strictly one after the other, and with the NEXT step's extraction
# of its 8 views started when the match lists are complete (parallel.submit_features: beside RANSAC, the host part and the render of
# its tiles), as bench.py's pipelined steps run it.  The exchanges are not in these numbers.
my_pairs = [order[p] for p in my]
my_work = work[0::world]
local = {i: imgs[i] for i in mine}
rsub = ("range",) + ranges[0]


def rank_step(handle, start_next):
    if handle is None:
        handle = par.submit_features(inp, local)
    for fu in handle["futures"]:
        fu.result()
    fm.match_pairs_csr(descs, my_pairs, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True)
    nxt = par.submit_features(inp, local) if start_next else None
    ransac(my_work)
    host()
    rp.renderPanorama(inp, imgs, sizes, cams_e, "spherical", comp["ref"], opts, device_out=True, tile_subset=rsub)
    return nxt


capi.check(capi.lib.aps_set_thread_stream_priority(1))
for label, pipe in (("one after the other", False), ("next extraction started after the matching", True)):
    rank_step(None, False)
    sync()
    K = 12
    t0 = time.perf_counter()
    h_ = None
    for k in range(K):
        h_ = rank_step(h_, pipe and k + 1 < K)
    sync()
    print(f"one rank of eight, {K} steps {label}: {1e3 * (time.perf_counter() - t0) / K:.2f} ms per step")
