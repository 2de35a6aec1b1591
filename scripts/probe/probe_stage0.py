"""Stage-0 proof in front of the int8 screen: survivor rates on the bench scene's descriptors, simulated with torch
(fp32 matmuls + the int8 codes' arithmetic), before any kernel is written.

  L1 <= d1 : min_j ||P a - P b_j||^2 over ALL columns, P = orthonormal projection to k dims (PCA of a sample, optionally
             rotated inside the subspace so that the int8 codes see equal ranges), evaluated on int8 codes with the
             measured residual norms (Cauchy-Schwarz slack 2E) - one K=k int8 product instead of K=128;
  H2 >= d2 : second smallest EXACT distance over a fixed 1/s sample of the columns.
A row is dismissed when L1 > r^2 H2 (matchFeaturesScratch.m:170-178, r = 0.6).

python scripts/probe/probe_stage0.py
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
NX, NY, W, H, F, OV = 8, 8, 3840, 2160, 8000.0, 0.4
cams = synth.grid_cameras(NX, NY, W, H, F, 2 * np.arctan(W / (2 * F)) * (1 - OV), 2 * np.arctan(H / (2 * F)) * (1 - OV), 1.0, 12345)
ids = [0, 1, 2, 9, 27, 45, 63]
imgs = [synth.render_view(cams[i], H, W, 12345, "cuda", finest_px=16.0) for i in ids]
input_ = pl.default_input(bands=5)
descs, _ = pl.extract_features(input_, imgs)
descs = [torch.as_tensor(d, device="cuda").float() for d in descs]
descs = [d / (d.norm(dim=1, keepdim=True) + 1.1920929e-07) for d in descs]
del imgs
R2 = 0.36

allx = torch.cat(descs)
mu = allx.mean(0)
C = (allx - mu).T @ (allx - mu) / allx.shape[0]
ev, evec = torch.linalg.eigh(C.double())
evec = evec.flip(1).float()  # descending
ev = ev.flip(0)
print("PCA energy kept: " + " ".join(f"k={k}:{float(ev[:k].sum() / ev.sum()):.3f}" for k in (16, 24, 32, 48, 64, 96)))
print(f"mean norm^2 {float((mu * mu).sum()):.3f}  total var {float(ev.sum()):.3f}")
g = torch.Generator(device="cpu").manual_seed(1)


def basis(k, rotate):
    P = evec[:, :k]
    if rotate:
        Q, _ = torch.linalg.qr(torch.randn(k, k, generator=g).double())
        P = (P.double() @ Q.cuda()).float()
    return P


def exact_top2(A, B):
    D = (A * A).sum(1, keepdim=True) + (B * B).sum(1)[None] - 2 * A @ B.T
    v, _ = torch.topk(D, 2, dim=1, largest=False)
    return v[:, 0].clamp_min(0), v[:, 1].clamp_min(0)


def q8_lower(PA, PB):
    """L1 from int8 codes of the projected sets: per-row symmetric 7-bit code on the A side, per-set 8-bit offset code on
    the B side, the column term -||Pb_j||^2/2 carried EXACTLY (floored pieces in the real kernel), slack 2E."""
    ra = PA.abs().amax(1, keepdim=True).clamp_min(1e-30)
    qa = torch.round(PA * (127 / ra)).clamp(-127, 127)
    ea = (PA - qa * (ra / 127)).norm(dim=1)
    mx, mn = PB.max(), PB.min()
    sb = 255 / (mx - mn)
    cb = torch.round(mn * sb) + 128
    qb = (torch.round(PB * sb) - cb).clamp(-128, 127)
    eb = (PB - (qb + cb) / sb).norm(dim=1)
    dotq = (qa * (ra / 127)) @ ((qb + cb) / sb).T  # = the value the integer dot stands for
    nb2 = (PB * PB).sum(1)
    s = dotq - nb2[None] / 2
    E = ea * PB.norm(dim=1).max() + (PA.norm(dim=1) + ea) * eb.max()
    L1 = (PA * PA).sum(1) - 2 * (s.amax(1) + E)
    return L1, E


def sample_h2(A, B, step):
    Bs = B[::step]
    _, d2 = exact_top2(A, Bs)
    return d2


pairs = [(0, 1), (0, 3), (0, 2), (0, 4), (0, 6), (4, 5), (1, 6)]
for (a, b) in pairs:
    A, B = descs[a], descs[b]
    d1, d2 = exact_top2(A, B)
    passf = (d1 <= R2 * d2)
    print(f"\npair {ids[a]}-{ids[b]}: nA {A.shape[0]} nB {B.shape[0]}  d1 med {float(d1.median()):.3f} d2 med {float(d2.median()):.3f}  filter pass {float(passf.float().mean()):.4f}")
    h2 = {s: sample_h2(A, B, s) for s in (4, 8, 16)}
    print("   sample H2 / d2 median: " + " ".join(f"1/{s}:{float((h2[s] / d2.clamp_min(1e-9)).median()):.3f}" for s in h2))
    for rot in (False, True):
        for k in (24, 32, 48, 64):
            P = basis(k, rot)
            PA, PB = (A - mu) @ P, (B - mu) @ P
            pd1, _ = exact_top2(PA, PB)
            L1q, E = q8_lower(PA, PB)
            line = f"   k={k:3d} rot={int(rot)}  proj d1/d1 med {float((pd1 / d1.clamp_min(1e-9)).median()):.3f}  2E med {float(2 * E.median()):.4f} |"
            for s in (4, 8, 16):
                surv_f = ~(pd1 > R2 * h2[s])
                surv_q = ~(L1q > R2 * h2[s])
                line += f" 1/{s}: surv fp32 {float(surv_f.float().mean()):.3f} int8 {float(surv_q.float().mean()):.3f} |"
            print(line)
    # the present full-dimension screen for comparison (same code arithmetic, k = 128, H2 from the int8 D1 over all columns)
    L1q, E = q8_lower(A, B)
    print(f"   full-dim int8 L1 with exact d2: surv {float((~(L1q > R2 * d2)).float().mean()):.3f}  (2E med {float(2 * E.median()):.4f})")
