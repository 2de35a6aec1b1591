"""Seeded synthetic inputs: a procedural "world" seen through ideal pinhole cameras
K = [f 0 W/2; 0 f H/2; 0 0 1] (the reference's own convention, bundleAdjustmentRKf.m:1897-1901) on a
yaw/pitch grid.  The texture is a function of the WORLD RAY (multi-octave lattice noise plus sparse blobs
at several scales), so overlapping views agree at any focal length and nothing has to be stored.

This is data generation, not the product: it runs with torch on whatever device it is given (the GPU for
4K benches, the CPU for small test scenes) and never touches /root/reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def rot_yaw_pitch(yaw, pitch, roll=0.0):
    """World->camera rotation of a camera looking along +z, yawed about y and pitched about x."""
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    c2w = Ry @ Rx @ Rz
    return c2w.T


def grid_cameras(nx, ny, W, H, f, yaw_step, pitch_step, jitter_deg=1.0, seed=12345):
    """nx x ny yaw/pitch grid centred on the forward direction, with seeded jitter."""
    rng = np.random.default_rng(seed)
    cams = []
    for iy in range(ny):
        for ix in range(nx):
            yaw = (ix - (nx - 1) / 2) * yaw_step + math.radians(jitter_deg) * rng.uniform(-1, 1)
            pitch = (iy - (ny - 1) / 2) * pitch_step + math.radians(jitter_deg) * rng.uniform(-1, 1)
            roll = math.radians(jitter_deg) * rng.uniform(-1, 1)
            K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
            cams.append({"K": K, "R": rot_yaw_pitch(yaw, pitch, roll), "f": f, "noRotation": 0})
    return cams


def _hash3(ix, iy, iz, seed):
    """Integer lattice hash -> uniform [0,1) float32 (works on int64 tensors, any device)."""
    h = (ix * 73856093) ^ (iy * 19349663) ^ (iz * 83492791) ^ (seed * 2654435761)
    h = h & 0xFFFFFFFF
    h = ((h ^ (h >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    h = ((h ^ (h >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return (h & 0xFFFFFF).to(torch.float32) / float(1 << 24)


def _value_noise(p, seed):
    """Trilinear lattice noise at points p (..., 3) float32."""
    p0 = torch.floor(p)
    t = p - p0
    t = t * t * (3 - 2 * t)
    i = p0.to(torch.int64)
    out = 0
    for dx in (0, 1):
        wx = t[..., 0] if dx else 1 - t[..., 0]
        for dy in (0, 1):
            wy = t[..., 1] if dy else 1 - t[..., 1]
            for dz in (0, 1):
                wz = t[..., 2] if dz else 1 - t[..., 2]
                out = out + wx * wy * wz * _hash3(i[..., 0] + dx, i[..., 1] + dy, i[..., 2] + dz, seed)
    return out


def _blobs(p, seed):
    """One bright/dark gaussian blob per lattice cell (jittered centre, random sign and size)."""
    c = torch.floor(p).to(torch.int64)
    out = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                cx, cy, cz = c[..., 0] + dx, c[..., 1] + dy, c[..., 2] + dz
                ox = cx.to(torch.float32) + _hash3(cx, cy, cz, seed + 1)
                oy = cy.to(torch.float32) + _hash3(cx, cy, cz, seed + 2)
                oz = cz.to(torch.float32) + _hash3(cx, cy, cz, seed + 3)
                amp = _hash3(cx, cy, cz, seed + 4) * 2 - 1
                rad = 0.12 + 0.2 * _hash3(cx, cy, cz, seed + 5)
                d2 = (p[..., 0] - ox) ** 2 + (p[..., 1] - oy) ** 2 + (p[..., 2] - oz) ** 2
                out = out + amp * torch.exp(-d2 / (2 * rad * rad))
    return out


def world_color(rays, f, seed=12345, noise_sigma=0.0, gain=1.0, gen=None, finest_px=3.0):
    """RGB in [0,1] for unit world rays (..., 3): texture octaves from coarse down to `finest_px`-pixel
    lattice cells (the knob that sets how many SIFT features a view produces)."""
    rgb = []
    base = f / finest_px  # lattice frequency (cells per radian) of the finest octave
    for ch in range(3):
        acc = 0.5
        amp_n, amp_b = 0.22, 0.30
        for o in range(7):
            freq = base / (2.0 ** o)
            if freq < 2:
                break
            acc = acc + amp_n * (_value_noise(rays * freq, seed + 17 * o + 101 * ch) - 0.5) * (0.6 + 0.1 * o)
            if o >= 1:
                acc = acc + amp_b * 0.5 * _blobs(rays * (freq / 3.0), seed + 31 * o + 7) * (0.5 if ch else 0.6)
        rgb.append(acc)
    img = torch.stack(rgb, dim=-1) * gain
    if noise_sigma > 0:
        img = img + (noise_sigma / 255.0) * torch.randn(img.shape, device=img.device, generator=gen)
    return img.clamp(0, 1)


def render_view(cam, H, W, seed=12345, device="cpu", noise_sigma=0.0, gain=1.0, rows_per_chunk=256,
                finest_px=3.0):
    """uint8 H x W x 3 image of the world through `cam` (pixel (1,1) is the top-left pixel centre).
    On a CUDA device without sensor noise this is ONE launch of the library's synth kernel (same formulas);
    the torch implementation below serves CPU tests and the noisy variant."""
    if str(device).startswith("cuda") and noise_sigma == 0:
        from . import _capi

        out = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
        Kc = np.ascontiguousarray(np.asarray(cam["K"], np.float64))
        Rc = np.ascontiguousarray(np.asarray(cam["R"], np.float64))
        _capi.check(_capi.lib.aps_synth_view(_capi.ptr(Kc), _capi.ptr(Rc), int(H), int(W), int(seed) & 0xFFFFFFFF,
                                             float(finest_px), float(gain), _capi.ptr(out)))
        return out
    K = torch.tensor(np.asarray(cam["K"], np.float64), dtype=torch.float32, device=device)
    R = torch.tensor(np.asarray(cam["R"], np.float64), dtype=torch.float32, device=device)
    f = float(K[0, 0])
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
    gen = torch.Generator(device=device).manual_seed(seed + 999)
    xs = torch.arange(1, W + 1, dtype=torch.float32, device=device)
    for r0 in range(0, H, rows_per_chunk):
        r1 = min(H, r0 + rows_per_chunk)
        ys = torch.arange(r0 + 1, r1 + 1, dtype=torch.float32, device=device)
        X, Y = torch.meshgrid(xs, ys, indexing="xy")
        cx = (X - K[0, 2]) / K[0, 0]
        cy = (Y - K[1, 2]) / K[1, 1]
        rc = torch.stack([cx, cy, torch.ones_like(cx)], dim=-1)
        rw = rc @ R  # R' * rayC for row vectors
        rw = rw / rw.norm(dim=-1, keepdim=True)
        img = world_color(rw, f, seed, noise_sigma, gain, gen, finest_px)
        out[r0:r1] = (img * 255.0 + 0.5).to(torch.uint8)
    return out


def make_scene(nx, ny, W, H, f, overlap=0.4, seed=12345, device="cpu", jitter_deg=1.0, gains=False,
               finest_px=3.0):
    """Cameras + images of an nx x ny grid with the given fractional overlap between neighbours."""
    fov_x = 2 * math.atan(W / (2 * f))
    fov_y = 2 * math.atan(H / (2 * f))
    cams = grid_cameras(nx, ny, W, H, f, fov_x * (1 - overlap), fov_y * (1 - overlap), jitter_deg, seed)
    rng = np.random.default_rng(seed + 1)
    images = []
    for i, cam in enumerate(cams):
        g = float(rng.uniform(0.8, 1.25)) if gains else 1.0
        images.append(render_view(cam, H, W, seed, device, 1.0 if gains else 0.0, g, finest_px=finest_px))
    return images, cams
