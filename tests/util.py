"""Shared helpers for the tests (seeded synthetic data only; nothing reads /root/reference)."""
import numpy as np


def sift_like(rng, n, dim=128, unit=True):
    """Non-negative, clipped-at-0.2, unit-norm vectors like SIFT descriptors."""
    x = rng.gamma(0.6, 1.0, size=(n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-12
    x = np.minimum(x, 0.2)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-12
    if not unit:
        x = np.round(x * 512).clip(0, 255)
    return x.astype(np.float32)


def planted_pair(rng, n1, n2, n_common, noise=0.02, unit=True):
    """Two descriptor sets sharing n_common noisy correspondences at random positions."""
    a = sift_like(rng, n1)
    b = sift_like(rng, n2)
    n_common = min(n_common, n1, n2)
    ia = rng.permutation(n1)[:n_common]
    ib = rng.permutation(n2)[:n_common]
    pert = a[ia] + noise * rng.standard_normal((n_common, a.shape[1])).astype(np.float32)
    pert = np.maximum(pert, 0)
    pert /= np.linalg.norm(pert, axis=1, keepdims=True) + 1e-12
    b[ib] = pert
    if not unit:
        a = np.round(a * 512).clip(0, 255).astype(np.float32)
        b = np.round(b * 512).clip(0, 255).astype(np.float32)
    return a.astype(np.float32), b.astype(np.float32), ia, ib


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
