"""A tiny in-memory scene store with the layout of LaRa's gobjaverse HDF5 file (dataLoader/gobjverse.py:24-44,
:124-146), seeded; stands in for h5py.File in the loader tests and in tests/golden/make_loader_fixture.py."""
import numpy as np


class _Group(dict):
    """dict with the parts of the h5py group interface the loader uses (keys(), [name]; leaves are numpy arrays)."""


def make_store(seed=5, n_scenes=12, n_views=10, res=32):
    rng = np.random.default_rng(seed)
    store = _Group()
    for s in range(n_scenes):
        sc = _Group()
        for v in range(n_views):
            az, el, r = rng.uniform(0, 2 * np.pi), rng.uniform(-0.4, 0.6), rng.uniform(1.6, 2.1)
            pos = r * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
            fwd = -pos / np.linalg.norm(pos)
            right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
            c2w = np.eye(4, dtype=np.float32)
            c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, np.cross(fwd, right), fwd, pos
            sc[f"c2w_{v}"] = c2w
            sc[f"fov_{v}"] = np.array([0.75, 0.7 + 0.01 * v], dtype=np.float32)
            sc[f"image_{v}"] = rng.integers(0, 256, (res, res, 4), dtype=np.uint8)
            sc[f"normal_{v}"] = rng.integers(0, 256, (res, res, 3), dtype=np.uint8)
        groups = _Group()
        for n in (1, 4):
            for k in range(n):
                groups[f"groups_{n}_{k}"] = np.array(sorted(rng.choice(n_views, size=3, replace=False)))
        sc["groups"] = groups
        store[f"scene_{s:04d}"] = sc
    return store
