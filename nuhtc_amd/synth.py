"""Synthetic inputs for tests and bench: no dataset or slide is available offline."""
import numpy as np


def noise_tiles(n, size=256, seed=1234):
    """S-noise: uniform random uint8 tiles, (n,size,size,3)."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, size, size, 3), dtype=np.uint8)


def _blur3(img, sigma=1.0):
    """separable Gaussian, edge-replicated, float32 (kept dependency-free on purpose)."""
    r = 3
    x = np.arange(-r, r + 1, dtype=np.float32)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    k /= k.sum()
    p = np.pad(img, ((r, r), (0, 0), (0, 0)), mode='edge')
    img = sum(k[i] * p[i:i + img.shape[0]] for i in range(2 * r + 1))
    p = np.pad(img, ((0, 0), (r, r), (0, 0)), mode='edge')
    return sum(k[i] * p[:, i:i + img.shape[1]] for i in range(2 * r + 1))


def nuclei_tile(tile_index, size=256, mean_count=60):
    """S-nuclei: H&E-like background with Poisson(mean_count) dark ellipses; rng = default_rng(2024+tile_index)."""
    rng = np.random.default_rng(2024 + int(tile_index))
    img = np.array([200.0, 160.0, 195.0], np.float32)[None, None, :] + rng.normal(0, 8, (size, size, 3)).astype(np.float32)
    k = int(rng.poisson(mean_count * (size / 256.0) ** 2))
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for _ in range(k):
        cx, cy = rng.uniform(0, size, 2)
        a, b = rng.uniform(5, 12, 2)
        th = rng.uniform(0, np.pi)
        col = np.array([90.0, 50.0, 130.0], np.float32) + rng.normal(0, 15, 3).astype(np.float32)
        dx, dy = xx - cx, yy - cy
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        m = (u / a) ** 2 + (v / b) ** 2 <= 1.0
        img[m] = col
    img = _blur3(img, 1.0)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def nuclei_tiles(n, size=256, start=0):
    return np.stack([nuclei_tile(start + i, size) for i in range(n)])


def fixed_load_rois(n_tiles, n_rois=1064, net_size=512, seed=7, size=(12, 40)):
    """Fixed-load mode: per tile `n_rois` boxes, sizes U(12,40) px in network space. (n_tiles,n_rois,4) f32."""
    rng = np.random.default_rng(seed)
    wh = rng.uniform(size[0], size[1], (n_tiles, n_rois, 2)).astype(np.float32)
    c = rng.uniform(0, net_size, (n_tiles, n_rois, 2)).astype(np.float32)
    b = np.concatenate([c - wh / 2, c + wh / 2], -1)
    return np.clip(b, 0, net_size).astype(np.float32)
