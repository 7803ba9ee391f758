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


def canvas_side(grid, step=192, size=256):
    return step * (grid - 1) + size


def nuclei_canvas(grid, step=192, size=256, rows=None, mean_count=60):
    """Synthetic slide for BASELINE configs[2]: one S-nuclei canvas that a `grid` x `grid` tiling of
    `size`-pixel tiles at stride `step` covers exactly, so neighbouring tiles share (size - step)-pixel overlaps and
    nuclei cross tile borders.  Nuclei are seeded per step x step cell (rng = default_rng(2024 + cell)), noise per
    cell row, so any rank can render just the tile rows `rows = (r0, r1)` it owns and gets the same pixels.
    Returns (uint8 (h, side, 3) band, y0) with the band starting at canvas row y0."""
    side = canvas_side(grid, step, size)
    r0, r1 = (0, grid) if rows is None else rows
    y0, y1 = r0 * step, min((r1 - 1) * step + size, side)
    ncell = (side + step - 1) // step
    dens = mean_count * (step / 256.0) ** 2
    out = np.empty((y1 - y0, side, 3), np.uint8)
    halo = 16                                    # largest semi-axis 12 + blur radius 3, rounded up
    nuclei = {}                                  # cell row -> its nuclei (kept for the three bands that draw them)
    for cr in range(max(y0 - halo, 0) // step, min((y1 + halo - 1) // step, ncell - 1) + 1):
        # band of one cell row (+ halo), clipped to the requested rows
        b0, b1 = max(cr * step, y0), min((cr + 1) * step, y1)
        if b1 <= b0:
            continue
        e0, e1 = max(b0 - 3, 0), min(b1 + 3, side)           # rows the blur of [b0, b1) reads
        img = np.empty((e1 - e0, side, 3), np.float32)
        for rr in range(e0 // step, (e1 - 1) // step + 1):   # noise is seeded per cell row
            n0, n1 = max(rr * step, e0), min((rr + 1) * step, e1)
            full = np.random.default_rng(7_000_000 + rr).standard_normal((min(step, side - rr * step), side, 3), dtype=np.float32) * 8.0
            img[n0 - e0:n1 - e0] = full[n0 - rr * step:n1 - rr * step]
        img += np.array([200.0, 160.0, 195.0], np.float32)
        for rr in range(max(cr - 1, 0), min(cr + 1, ncell - 1) + 1):      # nuclei of this and the adjacent cell rows
            if rr not in nuclei:
                lst = []
                for cc in range(ncell):
                    rng = np.random.default_rng(2024 + rr * ncell + cc)
                    for _ in range(int(rng.poisson(dens))):
                        cx, cy = rng.uniform(0, step, 2)
                        a, b = rng.uniform(5, 12, 2)
                        th = rng.uniform(0, np.pi)
                        col = np.array([90.0, 50.0, 130.0], np.float32) + rng.normal(0, 15, 3).astype(np.float32)
                        lst.append((cx + cc * step, cy + rr * step, a, b, np.cos(th), np.sin(th), col))
                nuclei[rr] = lst
                nuclei.pop(rr - 3, None)
            for cx, cy, a, b, ct, st, col in nuclei[rr]:
                r = int(max(a, b)) + 1
                ya, yb = max(int(cy) - r, e0), min(int(cy) + r + 2, e1)
                xa, xb = max(int(cx) - r, 0), min(int(cx) + r + 2, side)
                if yb <= ya or xb <= xa:
                    continue
                dy = (np.arange(ya, yb, dtype=np.float32) - np.float32(cy))[:, None]
                dx = (np.arange(xa, xb, dtype=np.float32) - np.float32(cx))[None, :]
                u = dx * np.float32(ct) + dy * np.float32(st)
                v = dy * np.float32(ct) - dx * np.float32(st)
                m = (u / np.float32(a)) ** 2 + (v / np.float32(b)) ** 2 <= 1.0
                img[ya - e0:yb - e0, xa:xb][m] = col
        # blur with the rows around the band present (edge replication only at the canvas border)
        pad_top, pad_bot = 3 - (b0 - e0), 3 - (e1 - b1)
        src = np.pad(img, ((pad_top, pad_bot), (0, 0), (0, 0)), mode='edge') if pad_top or pad_bot else img
        x = np.arange(-3, 4, dtype=np.float32)
        k = np.exp(-0.5 * x ** 2)
        k /= k.sum()
        v = sum(k[i] * src[i:i + (b1 - b0)] for i in range(7))
        p = np.pad(v, ((0, 0), (3, 3), (0, 0)), mode='edge')
        v = sum(k[i] * p[:, i:i + side] for i in range(7))
        out[b0 - y0:b1 - y0] = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return out, y0


class CanvasTiles:
    """Lazy (n, size, size, 3) view of the tiles [lo, hi) of a grid x grid tiling (row-major) over a canvas band."""

    def __init__(self, band, y0, grid, lo, hi, step=192, size=256):
        self.band, self.y0, self.grid, self.lo, self.hi, self.step, self.size = band, y0, grid, lo, hi, step, size
        self.shape = (hi - lo, size, size, 3)
        idx = np.arange(lo, hi)
        self.coords = np.stack([idx % grid * step, idx // grid * step], 1).astype(np.int64)

    @staticmethod
    def grid_coords(grid, step=192):
        """(grid * grid, 2) tile origins (x, y), row-major: the whole slide's coordinate list."""
        idx = np.arange(grid * grid)
        return np.stack([idx % grid * step, idx // grid * step], 1).astype(np.int64)

    def __len__(self):
        return self.hi - self.lo

    def __getitem__(self, s):
        if isinstance(s, (int, np.integer)):
            x, y = self.coords[s]
            return self.band[y - self.y0:y - self.y0 + self.size, x:x + self.size]
        return np.stack([self[i] for i in range(*s.indices(len(self)))]) if len(range(*s.indices(len(self)))) else np.zeros((0,) + self.shape[1:], np.uint8)


def _canvas_chunk(args):
    grid, step, size, a, b = args
    band, y0 = nuclei_canvas(grid, step, size, rows=(a, b))
    return a, band


def nuclei_canvas_parallel(grid, step=192, size=256, rows=None, workers=8):
    """nuclei_canvas over a process pool (identical pixels: every chunk is seeded the same way).  Call it before the
    process touches the GPU (the pool forks)."""
    import multiprocessing as mp
    r0, r1 = (0, grid) if rows is None else rows
    n = max(1, min(workers, r1 - r0))
    cuts = [r0 + (r1 - r0) * i // n for i in range(n + 1)]
    jobs = [(grid, step, size, cuts[i], cuts[i + 1]) for i in range(n) if cuts[i + 1] > cuts[i]]
    if len(jobs) == 1:
        return nuclei_canvas(grid, step, size, rows=(r0, r1))
    with mp.get_context('fork').Pool(len(jobs)) as pool:
        parts = pool.map(_canvas_chunk, jobs)
    # chunk [a, b) owns pixel rows [a*step, b*step); the last one keeps its tail
    out = [band[:(jobs[i][4] - jobs[i][3]) * step] if i + 1 < len(jobs) else band for i, (a, band) in enumerate(parts)]
    return np.concatenate(out, 0), r0 * step
