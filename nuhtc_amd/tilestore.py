"""Tile sources of the WSI path with per-rank lazy reads.

The reference's contract (`Whole_Slide_Bag_FP`, tools/wsi_core/WholeSlideImage.py:832-898): a coordinate list (the .h5 `coords`
dataset with attrs patch_size / patch_level, written by `process_contours` :407-493, grid :460-466) plus a slide handle;
`__getitem__(idx)` reads `wsi.read_region(coord, patch_level, (patch_size, patch_size)).convert('RGB')` -> (tile, coord).
OpenSlide and HDF5 are not available offline, so a slide is a level-0 RGB array kept on disk and memory-mapped, and the
coordinate file is a .npy / .npz:

    slide.npy                     (H, W, 3) uint8, opened with mmap: a rank touches only the pages of the tiles it owns
    slide.npy + coords.npy/.npz   explicit (N, 2) int64 level-0 (x, y) origins (`coords`; optional `patch_size`) = the .h5 contract
    store directory               slide.npy + coords.npy (+ meta.json {"patch_size": ..}), see `write_store`
    tiles.npz                     pre-cut `tiles` (N, P, P, 3) + `coords` (N, 2) (small inputs, tests)

`TileBag` gives len(), `coords` (all of them: 16 bytes per tile) and `read(lo, hi)` (only those tiles are materialised), so that
`tools/infer_wsi.py` on N ranks reads each tile once, on the rank that owns it.
"""
import json
import os

import numpy as np


def grid_coords(height, width, step_size):
    """Level-0 origins of the full grid, row-major, (x, y): np.arange(0, size, step) per axis (WholeSlideImage.py:460-466)."""
    xs = np.arange(0, width, step_size)
    ys = np.arange(0, height, step_size)
    gy, gx = np.meshgrid(ys, xs, indexing='ij')
    return np.stack([gx.reshape(-1), gy.reshape(-1)], 1).astype(np.int64)


class TileBag:
    """`Whole_Slide_Bag_FP` over an array slide: tile i = the patch_size square at level-0 origin coords[i], RGB, zero padded
    past the slide edge (use_padding, WholeSlideImage.py:419-421)."""

    def __init__(self, slide, coords, patch_size=256, tiles=None):
        self.slide = slide                      # (H, W, >=3) uint8 array or memmap; None when `tiles` are pre-cut
        self.coords = np.asarray(coords, np.int64).reshape(-1, 2)
        self.patch_size = int(patch_size)
        self.tiles = tiles                      # optional pre-cut (N, P, P, 3) array
        self.reads = 0                          # tiles materialised so far (tests: a rank reads only its shard)
        if tiles is not None and len(tiles) != len(self.coords):
            raise ValueError('tiles and coords differ in length')

    def __len__(self):
        return len(self.coords)

    def read(self, lo, hi):
        """Tiles [lo, hi) as one (hi - lo, P, P, 3) uint8 array."""
        lo, hi = max(0, int(lo)), min(len(self), int(hi))
        P = self.patch_size
        self.reads += max(0, hi - lo)
        if self.tiles is not None:
            return np.ascontiguousarray(self.tiles[lo:hi])
        if hasattr(self.slide, 'read_regions'):          # a TIFF slide (nuhtc_amd.tiffslide): tiles assembled from its decoded TIFF tiles
            return self.slide.read_regions(self.coords[lo:hi], P)
        H, W = self.slide.shape[:2]
        out = np.zeros((max(0, hi - lo), P, P, 3), np.uint8)
        for k, (x, y) in enumerate(self.coords[lo:hi]):
            x, y = int(x), int(y)
            y0, y1, x0, x1 = max(y, 0), min(y + P, H), max(x, 0), min(x + P, W)
            if y1 > y0 and x1 > x0:
                out[k, y0 - y:y1 - y, x0 - x:x1 - x] = self.slide[y0:y1, x0:x1, :3]
        return out

    def view(self, lo, hi):
        """Tiles [lo, hi) as a lazy (n, P, P, 3) sequence: `view[a:b]` cuts / decodes only those tiles (the slide loop takes a batch at a time
        while earlier batches are on the GPU; a rank's shard never sits in host memory as a whole)."""
        return BagView(self, max(0, int(lo)), min(len(self), int(hi)))

    def __getitem__(self, i):
        """(tile, coord) like the reference's dataset item."""
        i = int(i)
        if i < 0:
            i += len(self)
        return self.read(i, i + 1)[0], self.coords[i]


class BagView:
    def __init__(self, bag, lo, hi):
        self.bag, self.lo, self.hi = bag, lo, max(lo, hi)
        self.shape = (self.hi - self.lo, bag.patch_size, bag.patch_size, 3)

    def __len__(self):
        return self.hi - self.lo

    def __getitem__(self, s):
        if isinstance(s, (int, np.integer)):
            i = int(s) + (len(self) if s < 0 else 0)
            return self.bag.read(self.lo + i, self.lo + i + 1)[0]
        a, b, step = s.indices(len(self))
        if step != 1:
            raise IndexError('a tile view is read in contiguous runs')
        return self.bag.read(self.lo + a, self.lo + max(a, b))


def write_store(path, slide, coords, patch_size=256):
    """Writes the directory form: slide.npy, coords.npy, meta.json."""
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, 'slide.npy'), np.ascontiguousarray(slide, np.uint8))
    np.save(os.path.join(path, 'coords.npy'), np.asarray(coords, np.int64).reshape(-1, 2))
    with open(os.path.join(path, 'meta.json'), 'w') as f:
        json.dump({'patch_size': int(patch_size), 'patch_level': 0}, f)


def _load_coords(path):
    if path.endswith('.h5'):      # the reference's coordinate file (nuhtc_amd.h5coords)
        from . import h5coords
        r = h5coords.read_coords(path)
        return r['coords'], int(r['attrs']['patch_size']) if 'patch_size' in r['attrs'] else None
    if path.endswith('.npz'):
        z = np.load(path)
        ps = int(z['patch_size']) if 'patch_size' in z.files else None
        return np.asarray(z['coords'], np.int64).reshape(-1, 2), ps
    return np.asarray(np.load(path), np.int64).reshape(-1, 2), None


def open_slide(path):
    """Level-0 image of a .npy slide, memory-mapped (nothing is read until tiles are cut)."""
    a = np.load(path, mmap_mode='r')
    if a.ndim != 3 or a.shape[2] < 3 or a.dtype != np.uint8:
        raise ValueError(f'{path}: a slide is an (H, W, 3) uint8 array')
    return a


def open_source(source, patch_size=256, step_size=192, coords=None, coords_fn=None):
    """-> TileBag.  `source`: .npy slide, store directory, or .npz of pre-cut tiles.  `coords`: optional path of a coordinate
    file (the reference's .h5 role); `coords_fn(slide) -> (N, 2)` computes them instead (tissue segmentation); default = the grid."""
    if os.path.isdir(source):
        slide = open_slide(os.path.join(source, 'slide.npy'))
        meta = {}
        if os.path.exists(os.path.join(source, 'meta.json')):
            with open(os.path.join(source, 'meta.json')) as f:
                meta = json.load(f)
        if int(meta.get('patch_level', 0)) != 0:
            raise ValueError('only patch_level 0 stores are supported (array slides have one level)')
        c, _ = _load_coords(os.path.join(source, 'coords.npy'))
        return TileBag(slide, c, int(meta.get('patch_size', patch_size)))
    if source.endswith('.npz'):
        z = np.load(source)
        tiles = z['tiles']
        return TileBag(None, z['coords'], tiles.shape[1], tiles=tiles)
    slide = open_slide(source)
    if coords is not None:
        c, ps = _load_coords(coords)
        return TileBag(slide, c, ps or patch_size)
    if coords_fn is not None:
        return TileBag(slide, coords_fn(slide), patch_size)
    return TileBag(slide, grid_coords(slide.shape[0], slide.shape[1], step_size), patch_size)
