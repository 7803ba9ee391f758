"""Tile sources with per-rank lazy reads (nuhtc_amd.tilestore): the `Whole_Slide_Bag_FP` contract of the reference
(tools/wsi_core/WholeSlideImage.py:832-898: coords + patch_size -> RGB tiles at level-0 origins, zero padded) over memory-mapped
array slides.  CPU only."""
import os

import numpy as np

from nuhtc_amd import parallel, tilestore, tissue, wsi


def _slide(h=300, w=420, seed=0):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, (h, w, 3)).astype(np.uint8)


def test_grid_bag_equals_eager_tile_grid(tmp_path):
    img = _slide()
    np.save(tmp_path / 's.npy', img)
    bag = tilestore.open_source(str(tmp_path / 's.npy'), 64, 48)
    tiles, coords = wsi.tile_grid(img, 64, 48)
    assert np.array_equal(bag.coords, coords) and len(bag) == len(tiles)
    assert np.array_equal(bag.read(0, len(bag)), tiles)
    t, c = bag[len(bag) - 1]                      # the reference's dataset item: (tile, coord), last tile is padded
    assert np.array_equal(t, tiles[-1]) and tuple(c) == tuple(coords[-1])


def test_each_rank_reads_only_its_shard(tmp_path):
    img = _slide(seed=1)
    np.save(tmp_path / 's.npy', img)
    tiles, _ = wsi.tile_grid(img, 64, 48)
    seen = []
    for rank in range(3):
        bag = tilestore.open_source(str(tmp_path / 's.npy'), 64, 48)
        assert isinstance(bag.slide, np.memmap)   # nothing materialised by opening
        lo, hi = parallel.shard_range(len(bag), rank, 3)
        got = bag.read(lo, hi)
        assert bag.reads == hi - lo and np.array_equal(got, tiles[lo:hi])
        seen.append(got)
    assert np.array_equal(np.concatenate(seen, 0), tiles)


def test_coords_file_and_store_directory(tmp_path):
    img = _slide(seed=2)
    np.save(tmp_path / 's.npy', img)
    coords = np.array([[0, 0], [100, 37], [400, 280], [-10, -20]], np.int64)          # incl. past-the-edge and negative origins
    np.savez(tmp_path / 'c.npz', coords=coords, patch_size=32)
    bag = tilestore.open_source(str(tmp_path / 's.npy'), 256, 192, coords=str(tmp_path / 'c.npz'))
    assert bag.patch_size == 32
    ref = np.zeros((4, 32, 32, 3), np.uint8)
    ref[0] = img[0:32, 0:32]
    ref[1] = img[37:69, 100:132]
    ref[2, :20, :20] = img[280:300, 400:420]
    ref[3, 20:, 10:] = img[0:12, 0:22]
    assert np.array_equal(bag.read(0, 4), ref)
    tilestore.write_store(str(tmp_path / 'st'), img, coords[:3], patch_size=32)
    b2 = tilestore.open_source(str(tmp_path / 'st'))
    assert b2.patch_size == 32 and np.array_equal(b2.read(0, 3), ref[:3]) and np.array_equal(b2.coords, coords[:3])
    # .npz of pre-cut tiles
    np.savez(tmp_path / 't.npz', tiles=ref, coords=coords)
    b3 = tilestore.open_source(str(tmp_path / 't.npz'))
    assert len(b3) == 4 and np.array_equal(b3.read(1, 3), ref[1:3])


def test_tissue_coords_bag_equals_read_tiles(tmp_path):
    from test_tissue import synthetic_slide
    img = synthetic_slide()[0]
    np.save(tmp_path / 's.npy', img)
    fn = lambda s: tissue.tissue_tile_coords(s, 256, 192, scale=8)[0]
    bag = tilestore.open_source(str(tmp_path / 's.npy'), 256, 192, coords_fn=fn)
    coords, _, _ = tissue.tissue_tile_coords(img, 256, 192, scale=8)
    assert len(bag) == len(coords) > 40 and np.array_equal(bag.coords, coords)
    lo, hi = parallel.shard_range(len(bag), 1, 4)
    assert np.array_equal(bag.read(lo, hi), tissue.read_tiles(img, coords[lo:hi], 256)) and bag.reads == hi - lo


def test_part_from_lists_round_trip():
    """The per-detection fallback of one overflowing batch joins the packed batches as one more part (wsi.infer_tiles)."""
    rng = np.random.RandomState(3)
    rec = dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
    for i in range(7):
        h, w = rng.randint(3, 40), rng.randint(3, 70)
        m = rng.rand(h, w) > 0.4
        m[0, 0] = m[-1, -1] = True
        x0, y0 = rng.randint(0, 1000), rng.randint(0, 1000)
        rec['tile'].append(i // 2)
        rec['box'].append(np.array([x0, y0, x0 + w, y0 + h], np.float64))
        rec['score'].append(float(rng.rand()))
        rec['label'].append(int(rng.randint(0, 5)))
        rec['mask'].append((m, x0, y0))
        n = rng.randint(4, 12)
        ring = rng.randint(0, 100, (n, 2)).astype(np.int64)
        rec['ring'].append(np.concatenate([ring, ring[:1]], 0))
    part = wsi._part_from_lists(rec)
    out = wsi._records_from_parts([part, part])
    assert len(out['tile']) == 14
    for k in range(14):
        i = k % 7
        assert out['tile'][k] == rec['tile'][i] and out['label'][k] == rec['label'][i] and out['score'][k] == rec['score'][i]
        assert np.array_equal(out['box'][k], rec['box'][i]) and np.array_equal(out['ring'][k], rec['ring'][i])
        m, x0, y0 = out['mask'][k]
        assert np.array_equal(m, rec['mask'][i][0]) and (x0, y0) == rec['mask'][i][1:]
    # and it packs like the list form
    a = wsi.pack_records(out)
    b = wsi.pack_records(dict(tile=rec['tile'] * 2, box=rec['box'] * 2, score=rec['score'] * 2, label=rec['label'] * 2, mask=rec['mask'] * 2, ring=rec['ring'] * 2))
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x.numpy(), y.numpy())
