"""Tiled pyramidal TIFF / Aperio SVS slides through libtiff (nuhtc_amd.tiffslide): the OpenSlide calls of the reference's slide classes
(tools/wsi_core/WholeSlideImage.py:30-41,144-146,409-421,890-896: level_dimensions, level_downsamples, get_best_level_for_downsample,
read_region(location in level-0 pixels, level, size).convert('RGB')) restated for the formats whose pixels are plain TIFF.  OpenSlide itself
is absent: lossless files are checked bit-exactly against the arrays they were made from and against PIL's independent TIFF reader, JPEG
files against PIL's decode within the codecs' rounding."""
import os

import numpy as np
import pytest

from nuhtc_amd import tiffslide as T

pytestmark = pytest.mark.skipif(not T.available(), reason='libtiff not found')

DESC = 'Aperio Image Library v11.2.1\n1300x1000 [0,0 1300x1000] (240x240) JPEG/RGB Q=90|AppMag = 40|StripeWidth = 2040|MPP = 0.2520|Left = 25.6|Top = 23.4'


def _image(H=1000, W=1300):
    from nuhtc_amd import synth
    img = np.concatenate([np.concatenate([synth.nuclei_tile(10 + 6 * r + c, 256) for c in range(-(-W // 256))], 1) for r in range(-(-H // 256))], 0)
    return np.ascontiguousarray(img[:H, :W])


@pytest.mark.parametrize('compression', ['lzw', 'deflate', 'none'])
def test_lossless_pyramid_every_access_is_bit_exact(tmp_path, compression):
    img = _image()
    p = str(tmp_path / 's.svs')
    pyr = T.write_pyramid(p, img, levels=3, tile=240, compression=compression, description=DESC)
    s = T.TiffSlide(p)
    # the stripped thumbnail between level 0 and level 1 (where an SVS keeps it) is not a level
    assert s.level_count == 3 and s.level_dimensions == ((1300, 1000), (650, 500), (325, 250)) and s.level_downsamples == (1.0, 2.0, 4.0)
    assert s.shape == (1000, 1300, 3) and s.dimensions == (1300, 1000)
    assert s.properties['openslide.vendor'] == 'aperio' and s.properties['aperio.AppMag'] == '40' and s.properties['openslide.mpp-x'] == '0.2520'
    assert s.properties['openslide.objective-power'] == '40' and s.properties['openslide.level[2].downsample'] == '4.0'
    assert [s.get_best_level_for_downsample(d) for d in (0.5, 1, 1.9, 2, 3.99, 4, 64)] == [0, 0, 0, 1, 1, 2, 2]
    # level 0 through the array protocol: inside, across tile borders, at the edges (partial tiles)
    for (y0, y1, x0, x1) in [(0, 256, 0, 256), (100, 400, 200, 700), (239, 241, 479, 481), (900, 1000, 1200, 1300), (990, 1000, 0, 1300)]:
        assert np.array_equal(s[y0:y1, x0:x1], img[y0:y1, x0:x1]) and np.array_equal(s[y0:y1, x0:x1, :3], img[y0:y1, x0:x1])
    assert np.array_equal(np.asarray(s), img)
    # read_region: location in LEVEL-0 pixels whatever the level, size in level pixels, zeros outside the level
    r = s.read_region((-50, 950), 0, (256, 256))
    assert r.shape == (256, 256, 3) and not r[:, :50].any() and not r[50:].any() and np.array_equal(r[:50, 50:], img[950:1000, 0:206])
    assert np.array_equal(s.read_region((400, 200), 1, (100, 80)), pyr[1][100:180, 200:300])
    assert np.array_equal(s.read_region((1200, 900), 2, (64, 64))[:25, :25], pyr[2][225:250, 300:325]) and not s.read_region((1200, 900), 2, (64, 64))[25:].any()
    for k in range(3):
        assert np.array_equal(s.level_image(k), pyr[k])
    # strided views: the image at that downsample from the best level, nearest level pixel
    assert np.array_equal(s[::4, ::4], pyr[2][:250, :325]) and np.array_equal(s[::2, ::2], pyr[1][:500, :650])
    assert np.array_equal(s[::8, ::8], pyr[2][::2, ::2][:125, :163])
    ys, xs = (np.arange(0, 1000, 3) / 2).astype(int), (np.arange(0, 1300, 3) / 2).astype(int)        # stride 3 -> level 1 (downsample 2)
    assert np.array_equal(s[::3, ::3], pyr[1][ys][:, xs])
    assert np.array_equal(s[100:900:4, 40:1000:4], pyr[2][25:225, 10:250])
    # a batch (one thread, and on a pool with a TIFF handle per thread) = the same tiles one at a time
    coords = np.array([[x, y] for y in range(-64, 1000, 192) for x in range(-64, 1300, 192)])
    pooled = T.TiffSlide(p, threads=4)
    for tiles in (s.read_regions(coords, 256), pooled.read_regions(coords, 256)):
        for c, t in zip(coords, tiles):
            assert np.array_equal(t, s.read_region((int(c[0]), int(c[1])), 0, (256, 256)))
    s.close()
    pooled.close()


def test_against_pils_own_tiff_reader(tmp_path):
    """An independent reader of the same files: every directory PIL decodes equals the level (lossless), or stays within the JPEG codecs'
    rounding (PIL bundles its own libtiff / libjpeg-turbo)."""
    from PIL import Image
    img = _image(700, 900)
    p = str(tmp_path / 'l.tif')
    T.write_pyramid(p, img, levels=3, tile=256, compression='deflate', thumbnail=False)
    s = T.TiffSlide(p)
    assert s.properties['openslide.vendor'] == 'generic-tiff'
    im = Image.open(p)
    assert im.n_frames == 3
    for k in range(3):
        im.seek(k)
        assert np.array_equal(np.asarray(im.convert('RGB')), s.level_image(k))
    pj = str(tmp_path / 'j.svs')
    T.write_pyramid(pj, img, levels=2, tile=240, compression='jpeg', description=DESC, quality=95, thumbnail=False)
    sj = T.TiffSlide(pj)
    a = sj.level_image(0).astype(np.int32)
    assert np.abs(a - img).mean() < 3.0                      # it is the picture (JPEG at quality 95 of hard-edged synthetic nuclei)
    im = Image.open(pj)
    b = np.asarray(im.convert('RGB')).astype(np.int32)
    d = np.abs(a - b)
    print('jpeg level 0: this reader vs PIL: mean', d.mean(), 'max', d.max())
    assert d.mean() < 0.5 and d.max() <= 12              # two JPEG decoders (chroma upsampling / IDCT rounding), not two pictures


@pytest.mark.skipif(T._j2k_decoder() is None, reason='Pillow without OpenJPEG')
def test_aperio_jpeg2000_tiles(tmp_path):
    """Aperio's private compressions 33005 (JPEG 2000, RGB) and 33003 (JPEG 2000, YCbCr): a raw codestream per tile, read raw through libtiff and
    decoded by Pillow's OpenJPEG.  Written here reversibly: the RGB kind returns the source pixels, the YCbCr kind the fixed-point JFIF
    inverse (libjpeg's jdcolor.c tables) of the stored components -- evaluated independently below in floating point, within the 1 LSB of the
    16-bit fixed point -- and the source within the forward + inverse rounding."""
    img = _image(700, 900)
    for kind, tol in (('j2k-rgb', 0), ('j2k-ycbcr', 1)):
        p = str(tmp_path / f'{kind}.svs')
        pyr = T.write_pyramid(p, img, levels=2, tile=240, compression=kind, description=DESC.replace('JPEG/RGB', 'J2K/KDU'))
        s = T.TiffSlide(p)
        assert s.level_count == 2 and s.properties['aperio.AppMag'] == '40'
        for k in range(2):
            assert np.abs(s.level_image(k).astype(np.int32) - pyr[k]).max() <= tol
        t = s.read_regions(np.array([[-30, 500], [700, 10]]), 256)
        assert np.abs(t[0][:200, 30:].astype(np.int32) - img[500:700, 0:226]).max() <= tol and not t[0][200:].any() and not t[0][:, :30].any()
        assert np.abs(t[1][:, :200].astype(np.int32) - img[10:266, 700:900]).max() <= tol and not t[1][:, 200:].any()
        s.close()
    ycc = np.random.default_rng(1).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    f = ycc.astype(np.float64)
    want = np.stack([f[..., 0] + 1.402 * (f[..., 2] - 128), f[..., 0] - 0.344136 * (f[..., 1] - 128) - 0.714136 * (f[..., 2] - 128), f[..., 0] + 1.772 * (f[..., 1] - 128)], -1)
    assert np.abs(T._ycbcr_to_rgb(ycc).astype(np.float64) - np.clip(np.round(want), 0, 255)).max() <= 1


def test_not_a_slide(tmp_path):
    (tmp_path / 'x.svs').write_bytes(b'II*\0garbage')
    with pytest.raises(T.TiffError):
        T.TiffSlide(str(tmp_path / 'x.svs'))
    with pytest.raises(T.TiffError):
        T.TiffSlide(str(tmp_path / 'missing.tif'))
    img = _image(300, 400)
    p = str(tmp_path / 's.tif')
    T.write_pyramid(p, img, levels=1, tile=256, compression='none', thumbnail=False)
    s = T.TiffSlide(p)
    with pytest.raises(T.TiffError):
        s[5]
    assert s[0:0, 0:10].shape == (0, 10, 3) and s[290:400, 390:500].shape == (10, 10, 3)


def test_folder_of_svs_slides_through_seg_and_patch_and_the_tile_bag(tmp_path):
    """The folder level of the tool on `.svs` files (the reference's default --slide_ext): opened, segmented on the pyramid, tiled; the bag's
    tiles are the level-0 pixels, zero padded past the edge; the twin `.npy` slide gives nearly the same tile list (its 64x image is point-
    sampled from level 0, the TIFF's comes from the box-filtered pyramid level)."""
    from nuhtc_amd import slides, tilestore
    from test_tissue import tissue_slide_with_holes
    img, *_ = tissue_slide_with_holes(H=1536, W=2048)
    src = tmp_path / 'wsi'
    os.makedirs(src)
    T.write_pyramid(str(src / 'a.svs'), img, levels=4, tile=240, compression='lzw', description=DESC)
    np.save(src / 'b.npy', img)
    (src / 'c.svs').write_bytes(b'not a slide')
    outs = {}
    for ext in ('.svs', '.npy'):
        out = tmp_path / ('out' + ext[1:])
        dirs = dict(source=str(src), save_dir=str(out), patch_save_dir=str(out / 'patches'), mask_save_dir=str(out / 'masks'), stitch_save_dir=str(out / 'stitches'))
        for k, v in dirs.items():
            if k != 'source':
                os.makedirs(v)
        seg, flt, vis, pat = slides.default_parameters()
        names = [n for n in sorted(os.listdir(src)) if n.endswith(ext)]
        slides.seg_and_patch(**dirs, seg_params=seg, filter_params=flt, vis_params=vis, patch_params=pat, patch_size=128, step_size=128, seg=True, patch=True,
                             stitch=True, seg_downsample=8, slides=names, log=lambda *a: None)
        outs[ext] = out
    rows = open(outs['.svs'] / 'process_list_autogen.csv').read().splitlines()
    assert rows[1].startswith('a.svs,0,processed,3,') and rows[2].startswith('c.svs,0,failed_open,')
    ca, _, _ = slides.load_coords(str(outs['.svs'] / 'patches'), 'a')
    cb, _, _ = slides.load_coords(str(outs['.npy'] / 'patches'), 'b')
    def cover(c):                                  # the grids start at the contours' bounding boxes, a level pixel (8 px) apart: compare what the tiles cover
        m = np.zeros((1536 // 8 + 16, 2048 // 8 + 16), bool)
        for x, y in c // 8:
            m[y:y + 16, x:x + 16] = True
        return m
    ma, mb = cover(ca), cover(cb)
    assert len(ca) > 40 and abs(len(ca) - len(cb)) <= 0.1 * len(cb) and (ma & mb).sum() / (ma | mb).sum() > 0.9
    for f in (outs['.svs'] / 'masks' / 'a.png', outs['.svs'] / 'stitches' / 'a.jpg'):
        assert os.path.exists(f)
    slide = slides.open_array_slide(str(src / 'a.svs'))
    bag = tilestore.TileBag(slide, np.concatenate([ca, [[1984, 1472]]], 0), 128)          # + one tile that hangs over the corner
    tiles = bag.read(0, len(bag))
    for (x, y), t in zip(bag.coords[:-1], tiles[:-1]):
        want = np.zeros((128, 128, 3), np.uint8)             # use_padding: tiles of the last grid row / column hang over the edge
        crop = img[y:y + 128, x:x + 128]
        want[:crop.shape[0], :crop.shape[1]] = crop
        assert np.array_equal(t, want)
    assert np.array_equal(tiles[-1][:64, :64], img[1472:, 1984:]) and not tiles[-1][64:].any() and not tiles[-1][:, 64:].any()
    assert bag.reads == len(bag)


def test_pyramid_without_a_64x_level_is_segmented_on_its_own_level(tmp_path):
    """`seg_level = -1` on a pyramid FILE means the file's level from `get_best_level_for_downsample(64)` (tools/infer_wsi.py:222-229), whatever
    its true downsample: a 1 / 2 / 4 pyramid (a typical Aperio file is 1 / 4 / 16 / 32) is segmented on its 4x level, read whole
    (WholeSlideImage.py:159), thresholds and contour scaling from that level's own (non-integer) downsample pair (:176-191, :371-386) -- not on
    a virtual 64x image.  The tile list of the `.svs` equals, as one sequence, what the oracle's restatement of the reference gives for that
    level image; the process list records the level."""
    from nuhtc_amd import slides
    from oracle import tissue as OT
    from test_tissue import tissue_slide_with_holes
    img = np.ascontiguousarray(tissue_slide_with_holes(H=1024, W=1280)[0][:770, :1030])      # odd level sizes: non-integer downsample pairs
    src = tmp_path / 'wsi'
    os.makedirs(src)
    pyr = T.write_pyramid(str(src / 'a.svs'), img, levels=3, tile=240, compression='lzw', description=DESC)
    assert [p.shape[:2] for p in pyr] == [(770, 1030), (385, 515), (192, 257)]
    out = tmp_path / 'out'
    dirs = dict(source=str(src), save_dir=str(out), patch_save_dir=str(out / 'patches'), mask_save_dir=str(out / 'masks'), stitch_save_dir=str(out / 'stitches'))
    for k, v in dirs.items():
        if k != 'source':
            os.makedirs(v)
    seg, flt, vis, pat = slides.default_parameters()
    flt = dict(flt, a_t=1, a_h=1)                       # (the default area thresholds are sized for 64x levels of real slides)
    slides.seg_and_patch(**dirs, seg_params=seg, filter_params=flt, vis_params=vis, patch_params=pat, patch_size=64, step_size=64, seg=True, patch=True,
                         stitch=True, slides=['a.svs'], log=lambda *a: None)
    row = open(out / 'process_list_autogen.csv').read().splitlines()[1].split(',')
    assert row[:4] == ['a.svs', '0', 'processed', '2']                  # seg_level = the file's level 2
    got, ps, _ = slides.load_coords(str(out / 'patches'), 'a')
    scale = (1030 / 257.0, 770 / 192.0)
    want = []
    for c, hs in OT.segment_tissue(pyr[2], 1, a_t=1, a_h=1, level_scale=scale):
        want += OT.contour_tile_coords(c, hs, 64, 64)
    assert ps == 64 and len(want) > 30 and np.array_equal(got, np.asarray(want, np.int64).reshape(-1, 2))
    from PIL import Image
    assert Image.open(out / 'masks' / 'a.png').size == (257, 192) and Image.open(out / 'stitches' / 'a.jpg').size == (257, 192)
