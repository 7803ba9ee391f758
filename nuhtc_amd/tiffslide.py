"""Whole-slide images stored as tiled (pyramidal) TIFF -- Aperio `.svs`, generic tiled `.tif` / `.tiff` -- read through libtiff.

The reference opens slides with OpenSlide (tools/wsi_core/WholeSlideImage.py:30-41: `openslide.open_slide`, `level_dimensions`,
`level_downsamples`, `get_best_level_for_downsample`, `read_region(location, level, size).convert('RGB')`); OpenSlide does not exist in this
image, libtiff (the library OpenSlide itself decodes these two formats with) does.  `TiffSlide` restates the part of OpenSlide's behaviour the
path uses, for the formats whose pixels are plain TIFF:

  * levels = the TILED directories of the file, largest first (OpenSlide's Aperio and generic-TIFF readers: stripped directories are the
    thumbnail / label / macro images, not levels); `level_downsamples[k]` = mean of the width and height ratios to level 0;
  * `get_best_level_for_downsample(d)` = the last level whose downsample is <= d (level 0 below the first);
  * `read_region((x, y), level, (w, h))`: (x, y) in LEVEL-0 pixels, the region w x h in level pixels; pixels outside the level are 0 -- OpenSlide
    returns transparent black there and the reference's `.convert('RGB')` drops the alpha;
  * JPEG tiles are decoded by libtiff with YCbCr -> RGB conversion; Aperio's JPEG 2000 tiles (compression 33003 YCbCr / 33005 RGB: a raw
    codestream per tile, no codec of libtiff) are read raw and decoded by Pillow's OpenJPEG plugin -- OpenSlide uses OpenJPEG too --, the YCbCr
    kind converted with libjpeg's fixed-point tables; chroma-subsampled codestreams are decoded as Pillow decodes them (no such file here to
    test); 16-bit and non-RGB layouts go through libtiff's RGBA renderer or are refused.
  PARITY UNPINNED against OpenSlide itself (absent): lossless tiles are bit-exact by construction (tests/test_tiffslide.py checks every access
  pattern against the source arrays and against PIL's own TIFF reader); for JPEG tiles OpenSlide decodes with libjpeg-turbo, whose chroma
  upsampling can differ from this libtiff's libjpeg by an LSB.

Array protocol for the rest of the package (nuhtc_amd.slides / tissue / tilestore treat a slide as an (H, W, 3) uint8 array): `shape`,
`slide[y0:y1, x0:x1, :3]` = level-0 pixels, `slide[::s, ::s]` = the image at downsample s taken from the best pyramid level (the reference
segments tissue on such a level image), `read_regions(coords, P)` = a batch of tiles."""
import ctypes
import ctypes.util
import glob
import os
import threading
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np

EXTENSIONS = ('.svs', '.tif', '.tiff')

_T = dict(IMAGEWIDTH=256, IMAGELENGTH=257, BITSPERSAMPLE=258, COMPRESSION=259, PHOTOMETRIC=262, IMAGEDESCRIPTION=270, SAMPLESPERPIXEL=277,
          ROWSPERSTRIP=278, PLANARCONFIG=284, TILEWIDTH=322, TILELENGTH=323, SUBFILETYPE=254, JPEGCOLORMODE=65538, JPEGQUALITY=65537)
_COMPRESSION_JPEG, _PHOTOMETRIC_RGB, _PHOTOMETRIC_YCBCR = 7, 2, 6
_APERIO_J2K_YCBCR, _APERIO_J2K_RGB = 33003, 33005          # Aperio's private compression tags: a raw JPEG 2000 codestream per tile


def _j2k_decoder():
    """Pillow's JPEG 2000 plugin (OpenJPEG) or None: libtiff has no codec for Aperio's 33003 / 33005 tiles, OpenSlide decodes them with OpenJPEG."""
    try:
        from PIL import Image, features
        return Image if features.check('jpg_2000') else None
    except Exception:
        return None


def _ycbcr_to_rgb(ycc):
    """JFIF YCbCr -> RGB in the 16-bit fixed point of libjpeg's jdcolor.c (the tables OpenSlide's JPEG 2000 path is modelled on; OpenSlide's own
    source is not here to compare: unpinned)."""
    y = ycc[..., 0].astype(np.int32)
    cb = ycc[..., 1].astype(np.int32) - 128
    cr = ycc[..., 2].astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)
_LIB = None
_TRIED = False


class TiffError(RuntimeError):
    pass


def _lib():
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    cands = [os.environ.get('NUHTC_TIFF_LIB'), ctypes.util.find_library('tiff')]
    for pat in ('/usr/lib/x86_64-linux-gnu/libtiff.so*', '/usr/lib64/libtiff.so*', '/usr/local/lib/libtiff.so*', '/opt/conda/lib/libtiff.so*'):
        cands += sorted(glob.glob(pat), key=len)
    P, I, U32, U16, S64, CP = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint16, ctypes.c_int64, ctypes.c_char_p
    for cand in cands:
        if not cand:
            continue
        try:
            lib = ctypes.CDLL(cand)
            for name, res, args in (('TIFFOpen', P, [CP, CP]), ('TIFFClose', None, [P]), ('TIFFIsTiled', I, [P]), ('TIFFSetDirectory', I, [P, U16]),
                                    ('TIFFNumberOfDirectories', U16, [P]), ('TIFFTileSize', S64, [P]), ('TIFFComputeTile', U32, [P, U32, U32, U32, U16]),
                                    ('TIFFReadEncodedTile', S64, [P, U32, P, S64]), ('TIFFReadRawTile', S64, [P, U32, P, S64]), ('TIFFWriteRawTile', S64, [P, U32, P, S64]),
                                    ('TIFFReadRGBATile', I, [P, U32, U32, P]),
                                    ('TIFFReadRGBAImageOriented', I, [P, U32, U32, P, I, I]), ('TIFFWriteEncodedTile', S64, [P, U32, P, S64]),
                                    ('TIFFWriteEncodedStrip', S64, [P, U32, P, S64]), ('TIFFWriteDirectory', I, [P]), ('TIFFSetWarningHandler', P, [P]),
                                    ('TIFFSetErrorHandler', P, [P]), ('TIFFIsCODECConfigured', I, [U16])):
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            lib.TIFFGetField.restype = I          # variadic: arguments are passed as explicit ctypes objects
            lib.TIFFSetField.restype = I
        except (OSError, AttributeError):
            continue
        lib.TIFFSetWarningHandler(None)
        lib.TIFFSetErrorHandler(None)
        lib._path = cand
        _LIB = lib
        break
    return _LIB


def available():
    return _lib() is not None


def is_tiff_slide(path):
    return os.path.isfile(path) and path.lower().endswith(EXTENSIONS)


def _get(lib, tif, tag, ctype):
    v = ctype()
    return v.value if lib.TIFFGetField(ctypes.c_void_p(tif), ctypes.c_uint32(_T[tag]), ctypes.byref(v)) else None


class _Level:
    __slots__ = ('dir', 'w', 'h', 'tw', 'th', 'tiled', 'compression', 'photometric', 'spp', 'bps', 'planar', 'fast')


class TiffSlide:
    def __init__(self, path, cache_tiles=256, threads=1):
        lib = _lib()
        if lib is None:
            raise TiffError('libtiff not found (set NUHTC_TIFF_LIB)')
        self.path, self._lib_, self._local, self._cache_tiles, self._threads = str(path), lib, threading.local(), int(cache_tiles), int(threads)
        self._pool = None
        self._all_handles, self._hlock = [], threading.Lock()
        tif = self._handle()
        dirs = []
        for d in range(lib.TIFFNumberOfDirectories(tif)):
            if not lib.TIFFSetDirectory(tif, d):
                break
            lv = _Level()
            lv.dir, lv.w, lv.h = d, _get(lib, tif, 'IMAGEWIDTH', ctypes.c_uint32), _get(lib, tif, 'IMAGELENGTH', ctypes.c_uint32)
            lv.tiled = bool(lib.TIFFIsTiled(tif))
            lv.tw = _get(lib, tif, 'TILEWIDTH', ctypes.c_uint32) if lv.tiled else None
            lv.th = _get(lib, tif, 'TILELENGTH', ctypes.c_uint32) if lv.tiled else None
            lv.compression = _get(lib, tif, 'COMPRESSION', ctypes.c_uint16) or 1
            lv.photometric = _get(lib, tif, 'PHOTOMETRIC', ctypes.c_uint16)
            lv.spp = _get(lib, tif, 'SAMPLESPERPIXEL', ctypes.c_uint16) or 1
            lv.bps = _get(lib, tif, 'BITSPERSAMPLE', ctypes.c_uint16) or 1
            lv.planar = _get(lib, tif, 'PLANARCONFIG', ctypes.c_uint16) or 1
            # the fast path hands the decoded tile over as it is: 8-bit RGB (or JPEG YCbCr converted by the codec), pixel-interleaved
            lv.fast = lv.tiled and lv.bps == 8 and lv.spp == 3 and lv.planar == 1 and (
                lv.photometric == _PHOTOMETRIC_RGB or (lv.photometric == _PHOTOMETRIC_YCBCR and lv.compression == _COMPRESSION_JPEG))
            if d == 0:
                desc = ctypes.c_char_p()
                self.description = desc.value.decode('latin-1', 'replace') if lib.TIFFGetField(ctypes.c_void_p(tif), ctypes.c_uint32(_T['IMAGEDESCRIPTION']), ctypes.byref(desc)) and desc.value else ''
            if lv.w and lv.h:
                dirs.append(lv)
        if not dirs:
            raise TiffError(f'{path}: not a TIFF file libtiff can open')
        first = dirs[0]
        self._j2k = None
        if first.compression in (_APERIO_J2K_YCBCR, _APERIO_J2K_RGB):
            self._j2k = _j2k_decoder()
            if self._j2k is None:
                raise TiffError(f'{path}: JPEG 2000 tiles (Aperio compression {first.compression}) need Pillow with OpenJPEG; convert the slide (e.g. to JPEG tiles)')
            self._threads = max(self._threads, min(16, os.cpu_count() or 4))      # ~10 ms per 240-pixel tile in OpenJPEG (the GIL is released): batches go to the pool
        elif not lib.TIFFIsCODECConfigured(first.compression):
            raise TiffError(f'{path}: TIFF compression {first.compression} is not configured in {lib._path}')
        if first.tiled:
            levels = [first]
            for lv in dirs[1:]:       # further tiled directories, strictly smaller, same aspect within a pixel of rounding: the pyramid
                if lv.tiled and lv.w < levels[-1].w and lv.h < levels[-1].h and abs(lv.w * first.h - lv.h * first.w) <= max(first.w, first.h) * 2:
                    levels.append(lv)
        else:
            if first.w * first.h > (1 << 27):
                raise TiffError(f'{path}: a stripped TIFF of {first.w} x {first.h} pixels is not a slide format (tiled TIFF expected)')
            levels = [first]
        self._levels = levels
        self.level_count = len(levels)
        self.level_dimensions = tuple((lv.w, lv.h) for lv in levels)
        self.level_downsamples = tuple((first.w / lv.w + first.h / lv.h) / 2.0 for lv in levels)
        self.dimensions = self.level_dimensions[0]
        self.properties = self._properties()
        self._stripped = None
        if not first.tiled:
            raster = np.empty((first.h, first.w, 4), np.uint8)
            lib.TIFFSetDirectory(tif, 0)
            if not lib.TIFFReadRGBAImageOriented(tif, first.w, first.h, raster.ctypes.data_as(ctypes.c_void_p), 1, 0):
                raise TiffError(f'{path}: libtiff cannot render this image')
            self._stripped = np.ascontiguousarray(raster[..., :3])

    # ---- handles, one per thread (a TIFF* keeps a current directory and decoder state)
    def _handle(self):
        h = getattr(self._local, 'tif', None)
        if h is None:
            h = self._lib_.TIFFOpen(os.fsencode(self.path), b'r')
            if not h:
                raise TiffError(f'{self.path}: cannot be opened as TIFF')
            self._local.tif, self._local.dir, self._local.cache = h, -1, OrderedDict()
            with self._hlock:
                self._all_handles.append(h)
        return h

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        with self._hlock:
            for h in self._all_handles:
                self._lib_.TIFFClose(h)
            self._all_handles = []
        self._local = threading.local()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _properties(self):
        p = {}
        d = self.description
        if d.startswith('Aperio'):           # "Aperio ...\n<geometry> JPEG/RGB Q=30|AppMag = 20|...|MPP = 0.4990": OpenSlide's aperio.* keys
            p['openslide.vendor'] = 'aperio'
            for part in d.split('|')[1:]:
                k, _, v = part.partition('=')
                if _:
                    p['aperio.' + k.strip()] = v.strip()
            if 'aperio.AppMag' in p:
                p['openslide.objective-power'] = p['aperio.AppMag']
            if 'aperio.MPP' in p:
                p['openslide.mpp-x'] = p['openslide.mpp-y'] = p['aperio.MPP']
        else:
            p['openslide.vendor'] = 'generic-tiff'
        for k, (w, h) in enumerate(self.level_dimensions):
            p[f'openslide.level[{k}].width'], p[f'openslide.level[{k}].height'] = str(w), str(h)
            p[f'openslide.level[{k}].downsample'] = repr(self.level_downsamples[k])
        p['openslide.level-count'] = str(self.level_count)
        return p

    def get_best_level_for_downsample(self, downsample):
        best = 0
        for k, d in enumerate(self.level_downsamples):
            if d <= downsample + 1e-9:
                best = k
        return best

    # ---- tiles
    def _tile(self, level, tx, ty):
        """Decoded tile (th, tw, 3) uint8 of `level` (full tile size; the part past the image edge is whatever the file holds)."""
        lv = self._levels[level]
        tif = self._handle()
        loc = self._local
        key = (level, tx, ty)
        hit = loc.cache.get(key)
        if hit is not None:
            loc.cache.move_to_end(key)
            return hit
        lib = self._lib_
        if loc.dir != lv.dir:
            if not lib.TIFFSetDirectory(tif, lv.dir):
                raise TiffError(f'{self.path}: directory {lv.dir} unreadable')
            loc.dir = lv.dir
            if lv.compression == _COMPRESSION_JPEG and lv.photometric == _PHOTOMETRIC_YCBCR:
                lib.TIFFSetField(ctypes.c_void_p(tif), ctypes.c_uint32(_T['JPEGCOLORMODE']), ctypes.c_int(1))      # JPEGCOLORMODE_RGB
        x, y = tx * lv.tw, ty * lv.th
        if lv.compression in (_APERIO_J2K_YCBCR, _APERIO_J2K_RGB):      # the tile's bytes are a raw JPEG 2000 codestream: Pillow (OpenJPEG) decodes it
            import io
            raw = ctypes.create_string_buffer(lv.tw * lv.th * 4 + 65536)
            n = lib.TIFFReadRawTile(tif, lib.TIFFComputeTile(tif, x, y, 0, 0), raw, len(raw))
            if n <= 0:
                raise TiffError(f'{self.path}: tile ({tx}, {ty}) of level {level} cannot be read')
            try:
                im = self._j2k.open(io.BytesIO(raw.raw[:n]))
                a = np.asarray(im)
            except Exception as e:
                raise TiffError(f'{self.path}: JPEG 2000 tile ({tx}, {ty}) of level {level} cannot be decoded ({e})')
            if a.ndim != 3 or a.shape[2] < 3 or a.dtype != np.uint8 or a.shape[0] > lv.th or a.shape[1] > lv.tw:
                raise TiffError(f'{self.path}: JPEG 2000 tile ({tx}, {ty}) decodes to {a.shape} {a.dtype}, expected 8-bit colour of at most {lv.tw} x {lv.th}')
            out = np.zeros((lv.th, lv.tw, 3), np.uint8)
            out[:a.shape[0], :a.shape[1]] = _ycbcr_to_rgb(a[..., :3]) if lv.compression == _APERIO_J2K_YCBCR else a[..., :3]
        elif lv.fast:
            out = np.empty((lv.th, lv.tw, 3), np.uint8)
            n = lib.TIFFReadEncodedTile(tif, lib.TIFFComputeTile(tif, x, y, 0, 0), out.ctypes.data_as(ctypes.c_void_p), out.nbytes)
            if n < 0:
                raise TiffError(f'{self.path}: tile ({tx}, {ty}) of level {level} cannot be decoded')
        else:                                   # anything else libtiff can render: RGBA raster, bottom row first
            ras = np.empty((lv.th, lv.tw, 4), np.uint8)
            if not lib.TIFFReadRGBATile(tif, x, y, ras.ctypes.data_as(ctypes.c_void_p)):
                raise TiffError(f'{self.path}: tile ({tx}, {ty}) of level {level} cannot be rendered')
            out = np.ascontiguousarray(ras[::-1, :, :3])      # (tif_getimage.c moves a partial edge tile's rows to the top of the flipped raster and zero-fills the rest)
        if self._cache_tiles > 0:
            loc.cache[key] = out
            if len(loc.cache) > self._cache_tiles:
                loc.cache.popitem(last=False)
        return out

    def _read_level(self, level, lx, ly, w, h):
        """(h, w, 3) uint8 at `level` with origin (lx, ly) in level pixels; 0 outside the level."""
        out = np.zeros((max(0, h), max(0, w), 3), np.uint8)
        lv = self._levels[level]
        x0, y0, x1, y1 = max(lx, 0), max(ly, 0), min(lx + w, lv.w), min(ly + h, lv.h)
        if x1 <= x0 or y1 <= y0:
            return out
        if self._stripped is not None:
            out[y0 - ly:y1 - ly, x0 - lx:x1 - lx] = self._stripped[y0:y1, x0:x1]
            return out
        for ty in range(y0 // lv.th, (y1 - 1) // lv.th + 1):
            for tx in range(x0 // lv.tw, (x1 - 1) // lv.tw + 1):
                t = self._tile(level, tx, ty)
                ax0, ay0 = max(x0, tx * lv.tw), max(y0, ty * lv.th)
                ax1, ay1 = min(x1, (tx + 1) * lv.tw), min(y1, (ty + 1) * lv.th)
                out[ay0 - ly:ay1 - ly, ax0 - lx:ax1 - lx] = t[ay0 - ty * lv.th:ay1 - ty * lv.th, ax0 - tx * lv.tw:ax1 - tx * lv.tw]
        return out

    def read_region(self, location, level, size):
        """OpenSlide's call: `location` (x, y) in level-0 pixels, `size` (w, h) in pixels of `level` -> (h, w, 3) uint8 RGB."""
        d = self.level_downsamples[level]
        return self._read_level(level, int(location[0] / d), int(location[1] / d), int(size[0]), int(size[1]))

    def read_regions(self, coords, patch_size, level=0):
        """Tiles at the level-0 origins `coords` as one (n, P, P, 3) array.  One thread by default: consecutive tiles of a slide share decoded
        TIFF tiles through the handle's cache (4.5 k tiles/s of 256 x 256 from 240 x 240 JPEG tiles on one core of the build container, against
        2.0-2.2 k with 4-8 threads, each with a cache of its own); `threads` > 1 decodes on a pool, one TIFF handle per thread."""
        coords = np.asarray(coords, np.int64).reshape(-1, 2)
        out = np.zeros((len(coords), patch_size, patch_size, 3), np.uint8)
        if self._threads <= 1:
            for k in range(len(coords)):
                out[k] = self.read_region((int(coords[k, 0]), int(coords[k, 1])), level, (patch_size, patch_size))
            return out
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=max(1, self._threads))

        def one(k):
            out[k] = self.read_region((int(coords[k, 0]), int(coords[k, 1])), level, (patch_size, patch_size))
        list(self._pool.map(one, range(len(coords))))
        return out

    # ---- array protocol
    @property
    def shape(self):
        return (self.dimensions[1], self.dimensions[0], 3)

    ndim, dtype = 3, np.dtype(np.uint8)

    def __len__(self):
        return self.dimensions[1]

    def __array__(self, dtype=None, copy=None):
        H, W = self.shape[:2]
        if H * W > (1 << 28):
            raise TiffError(f'{self.path}: refusing to materialise the whole {W} x {H} level-0 image; index the slide instead')
        a = self._read_level(0, 0, 0, W, H)
        return a.astype(dtype) if dtype is not None else a

    def level_image(self, level):
        w, h = self.level_dimensions[level]
        return self._read_level(level, 0, 0, w, h)

    def _axis(self, sl, n):
        if isinstance(sl, (int, np.integer)):
            raise TiffError('integer indices are not supported on a slide: use slices')
        start, stop, step = sl.indices(n)
        if step < 1:
            raise TiffError('negative steps are not supported on a slide')
        return start, max(start, stop), step

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (3 - len(key))
        H, W = self.shape[:2]
        y0, y1, sy = self._axis(key[0], H)
        x0, x1, sx = self._axis(key[1], W)
        if sy == 1 and sx == 1:
            return self._read_level(0, x0, y0, x1 - x0, y1 - y0)[:, :, key[2]]
        # strided view = the image at that downsample, taken from the best pyramid level by nearest level pixel
        level = self.get_best_level_for_downsample(min(sy, sx))
        lv, d = self._levels[level], self.level_downsamples[level]
        ys = np.minimum((np.arange(y0, y1, sy) / d).astype(np.int64), lv.h - 1)
        xs = np.minimum((np.arange(x0, x1, sx) / d).astype(np.int64), lv.w - 1)
        out = np.zeros((len(ys), len(xs), 3), np.uint8)
        if len(ys) == 0 or len(xs) == 0:
            return out[:, :, key[2]]
        band = lv.th if self._stripped is None else 1024         # one band of tile rows at a time: bounded memory on a single-level file too
        cx0, cx1 = int(xs[0]), int(xs[-1]) + 1
        for b0 in range(int(ys[0]) // band * band, int(ys[-1]) + 1, band):
            sel = np.nonzero((ys >= b0) & (ys < b0 + band))[0]
            if len(sel):
                r0, r1 = int(ys[sel[0]]), int(ys[sel[-1]]) + 1
                blk = self._read_level(level, cx0, r0, cx1 - cx0, r1 - r0)
                out[sel] = blk[ys[sel] - r0][:, xs - cx0]
        return out[:, :, key[2]]


def write_pyramid(path, image, levels=3, tile=256, compression='jpeg', description=None, quality=90, thumbnail=True):
    """Test / conversion helper: a tiled pyramidal TIFF in the layout of an Aperio SVS -- level 0 tiled, a stripped thumbnail second (as SVS has
    it: not a level), the further levels tiled, each level the 2 x 2 box mean of the one before.  compression: 'jpeg' (YCbCr), 'lzw', 'deflate' or
    'none'.  -> list of the level arrays written."""
    lib = _lib()
    if lib is None:
        raise TiffError('libtiff not found')
    comp = {'none': 1, 'lzw': 5, 'jpeg': 7, 'deflate': 8, 'j2k-ycbcr': _APERIO_J2K_YCBCR, 'j2k-rgb': _APERIO_J2K_RGB}[compression]
    image = np.ascontiguousarray(np.asarray(image)[:, :, :3], np.uint8)
    pyr = [image]
    for _ in range(1, levels):
        a = pyr[-1]
        h, w = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
        a = a[:h, :w].astype(np.uint16)
        pyr.append(((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) // 4).astype(np.uint8))
    tif = lib.TIFFOpen(os.fsencode(path), b'w8' if image.nbytes > (1 << 31) else b'w')
    if not tif:
        raise TiffError(f'cannot create {path}')
    V, U32, I = ctypes.c_void_p(tif), ctypes.c_uint32, ctypes.c_int

    def common(a, comp_):
        lib.TIFFSetField(V, U32(_T['IMAGEWIDTH']), U32(a.shape[1])); lib.TIFFSetField(V, U32(_T['IMAGELENGTH']), U32(a.shape[0]))
        lib.TIFFSetField(V, U32(_T['BITSPERSAMPLE']), I(8)); lib.TIFFSetField(V, U32(_T['SAMPLESPERPIXEL']), I(3)); lib.TIFFSetField(V, U32(_T['PLANARCONFIG']), I(1))
        lib.TIFFSetField(V, U32(_T['COMPRESSION']), I(comp_))
        if comp_ == 7:
            lib.TIFFSetField(V, U32(_T['PHOTOMETRIC']), I(_PHOTOMETRIC_YCBCR)); lib.TIFFSetField(V, U32(_T['JPEGQUALITY']), I(quality))
            lib.TIFFSetField(V, U32(_T['JPEGCOLORMODE']), I(1))
        else:
            lib.TIFFSetField(V, U32(_T['PHOTOMETRIC']), I(_PHOTOMETRIC_RGB))

    def tiled(a, first):
        common(a, comp)
        lib.TIFFSetField(V, U32(_T['TILEWIDTH']), U32(tile)); lib.TIFFSetField(V, U32(_T['TILELENGTH']), U32(tile))
        if first and description:
            lib.TIFFSetField(V, U32(_T['IMAGEDESCRIPTION']), ctypes.c_char_p(description.encode('latin-1')))
        if not first:
            lib.TIFFSetField(V, U32(_T['SUBFILETYPE']), U32(1))       # FILETYPE_REDUCEDIMAGE
        buf = np.zeros((tile, tile, 3), np.uint8)
        for ty in range(0, a.shape[0], tile):
            for tx in range(0, a.shape[1], tile):
                blk = a[ty:ty + tile, tx:tx + tile]
                buf[:] = 0
                buf[:blk.shape[0], :blk.shape[1]] = blk
                if comp in (_APERIO_J2K_YCBCR, _APERIO_J2K_RGB):      # a raw, reversible JPEG 2000 codestream per tile, as Aperio scanners write (theirs are lossy)
                    import io
                    from PIL import Image
                    src = buf
                    if comp == _APERIO_J2K_YCBCR:                    # forward JFIF transform (libjpeg jccolor.c fixed point), components stored as they are
                        r, g, b = (buf[..., k].astype(np.int32) for k in range(3))
                        src = np.stack([(19595 * r + 38470 * g + 7471 * b + 32768) >> 16, ((-11059 * r - 21709 * g + 32768 * b + 32767) >> 16) + 128,
                                        ((32768 * r - 27439 * g - 5329 * b + 32767) >> 16) + 128], -1).clip(0, 255).astype(np.uint8)
                    bio = io.BytesIO()
                    Image.fromarray(src).save(bio, format='JPEG2000', no_jp2=True, irreversible=False, mct=0)
                    raw = bio.getvalue()
                    if lib.TIFFWriteRawTile(tif, lib.TIFFComputeTile(tif, tx, ty, 0, 0), raw, len(raw)) < 0:
                        raise TiffError('TIFFWriteRawTile failed')
                    continue
                if lib.TIFFWriteEncodedTile(tif, lib.TIFFComputeTile(tif, tx, ty, 0, 0), buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) < 0:
                    raise TiffError('TIFFWriteEncodedTile failed')
        lib.TIFFWriteDirectory(tif)

    tiled(pyr[0], True)
    if thumbnail:
        t = np.ascontiguousarray(pyr[-1][::2, ::2])
        common(t, 1)
        lib.TIFFSetField(V, U32(_T['ROWSPERSTRIP']), U32(t.shape[0]))
        lib.TIFFWriteEncodedStrip(tif, 0, t.ctypes.data_as(ctypes.c_void_p), t.nbytes)
        lib.TIFFWriteDirectory(tif)
    for a in pyr[1:]:
        tiled(a, False)
    lib.TIFFClose(tif)
    return pyr
