"""Slide folders: the `seg_and_patch` half of the reference's tools/infer_wsi.py (:117-306) behind the same arguments.

    source folder -> process list (`initialize_df`, tools/wsi_core/batch_process_utils.py:17-82; written to
    <save_dir>/process_list_autogen.csv before every slide and at the end, :159,291) -> per slide: auto-skip when its coordinate
    file exists (:168-171), tissue segmentation (`--seg`, :254-265), mask picture (:267-270), tile coordinates (`--patch`,
    :272-276 -> patches/<slide_id>), stitched picture (`--stitch`, :278-284), status column.

Differences, all forced by the image (no OpenSlide, no h5py, no OpenCV) and stated where they apply:
  * a slide is a level-0 RGB array (`.npy`, memory-mapped), a store directory (nuhtc_amd.tilestore) or a tiled TIFF / Aperio `.svs` read through
    libtiff (nuhtc_amd.tiffslide: the two OpenSlide formats whose pixels are plain TIFF); other files of the folder get status `failed_open`
    and the loop goes on (the reference would raise inside OpenSlide);
  * the pyramid is virtual and dyadic: level k is the [::2^k] view of the array -- for a TIFF slide the image at that downsample taken from the
    file's best pyramid level --, so `seg_level = -1` resolves to the level whose downsample is `seg_downsample` (64, what
    `get_best_level_for_downsample(64)` gives for a pyramid that has it);
  * the coordinate file is patches/<slide_id>.npz (`coords`, `patch_size`, `patch_level`, `name`: the datasets / attributes of
    the reference's .h5, tools/wsi_core/wsi_utils.py `initialize_hdf5_bag` / `save_hdf5`), written uncompressed -- AND, wherever an HDF5
    back end exists (h5py or the HDF5 C library, nuhtc_amd.h5coords), the reference's own patches/<slide_id>.h5 beside it; a patch folder the
    reference's tool made (only .h5 files) is read as it is: auto-skip and the slide loop accept either file;
  * the two pictures are drawn with PIL (outline width as `visWSI` computes it; not pixel-equal to cv2.drawContours).
"""
import math
import os
import time

import numpy as np

from . import tilestore, tissue

# tools/infer_wsi.py:378-382
SEG_PARAMS = {'seg_level': -1, 'sthresh': 8, 'mthresh': 7, 'close': 4, 'use_otsu': False, 'keep_ids': 'none', 'exclude_ids': 'none'}
FILTER_PARAMS = {'a_t': 100, 'a_h': 16, 'max_n_holes': 8}
VIS_PARAMS = {'vis_level': -1, 'line_thickness': 250}
PATCH_PARAMS = {'use_padding': True, 'contour_fn': 'four_pt'}


def default_parameters(preset=None, preset_dir='presets'):
    """The four parameter dicts of main() (:378-398); `preset` = a .csv under presets/ whose first row overrides them."""
    seg, flt, vis, pat = dict(SEG_PARAMS), dict(FILTER_PARAMS), dict(VIS_PARAMS), dict(PATCH_PARAMS)
    if preset:
        import pandas as pd
        df = pd.read_csv(os.path.join(preset_dir, preset))
        for d in (seg, flt, vis, pat):
            for key in d:
                d[key] = df.loc[0, key]
    return seg, flt, vis, pat


# the process list's columns after `slide_id`: (name, numpy dtype or None for "as given", which parameter dict, conversion) -- the schema
# of tools/wsi_core/batch_process_utils.py:17-82 minus the heatmap / save_patches columns infer_wsi.py never asks for
_PROCESS_COLUMNS = (
    ('process', np.uint8, None, lambda _: 1), ('status', None, None, lambda _: 'tbp'),
    ('seg_level', np.int8, 'seg', int), ('sthresh', np.uint8, 'seg', int), ('mthresh', np.uint8, 'seg', int), ('close', np.uint32, 'seg', int),
    ('use_otsu', bool, 'seg', bool), ('keep_ids', None, 'seg', None), ('exclude_ids', None, 'seg', None),
    ('a_t', np.float32, 'filter', int), ('a_h', np.float32, 'filter', int), ('max_n_holes', np.uint32, 'filter', int),
    ('vis_level', np.int8, 'vis', int), ('line_thickness', np.uint32, 'vis', int),
    ('use_padding', bool, 'patch', bool), ('contour_fn', None, 'patch', None))


def initialize_df(slides, seg_params, filter_params, vis_params, patch_params):
    """The process list (the reference's initialize_df): one row per slide, every column filled with the run's parameter -- same columns,
    order and dtypes, so the CSV text equals the reference's (pinned in tests/golden/process_list_autogen.csv).  Given a DataFrame (a
    user's --process_list) instead of a list of names, its own values win: only missing cells are filled, missing columns are appended."""
    import pandas as pd
    params = dict(seg=seg_params, filter=filter_params, vis=vis_params, patch=patch_params)
    given = slides if isinstance(slides, pd.DataFrame) else None
    ids = given.slide_id.values if given is not None else slides
    n = len(ids)
    cols = {'slide_id': ids}
    for name, dtype, group, conv in _PROCESS_COLUMNS:
        v = params[group][name] if group else None
        v = conv(v) if conv else v
        cols[name] = np.full(n, v) if dtype is None else np.full(n, v, dtype=dtype)
    defaults = pd.DataFrame(cols)
    if given is None:
        return defaults
    for name in cols:
        if name not in given.columns:
            given[name] = cols[name]                         # appended after the user's columns, in schema order
        else:
            given[name] = given[name].where(given[name].notna(), defaults[name])
    return given


def coords_path(patch_save_dir, slide_id):
    """patches/<slide_id>.npz -- the role of the reference's patches/<slide_id>.h5."""
    return os.path.join(patch_save_dir, slide_id + '.npz')


def h5_path(patch_save_dir, slide_id):
    """patches/<slide_id>.h5 -- the reference's coordinate file (tools/infer_wsi.py:168,282,447)."""
    return os.path.join(patch_save_dir, slide_id + '.h5')


def save_coords(path, coords, patch_size, patch_level, name, level_dim=None, downsample=None):
    """patches/<id>.npz and, with an HDF5 back end, the reference's patches/<id>.h5 with the attributes WholeSlideImage.process_contour gives it
    (:483-489; `level_dim` = (W, H) of the patch level, `downsample` its level_downsamples entry)."""
    coords = np.asarray(coords, np.int64).reshape(-1, 2)
    np.savez(path, coords=coords, patch_size=np.int64(patch_size), patch_level=np.int64(patch_level), name=np.str_(name))
    from . import h5coords
    if h5coords.available():
        attrs = dict(patch_size=int(patch_size), patch_level=int(patch_level))
        if downsample is not None:
            attrs['downsample'] = tuple(float(v) for v in downsample)
        if level_dim is not None:
            attrs['downsampled_level_dim'] = tuple(int(v) for v in level_dim)
            attrs['level_dim'] = tuple(int(v) for v in level_dim)
        attrs['name'] = str(name)
        attrs['save_path'] = os.path.dirname(path)
        try:
            h5coords.write_coords(os.path.splitext(path)[0] + '.h5', coords, attrs)
        except (h5coords.H5Error, OSError) as e:       # the .npz above is what this package reads: the reference's twin is a courtesy
            import warnings
            warnings.warn(f'{path}: the .h5 twin of the coordinate file was not written ({e})')


def has_coords(patch_save_dir, slide_id):
    return os.path.isfile(coords_path(patch_save_dir, slide_id)) or os.path.isfile(h5_path(patch_save_dir, slide_id))


def load_coords(patch_save_dir, slide_id):
    """-> (coords int64 (n, 2), patch_size, patch_level) of a slide: the .npz, else the reference's .h5 (Whole_Slide_Bag_FP, WholeSlideImage.py:862-865)."""
    p = coords_path(patch_save_dir, slide_id)
    if os.path.isfile(p):
        z = np.load(p)
        return np.asarray(z['coords'], np.int64).reshape(-1, 2), int(z['patch_size']), int(z['patch_level']) if 'patch_level' in z.files else 0
    from . import h5coords
    r = h5coords.read_coords(h5_path(patch_save_dir, slide_id))
    return r['coords'], int(r['attrs']['patch_size']), int(r['attrs'].get('patch_level', 0))


def open_array_slide(path):
    """-> the slide as an (H, W, 3) uint8 array-like: a memory-mapped `.npy` slide, a store directory's slide.npy, or a tiled TIFF / Aperio SVS
    behind nuhtc_amd.tiffslide.TiffSlide (same indexing; `slide[::s, ::s]` comes from its pyramid)."""
    if os.path.isdir(path):
        path = os.path.join(path, 'slide.npy')
    from . import tiffslide
    if tiffslide.is_tiff_slide(path):
        return tiffslide.TiffSlide(path)
    return tilestore.open_slide(path)


def _ids(v):
    v = str(v)
    return [] if v == 'none' or len(v) == 0 else [int(t) for t in v.split(',')]


def _level_of(downsample):
    lvl = int(round(math.log2(downsample)))
    if downsample < 1 or 2 ** lvl != downsample:
        raise ValueError(f'seg_downsample must be a power of two (virtual dyadic pyramid), got {downsample}')
    return lvl


def is_pyramid(slide):
    """A slide with real pyramid levels (nuhtc_amd.tiffslide.TiffSlide): levels are then the FILE's levels, as in the reference; an array
    slide has the virtual dyadic pyramid slide[::2**k, ::2**k]."""
    return hasattr(slide, 'level_dimensions') and hasattr(slide, 'level_image')


def level_downsample(slide, level):
    """(x, y) downsample of a pyramid level the way `_assertLevelDownsamples` (WholeSlideImage.py:378-386) states it."""
    w0, h0 = slide.level_dimensions[0]
    w, h = slide.level_dimensions[level]
    return (w0 / float(w), h0 / float(h))


def vis_mask(slide, contours, holes, level, line_thickness=250, color=(0, 255, 0), hole_color=(0, 0, 255)):
    """`visWSI` (WholeSlideImage.py:201-256): the level image with the tissue contours (green) and their holes (blue) outlined;
    outline width int(line_thickness * sqrt(scale_x * scale_y)) like :221."""
    from PIL import Image, ImageDraw
    if is_pyramid(slide):
        dx, dy = level_downsample(slide, int(level))
        img = Image.fromarray(np.ascontiguousarray(np.asarray(slide.level_image(int(level)))[:, :, :3]))
    else:
        dx = dy = 2 ** int(level)
        img = Image.fromarray(np.ascontiguousarray(np.asarray(slide[::dx, ::dy, :3])))
    width = max(1, int(line_thickness * math.sqrt((1 / dx) * (1 / dy))))
    dr = ImageDraw.Draw(img)

    def outline(c, col):
        pts = [(int(x * (1 / dx)), int(y * (1 / dy))) for x, y in np.asarray(c).reshape(-1, 2)]      # scaleContourDim: astype(int32)
        if len(pts) > 1:
            dr.line(pts + pts[:1], fill=col, width=width)
        elif pts:
            dr.point(pts, fill=col)
    for c in contours or []:
        outline(c, color)
    for hs in holes or []:
        for h in hs:
            outline(h, hole_color)
    return img


def stitch_coords(slide, coords, patch_size, downscale=64, bg_color=(0, 0, 0)):
    """`StitchCoords` + `DrawMapFromCoords` (wsi_utils.py:259-293,200-225): the tiles of the coordinate file pasted, down-scaled,
    onto a black canvas of the level nearest `downscale`."""
    from PIL import Image
    if is_pyramid(slide):                                                 # the file's level nearest `downscale` (wsi_utils.py:261)
        lvl = slide.get_best_level_for_downsample(downscale)
        dsx, dsy = level_downsample(slide, lvl)
        lv = np.asarray(slide.level_image(lvl))[:, :, :3]
    else:
        dsx = dsy = 2 ** _level_of(downscale)
        lv = np.asarray(slide[::dsy, ::dsx, :3])
    h, w = lv.shape[:2]
    canvas = np.zeros((h, w, 3), np.uint8)
    canvas[:] = bg_color
    ps = int(math.ceil(patch_size / dsx))
    for x, y in np.asarray(coords, np.int64).reshape(-1, 2):
        cx, cy = int(math.ceil(x / dsx)), int(math.ceil(y / dsy))
        sub = lv[cy:cy + ps, cx:cx + ps]                                  # read_region(coord, vis_level, patch_size)
        canvas[cy:cy + sub.shape[0], cx:cx + sub.shape[1]] = sub
    return Image.fromarray(canvas)


def seg_and_patch(source, save_dir, patch_save_dir, mask_save_dir, stitch_save_dir, patch_size=256, step_size=256,
                  seg_params=None, filter_params=None, vis_params=None, patch_params=None, patch_level=0, use_default_params=False,
                  seg=False, save_mask=True, stitch=False, patch=False, no_auto_skip=False, process_list=None,
                  slides=None, seg_downsample=64, log=print):
    """tools/infer_wsi.py:117-306 with the same arguments (+ `slides`: an explicit list of file names instead of the folder listing,
    `seg_downsample`: the downsample `seg_level = -1` / `vis_level = -1` resolve to).  Returns (seg_times, patch_times)."""
    import pandas as pd
    seg_params = dict(SEG_PARAMS if seg_params is None else seg_params)
    filter_params = dict(FILTER_PARAMS if filter_params is None else filter_params)
    vis_params = dict(VIS_PARAMS if vis_params is None else vis_params)
    patch_params = dict(PATCH_PARAMS if patch_params is None else patch_params)
    if patch_level != 0:
        raise ValueError('--patch_level: array slides have one level; only patch_level 0 is supported')
    if slides is None:
        slides = sorted(os.listdir(source))
        # the reference keeps regular files only (:136); a store directory (slide.npy + coords.npy) is this build's other slide form
        slides = [s for s in slides if os.path.isfile(os.path.join(source, s)) or os.path.isfile(os.path.join(source, s, 'slide.npy'))]
    if process_list is None:
        df = initialize_df(slides, seg_params, filter_params, vis_params, patch_params)
    else:
        df = initialize_df(pd.read_csv(process_list), seg_params, filter_params, vis_params, patch_params)
    if 'a' in df.keys():
        raise NotImplementedError('legacy segmentation csv files (column "a") are not supported')
    process_stack = df[df['process'] == 1]
    total = len(process_stack)
    seg_times = patch_times = stitch_times = 0.
    auto_level = _level_of(seg_downsample)
    for i in range(total):
        df.to_csv(os.path.join(save_dir, 'process_list_autogen.csv'), index=False)
        idx = process_stack.index[i]
        slide = process_stack.loc[idx, 'slide_id']
        log('\n\nprogress: {:.2f}, {}/{}'.format(i / total, i, total))
        log('processing {}'.format(slide))
        df.loc[idx, 'process'] = 0
        slide_id, _ = os.path.splitext(slide)
        cpath = coords_path(patch_save_dir, slide_id)
        if not no_auto_skip and has_coords(patch_save_dir, slide_id):
            log('{} already exist in destination location, skipped'.format(slide_id))
            df.loc[idx, 'status'] = 'already_exist'
            continue
        try:
            img = open_array_slide(os.path.join(source, slide))
        except Exception as e:                                     # not an array slide (OpenSlide formats cannot be read here)
            log('cannot open {} as an array slide ({}): skipped'.format(slide, e))
            df.loc[idx, 'status'] = 'failed_open'
            continue
        H, W = img.shape[:2]
        cur_vis = {k: (vis_params if use_default_params else df.loc[idx])[k] for k in vis_params}
        cur_filter = {k: (filter_params if use_default_params else df.loc[idx])[k] for k in filter_params}
        cur_seg = {k: (seg_params if use_default_params else df.loc[idx])[k] for k in seg_params}
        cur_patch = {k: (patch_params if use_default_params else df.loc[idx])[k] for k in patch_params}
        # level -1 = "the level nearest 64x" (tools/infer_wsi.py:213-229): on a pyramid file the FILE's level from
        # get_best_level_for_downsample (level 0 for a single-level file), whatever its true downsample is -- a typical Aperio pyramid
        # (1 / 4 / 16 / 32) segments at 32x; on an array slide the virtual dyadic level of exactly `seg_downsample`
        pyr = is_pyramid(img)
        if pyr:
            best = 0 if len(img.level_dimensions) == 1 else img.get_best_level_for_downsample(seg_downsample)
        if cur_vis['vis_level'] < 0:
            cur_vis['vis_level'] = best if pyr else auto_level
        if cur_seg['seg_level'] < 0:
            cur_seg['seg_level'] = best if pyr else auto_level
        keep_ids, exclude_ids = _ids(cur_seg['keep_ids']), _ids(cur_seg['exclude_ids'])
        if pyr:
            if not 0 <= int(cur_seg['seg_level']) < len(img.level_dimensions) or not 0 <= int(cur_vis['vis_level']) < len(img.level_dimensions):
                log('{}: seg_level / vis_level outside the {} level(s) of the file, aborting'.format(slide_id, len(img.level_dimensions)))
                df.loc[idx, 'status'] = 'failed_seg'
                continue
            w, h = img.level_dimensions[int(cur_seg['seg_level'])]
            sds = level_downsample(img, int(cur_seg['seg_level']))
        else:
            sds = 2 ** int(cur_seg['seg_level'])
            w, h = -(-W // sds), -(-H // sds)
        if w * h > 1e8:
            log('level_dim {} x {} is likely too large for successful segmentation, aborting'.format(w, h))
            df.loc[idx, 'status'] = 'failed_seg'
            continue
        df.loc[idx, 'vis_level'] = cur_vis['vis_level']
        df.loc[idx, 'seg_level'] = cur_seg['seg_level']
        conts = holes = None
        seg_time = -1
        if seg:
            t0 = time.time()
            conts, holes = tissue.segment_tissue(img, scale=sds, level_image=img.level_image(int(cur_seg['seg_level'])) if pyr else None,
                                                 sthresh=int(cur_seg['sthresh']), mthresh=int(cur_seg['mthresh']),
                                                 close=int(cur_seg['close']), use_otsu=bool(cur_seg['use_otsu']),
                                                 filter_params=dict(a_t=cur_filter['a_t'], a_h=cur_filter['a_h'], max_n_holes=int(cur_filter['max_n_holes'])),
                                                 keep_ids=keep_ids, exclude_ids=exclude_ids)
            seg_time = time.time() - t0
        if save_mask:
            vis_mask(img, conts, holes, int(cur_vis['vis_level']), int(cur_vis['line_thickness'])).save(os.path.join(mask_save_dir, slide_id + '.png'))
        patch_time = -1
        if patch:
            t0 = time.time()
            if conts is None:
                # the reference cannot patch without --seg (contours_tissue is None, process_contours raises); here the whole slide
                # is tiled on the grid np.arange(0, size, step)
                coords = tilestore.grid_coords(H, W, step_size)
            else:
                parts = [tissue.contour_coords(c, hs, (W, H), patch_size, step_size, str(cur_patch['contour_fn']), bool(cur_patch['use_padding']))
                         for c, hs in zip(conts, holes)]
                coords = np.concatenate(parts, 0) if parts else np.zeros((0, 2), np.int64)
            if len(coords):                                        # the reference creates the .h5 with the first contour that yields tiles (:397-403)
                save_coords(cpath, coords, patch_size, patch_level, slide_id, level_dim=(W >> patch_level, H >> patch_level),
                            downsample=(float(1 << patch_level), float(1 << patch_level)))
            log('tissue segmentation: {} contour(s), {} tiles'.format(0 if conts is None else len(conts), len(coords)))
            patch_time = time.time() - t0
        stitch_time = -1
        if stitch and has_coords(patch_save_dir, slide_id):
            t0 = time.time()
            sc, sps, _ = load_coords(patch_save_dir, slide_id)
            stitch_coords(img, sc, sps, downscale=64).save(os.path.join(stitch_save_dir, slide_id + '.jpg'))
            stitch_time = time.time() - t0
        log('segmentation took {} seconds'.format(seg_time))
        log('patching took {} seconds'.format(patch_time))
        log('stitching took {} seconds'.format(stitch_time))
        df.loc[idx, 'status'] = 'processed'
        seg_times += seg_time
        patch_times += patch_time
        stitch_times += stitch_time
    if total:
        seg_times /= total
        patch_times /= total
        stitch_times /= total
    df.to_csv(os.path.join(save_dir, 'process_list_autogen.csv'), index=False)
    log('average segmentation time in s per slide: {}'.format(seg_times))
    log('average patching time in s per slide: {}'.format(patch_times))
    log('average stiching time in s per slide: {}'.format(stitch_times))
    return seg_times, patch_times


def slide_list(save_dir):
    """`Dataset_All_Bags` (WholeSlideImage.py:900-909): the slide_id column of <save_dir>/process_list_autogen.csv."""
    import pandas as pd
    return [str(s) for s in pd.read_csv(os.path.join(save_dir, 'process_list_autogen.csv'))['slide_id']]
