"""The command lines of the reference's entry points are accepted verbatim (SURVEY 8b "CLI: same flags/defaults").

`tests/golden/cli_reference_args.json` is every `add_argument` call of the reference's tools/infer.py, tools/infer_wsi.py,
tools/infer_patch.py and tools/nuclei_merge.py, read off their source by oracle/ref_harness/make_cli_golden.py; the parsers of this
repo's tools must declare each of them with the same flags, dest, default, type, action and required-ness (extras are allowed),
and must parse the README's own invocations.  `tests/golden/process_list_autogen.csv` is the text the reference's own `initialize_df`
+ `to_csv` give; nuhtc_amd.slides must write the same bytes."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
TYPES = {'int': int, 'float': float, 'str': str}


def _tool(name):
    spec = importlib.util.spec_from_file_location('tool_' + name.replace('.', '_'), os.path.join(ROOT, 'tools', name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('tool', ['infer.py', 'infer_wsi.py', 'infer_patch.py', 'nuclei_merge.py'])
def test_parser_declares_every_reference_argument(tool):
    table = json.load(open(os.path.join(GOLD, 'cli_reference_args.json')))['tools/' + tool]
    parser = _tool(tool).build_parser()
    assert parser.allow_abbrev is False            # `--patch` must not be read as a prefix of `--patch_size`
    by_flag = {}
    for a in parser._actions:
        for f in (a.option_strings or [a.dest]):
            by_flag[f] = a
    assert len(table) >= 5
    for row in table:
        acts = {id(by_flag[f]): by_flag[f] for f in row['flags'] if f in by_flag}
        assert len(acts) == 1 and all(f in by_flag for f in row['flags']), (tool, row['flags'])
        a = next(iter(acts.values()))
        if row['flags'][0].startswith('-'):
            assert sorted(a.option_strings) == sorted(row['flags']), row
        assert a.default == row.get('default', False if row.get('action') == 'store_true' else None), (row, a.default)
        if 'dest' in row:
            assert a.dest == row['dest']
        if 'type' in row:
            assert a.type is TYPES[row['type']], row
        if row.get('action') == 'store_true':
            assert a.nargs == 0 and a.const is True
        else:
            assert a.nargs is None
        assert bool(a.required) == bool(row.get('required', not row['flags'][0].startswith('-')))


def test_readme_command_lines_parse():
    """README.md:55-63 (docker) and :221-223 of the reference, token for token after the script name."""
    wsi = _tool('infer_wsi.py')
    a = wsi.parse_args('/data/wsi models/htc_lite_PanNuke_infer.py models/pannuke.pth --patch --seg --stitch --patch_size 256 --step_size 192 '
                       '--batch_size 16 --save_dir /data/wsi_infer --mode qupath'.split())
    assert (a.patch, a.seg, a.stitch, a.patch_size, a.step_size, a.batch_size, a.save_dir, a.mode) == (True, True, True, 256, 192, 16, '/data/wsi_infer', 'qupath')
    assert (a.margin, a.min_area, a.mag, a.num_workers, a.slide_ext, a.score_thr, a.det, a.no_auto_skip) == (0, 10, 40, 8, '.svs', 0.35, False, False)
    b = wsi.parse_args('demo/wsi configs/nuhtc/htc_lite_swin_pytorch_fpn_PanNuke_seasaw_CAS.py models/pannuke.pth --patch --seg --stitch '
                       '--patch_size 256 --step_size 192 --margin 1 --min_area 10 --batch_size 32 --save_dir demo/wsi_infer --mode qupath'.split())
    assert (b.margin, b.batch_size, b.patch) == (1, 32, True)
    # tools/infer_wsi.py:308 (the comment above the reference's parser) and the --magnification alias
    c = wsi.parse_args('demo/wsi m.py m.pth --patch --seg --stitch --patch_size 256 --step_size 224 --save_dir demo/wsi_res --det --magnification 20'.split())
    assert (c.det, c.step_size, c.mag, c.batch_size) == (True, 224, 20, 32)
    inf = _tool('infer.py')
    d = inf.parse_args('demo/imgs models/htc_lite_PanNuke_infer.py models/pannuke.pth'.split())
    assert (d.score_thr, d.output, d.device, d.async_test) == (0.35, None, 'cuda:0', False)
    m = _tool('nuclei_merge.py').parse_args('--geojson a.geojson --overlap_threshold 0.05 --merge_strategy probability'.split())
    assert (m.overlap_threshold, m.output_name, m.uniform_classification) == (0.05, None, False)


def _two_slides(tmp_path):
    from test_tissue import tissue_slide_with_holes
    src = tmp_path / 'wsi'
    src.mkdir()
    img, *_ = tissue_slide_with_holes(H=768, W=1024)
    np.save(src / 'a.npy', img)
    np.save(src / 'b.npy', img[:, ::-1].copy())
    (src / 'c.svs').write_bytes(b'not an array slide')
    return src, img


def test_seg_and_patch_writes_the_reference_process_list_and_coordinate_files(tmp_path):
    """nuhtc_amd.slides.seg_and_patch (tools/infer_wsi.py:117-306) over a folder: process_list_autogen.csv byte-equal to what the
    reference's own initialize_df + the loop's assignments give (golden), coordinate files = tissue tiles, pictures written,
    auto-skip on the second run, --no_auto_skip redoes, a process list restricts the run."""
    from nuhtc_amd import slides, tissue
    src, img = _two_slides(tmp_path)
    out = tmp_path / 'out'
    dirs = dict(source=str(src), save_dir=str(out), patch_save_dir=str(out / 'patches'), mask_save_dir=str(out / 'masks'), stitch_save_dir=str(out / 'stitches'))
    for k, v in dirs.items():
        if k != 'source':
            os.makedirs(v)
    np.savez(out / 'patches' / 'b.npz', coords=np.zeros((2, 2), np.int64), patch_size=64, patch_level=0, name='b')     # b was patched before
    log = []
    seg, flt, vis, pat = slides.default_parameters()
    slides.seg_and_patch(**dirs, seg_params=seg, filter_params=flt, vis_params=vis, patch_params=pat, patch_size=64, step_size=64, seg=True,
                         patch=True, stitch=True, seg_downsample=64, log=lambda *a: log.append(' '.join(str(x) for x in a)))
    got = open(out / 'process_list_autogen.csv').read()
    assert got == open(os.path.join(GOLD, 'process_list_autogen.csv')).read().replace('c.svs,1,tbp,-1,8,7,4,False,none,none,100.0,16.0,8,-1',
                                                                                     'c.svs,0,failed_open,-1,8,7,4,False,none,none,100.0,16.0,8,-1')
    assert any('b already exist in destination location, skipped' in l for l in log)
    assert slides.slide_list(str(out)) == ['a.npy', 'b.npy', 'c.svs']
    # at 64x the 768 x 1024 slide is 12 x 16 pixels: nothing passes the area filter -> no coordinate file for a (as the reference: no .h5)
    assert not os.path.exists(out / 'patches' / 'a.npz') and os.path.exists(out / 'masks' / 'a.png')
    # the same folder at a finer segmentation level, everything redone
    slides.seg_and_patch(**dirs, seg_params=seg, filter_params=flt, vis_params=vis, patch_params=pat, patch_size=64, step_size=64, seg=True,
                         patch=True, stitch=True, no_auto_skip=True, seg_downsample=8, log=lambda *a: None)
    z = np.load(out / 'patches' / 'a.npz')
    want, conts, holes = tissue.tissue_tile_coords(img, 64, 64, scale=8)
    assert int(z['patch_size']) == 64 and int(z['patch_level']) == 0 and str(z['name']) == 'a' and len(want) > 50
    assert np.array_equal(z['coords'], want)
    zb = np.load(out / 'patches' / 'b.npz')
    assert len(zb['coords']) > 50 and not np.array_equal(zb['coords'], want)
    from PIL import Image
    m = np.asarray(Image.open(out / 'masks' / 'a.png'))
    assert m.shape == (96, 128, 3) and (m == (0, 255, 0)).all(-1).sum() > 50          # level 3 picture with green outlines
    st = np.asarray(Image.open(out / 'stitches' / 'a.jpg'))
    assert st.shape == (12, 16, 3) and st.max() > 100
    rows = open(out / 'process_list_autogen.csv').read().splitlines()
    assert rows[1].startswith('a.npy,0,processed,3,') and ',3,250,True,four_pt' in rows[1]
    # a process list: only the rows with process == 1 are touched
    import pandas as pd
    pd.DataFrame({'slide_id': ['a.npy', 'b.npy'], 'process': [0, 1]}).to_csv(out / 'todo.csv', index=False)
    os.remove(out / 'patches' / 'a.npz')
    slides.seg_and_patch(**dirs, seg_params=seg, filter_params=flt, vis_params=vis, patch_params=pat, patch_size=64, step_size=64, seg=True,
                         patch=True, no_auto_skip=True, seg_downsample=8, process_list=str(out / 'todo.csv'), log=lambda *a: None)
    assert not os.path.exists(out / 'patches' / 'a.npz')
    assert slides.slide_list(str(out)) == ['a.npy', 'b.npy']


def test_preset_overrides_parameters(tmp_path):
    from nuhtc_amd import slides
    (tmp_path / 'presets').mkdir()
    (tmp_path / 'presets' / 'p.csv').write_text('seg_level,sthresh,mthresh,close,use_otsu,keep_ids,exclude_ids,a_t,a_h,max_n_holes,vis_level,line_thickness,use_padding,contour_fn\n'
                                                '-1,15,11,2,True,none,none,1,1,2,-1,50,True,four_pt_hard\n')
    seg, flt, vis, pat = slides.default_parameters('p.csv', preset_dir=str(tmp_path / 'presets'))
    assert (seg['sthresh'], seg['mthresh'], seg['close'], bool(seg['use_otsu'])) == (15, 11, 2, True)
    assert (flt['a_t'], flt['max_n_holes'], vis['line_thickness'], pat['contour_fn']) == (1, 2, 50, 'four_pt_hard')
