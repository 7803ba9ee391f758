"""The reference's coordinate files patches/<slide_id>.h5 (tools/wsi_core/WholeSlideImage.py:388-406,481-492 + wsi_utils.save_hdf5 :66-85 write
them, Whole_Slide_Bag_FP :862-865 reads them): nuhtc_amd.h5coords through the HDF5 C library (no h5py for the interpreter the suite runs on).
Pinned on a golden file the reference's own save_hdf5 wrote under the real h5py (oracle/ref_harness/make_h5_golden.py); the writer is also
checked against the HDF5 project's `h5dump` and, where the image's Python 3.9 is present, by h5py itself; the reader also against files made by
direct C-API calls in the forms other writers produce (32-bit integers, fixed-length ASCII strings, H5Dset_extent appends)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from nuhtc_amd import h5coords as H

pytestmark = pytest.mark.skipif(not H.available(), reason='no HDF5 back end (h5py or libhdf5 >= 1.10) in this environment')

ATTRS = dict(patch_size=256, patch_level=0, downsample=(1.0, 1.0), downsampled_level_dim=(40000, 30000), level_dim=(40000, 30000), name='TCGA-xx-ä', save_path='/data/out/patches')


def _h5dump():
    for c in (shutil.which('h5dump'), '/opt/conda/bin/h5dump', '/usr/bin/h5dump'):
        if c and os.path.exists(c):
            return c
    return None


def test_round_trip_and_reference_reader_fields(tmp_path):
    rng = np.random.default_rng(0)
    coords = np.stack([rng.integers(0, 2 ** 36, 500), rng.integers(0, 40000, 500)], 1).astype(np.int64)     # level-0 origins beyond 32 bits too
    p = H.write_coords(str(tmp_path / 's.h5'), coords, ATTRS)
    r = H.read_coords(p)
    assert r['coords'].dtype == np.int64 and np.array_equal(r['coords'], coords)
    assert r['chunks'] == (1, 2) and tuple(r['maxshape']) == (None, 2)                    # save_hdf5: chunk_shape (1,) + shape[1:], maxshape (None,) + shape[1:]
    a = r['attrs']
    assert set(a) == set(H.ATTR_ORDER)
    assert int(a['patch_size']) == 256 and int(a['patch_level']) == 0                     # what Whole_Slide_Bag_FP takes from the file
    assert np.array_equal(a['downsample'], [1.0, 1.0]) and np.asarray(a['downsample']).dtype == np.float64
    assert np.array_equal(a['level_dim'], [40000, 30000]) and np.array_equal(a['downsampled_level_dim'], [40000, 30000])
    assert a['name'] == 'TCGA-xx-ä' and a['save_path'] == '/data/out/patches'
    # an empty list of coordinates and a missing / foreign file
    r0 = H.read_coords(H.write_coords(str(tmp_path / 'e.h5'), np.zeros((0, 2), np.int64), dict(patch_size=64, patch_level=0)))
    assert r0['coords'].shape == (0, 2) and int(r0['attrs']['patch_size']) == 64
    (tmp_path / 'bad.h5').write_bytes(b'not hdf5')
    for bad in ('missing.h5', 'bad.h5'):
        with pytest.raises((H.H5Error, OSError)):
            H.read_coords(str(tmp_path / bad))


@pytest.mark.skipif(_h5dump() is None, reason='h5dump not installed')
def test_written_file_is_what_h5py_would_have_written(tmp_path):
    """`h5dump -p` (the HDF5 project's tool) of a written file: the dataset and attribute types, spaces and layout h5py gives
    create_dataset(shape, maxshape=(None, 2), chunks=(1, 2), dtype=int64) and attrs[...] = int / tuple of float / tuple of int / str."""
    coords = np.array([[0, 0], [256, 0], [512, 768]], np.int64)
    p = H.write_coords(str(tmp_path / 's.h5'), coords, ATTRS)
    out = subprocess.run([_h5dump(), '-p', p], capture_output=True, text=True, check=True).stdout
    flat = ' '.join(out.split())
    assert 'DATASET "coords" { DATATYPE H5T_STD_I64LE DATASPACE SIMPLE { ( 3, 2 ) / ( H5S_UNLIMITED, 2 ) } STORAGE_LAYOUT { CHUNKED ( 1, 2 )' in flat
    assert 'FILTERS { NONE }' in flat and '(0,0): 0, 0, (1,0): 256, 0, (2,0): 512, 768' in flat
    for name in ('patch_size', 'patch_level'):
        assert f'ATTRIBUTE "{name}" {{ DATATYPE H5T_STD_I64LE DATASPACE SCALAR' in flat
    assert 'ATTRIBUTE "downsample" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 2 ) / ( 2 ) } DATA { (0): 1, 1 }' in flat
    assert 'ATTRIBUTE "level_dim" { DATATYPE H5T_STD_I64LE DATASPACE SIMPLE { ( 2 ) / ( 2 ) } DATA { (0): 40000, 30000 }' in flat
    for name in ('name', 'save_path'):
        assert f'ATTRIBUTE "{name}" {{ DATATYPE H5T_STRING {{ STRSIZE H5T_VARIABLE; STRPAD H5T_STR_NULLTERM; CSET H5T_CSET_UTF8; CTYPE H5T_C_S1; }} DATASPACE SCALAR' in flat


@pytest.mark.skipif(H._lib() is None, reason='needs the HDF5 C library')
def test_reader_on_files_of_other_writers(tmp_path):
    """A file made by direct C-API calls the way save_hdf5 grows it -- created with the first contour's rows, then resized and written at the
    end per further contour (`dset.resize(len + n)`, `dset[-n:] = val`) -- with 32-bit integers and fixed-length ASCII string attributes."""
    lib = H._lib()
    hid, hsz, I, P = ctypes.c_int64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p
    HP = ctypes.POINTER(hsz)
    lib.H5Dset_extent.restype, lib.H5Dset_extent.argtypes = I, [hid, HP]
    lib.H5Sselect_hyperslab.restype, lib.H5Sselect_hyperslab.argtypes = I, [hid, I, HP, HP, HP, HP]
    i32 = hid.in_dll(lib, 'H5T_STD_I32LE_g').value
    nat32 = hid.in_dll(lib, 'H5T_NATIVE_INT32_g').value
    parts = [np.arange(10, dtype=np.int32).reshape(5, 2) * 256, np.arange(10, 16, dtype=np.int32).reshape(3, 2) * 256, np.arange(16, 30, dtype=np.int32).reshape(7, 2) * 256]
    path = str(tmp_path / 'other.h5')
    f = lib.H5Fcreate(path.encode(), 2, 0, 0)
    assert f >= 0
    space = lib.H5Screate_simple(2, (hsz * 2)(5, 2), (hsz * 2)((1 << 64) - 1, 2))
    pl = lib.H5Pcreate(lib._ids['H5P_DATASET_CREATE'])
    lib.H5Pset_chunk(pl, 2, (hsz * 2)(1, 2))
    d = lib.H5Dcreate2(f, b'coords', i32, space, 0, pl, 0)
    assert d >= 0
    lib.H5Sclose(space); lib.H5Pclose(pl)
    assert lib.H5Dwrite(d, nat32, 0, 0, 0, parts[0].ctypes.data_as(P)) >= 0
    n = 5
    for part in parts[1:]:
        assert lib.H5Dset_extent(d, (hsz * 2)(n + len(part), 2)) >= 0
        fs = lib.H5Dget_space(d)
        assert lib.H5Sselect_hyperslab(fs, 0, (hsz * 2)(n, 0), None, (hsz * 2)(len(part), 2), None) >= 0      # H5S_SELECT_SET
        ms = lib.H5Screate_simple(2, (hsz * 2)(len(part), 2), None)
        assert lib.H5Dwrite(d, nat32, ms, fs, 0, part.ctypes.data_as(P)) >= 0
        lib.H5Sclose(ms); lib.H5Sclose(fs)
        n += len(part)
    # attributes: 32-bit scalars, a fixed-length ASCII string
    sc = lib.H5Screate(0)
    for name, v in ((b'patch_size', 512), (b'patch_level', 1)):
        a = lib.H5Acreate2(d, name, i32, sc, 0, 0)
        assert lib.H5Awrite(a, nat32, ctypes.byref(ctypes.c_int32(v))) >= 0
        lib.H5Aclose(a)
    st = lib.H5Tcopy(lib._ids['H5T_C_S1'])
    lib.H5Tset_size(st, 8)
    a = lib.H5Acreate2(d, b'name', st, sc, 0, 0)
    assert lib.H5Awrite(a, st, ctypes.create_string_buffer(b'slide7\0\0', 8)) >= 0
    lib.H5Aclose(a); lib.H5Tclose(st); lib.H5Sclose(sc); lib.H5Dclose(d); lib.H5Fclose(f)
    r = H.read_coords(path)
    assert r['coords'].dtype == np.int64 and np.array_equal(r['coords'], np.concatenate(parts, 0))
    assert int(r['attrs']['patch_size']) == 512 and int(r['attrs']['patch_level']) == 1 and r['attrs']['name'] == 'slide7'
    assert r['chunks'] == (1, 2) and tuple(r['maxshape']) == (None, 2)


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_file_written_by_the_reference_with_h5py(tmp_path):
    """tests/golden/coords_reference.h5 was written by the reference's OWN save_hdf5 under the real h5py (oracle/ref_harness/make_h5_golden.py:
    mode 'w' with the attribute dict, then two appends, as process_contours does); coords_reference.json is what h5py reports for it.  The
    reader returns the same coordinates, attributes and layout; a file written here from those values has the same `h5dump -p` description."""
    import json
    want = json.load(open(os.path.join(GOLD, 'coords_reference.json')))
    r = H.read_coords(os.path.join(GOLD, 'coords_reference.h5'))
    assert r['coords'].dtype == np.int64 and r['coords'].tolist() == want['coords'] and list(r['coords'].shape) == want['shape'] == [85, 2]
    assert list(r['chunks']) == want['chunks'] == [1, 2] and list(r['maxshape']) == want['maxshape'] == [None, 2]
    assert set(r['attrs']) == set(want['attrs']) == set(H.ATTR_ORDER)
    for k, v in want['attrs'].items():
        got = r['attrs'][k]
        assert (got.tolist() if hasattr(got, 'tolist') else got) == v, k
        assert (np.asarray(got).dtype.name if not isinstance(got, str) else 'str') == want['attr_dtypes'][k], k
    assert int(r['attrs']['patch_level']) == want['patch_level'] and int(r['attrs']['patch_size']) == want['patch_size']
    if _h5dump() is not None:
        a = want['attrs']
        mine = H.write_coords(str(tmp_path / 'mine.h5'), np.array(want['coords']), {k: (tuple(a[k]) if isinstance(a[k], list) else a[k]) for k in H.ATTR_ORDER})
        dump = lambda p: subprocess.run([_h5dump(), '-p', p], capture_output=True, text=True, check=True).stdout.splitlines()[1:]
        assert dump(mine) == dump(os.path.join(GOLD, 'coords_reference.h5'))


@pytest.mark.skipif(not os.path.exists('/opt/conda/bin/python3.9'), reason="the image's second interpreter (the one that has h5py) is not here")
def test_written_file_read_by_the_real_h5py(tmp_path):
    """The other direction: a file written here, opened by h5py 3.3 the way Whole_Slide_Bag_FP opens it (WholeSlideImage.py:862-865, 890-891)."""
    import json
    probe = subprocess.run(['/opt/conda/bin/python3.9', '-c', 'import h5py'], capture_output=True)
    if probe.returncode != 0:
        pytest.skip('no h5py in /opt/conda/bin/python3.9')
    coords = np.stack([np.arange(40) * 192, np.arange(40)[::-1] * 192 + 2 ** 33], 1).astype(np.int64)
    p = H.write_coords(str(tmp_path / 's.h5'), coords, ATTRS)
    code = ("import h5py, json, sys\n"
            "f = h5py.File(sys.argv[1], 'r'); d = f['coords']\n"
            "print(json.dumps(dict(n=len(d), patch_level=int(f['coords'].attrs['patch_level']), patch_size=int(f['coords'].attrs['patch_size']), coord7=d[7].tolist(),\n"
            "    all=d[:].tolist(), dtype=str(d.dtype), chunks=list(d.chunks), maxshape=[v for v in d.maxshape], name=d.attrs['name'], save_path=d.attrs['save_path'],\n"
            "    downsample=d.attrs['downsample'].tolist(), level_dim=d.attrs['level_dim'].tolist(), names=sorted(d.attrs.keys()))))\n")
    out = json.loads(subprocess.run(['/opt/conda/bin/python3.9', '-c', code, p], capture_output=True, text=True, check=True).stdout)
    assert out['n'] == 40 and out['all'] == coords.tolist() and out['coord7'] == coords[7].tolist() and out['dtype'] == 'int64'
    assert out['patch_level'] == 0 and out['patch_size'] == 256 and out['chunks'] == [1, 2] and out['maxshape'] == [None, 2]
    assert out['name'] == ATTRS['name'] and out['save_path'] == ATTRS['save_path'] and out['downsample'] == [1.0, 1.0] and out['level_dim'] == [40000, 30000]
    assert out['names'] == sorted(H.ATTR_ORDER)


def test_patch_folder_of_the_reference_is_consumed_as_it_is(tmp_path):
    """seg_and_patch leaves BOTH files; with only the .h5 (what a run of the reference's tool leaves in <save_dir>/patches) the slide is
    auto-skipped as already patched and the slide loop gets the same coordinates and patch size."""
    from nuhtc_amd import slides, tilestore
    from test_tissue import tissue_slide_with_holes
    src = tmp_path / 'slides'
    os.makedirs(src)
    img, *_ = tissue_slide_with_holes(H=768, W=1024)
    np.save(src / 'a.npy', img)
    out = tmp_path / 'out'
    dirs = dict(source=str(src), save_dir=str(out), patch_save_dir=str(out / 'patches'), mask_save_dir=str(out / 'masks'), stitch_save_dir=str(out / 'stitches'))
    for k, v in dirs.items():
        if k != 'source':
            os.makedirs(v)
    seg, flt, vis, pat = slides.default_parameters()
    run = lambda log, **kw: slides.seg_and_patch(**dirs, seg_params=seg, filter_params=flt, vis_params=vis, patch_params=pat, patch_size=64, step_size=64,
                                                  seg=True, patch=True, stitch=True, seg_downsample=8, log=lambda *a: log.append(' '.join(str(x) for x in a)), **kw)
    run([])
    npz, h5 = out / 'patches' / 'a.npz', out / 'patches' / 'a.h5'
    assert npz.exists() and h5.exists()
    z = np.load(npz)
    r = H.read_coords(str(h5))
    assert len(z['coords']) > 20 and np.array_equal(r['coords'], z['coords'])
    assert int(r['attrs']['patch_size']) == 64 and int(r['attrs']['patch_level']) == 0 and r['attrs']['name'] == 'a'
    assert np.array_equal(r['attrs']['level_dim'], [1024, 768]) and np.array_equal(r['attrs']['downsample'], [1.0, 1.0])       # (W, H) as level_dimensions
    assert r['attrs']['save_path'] == str(out / 'patches')
    os.remove(npz)                                              # the folder as the reference's tool leaves it
    assert slides.has_coords(str(out / 'patches'), 'a')
    c, ps, lvl = slides.load_coords(str(out / 'patches'), 'a')
    assert np.array_equal(c, z['coords']) and (ps, lvl) == (64, 0)
    log = []
    run(log)
    assert any('a already exist in destination location, skipped' in l for l in log) and not npz.exists()
    os.remove(out / 'stitches' / 'a.jpg')
    c2, ps2 = tilestore._load_coords(str(h5))                   # tools/infer_wsi.py --coords <the reference's file>
    assert np.array_equal(c2, c) and ps2 == 64
