"""The reference's tile-coordinate files, patches/<slide_id>.h5 (HDF5).

Written by tools/wsi_core/WholeSlideImage.py:388-406,481-492 through wsi_utils.save_hdf5 (:66-85): ONE dataset `coords`, int64 (n, 2) level-0
origins, chunks (1, 2), maxshape (None, 2) -- the reference appends contour by contour --, with the attributes `patch_size`, `patch_level`
(int64 scalars), `downsample` (float64 (2,)), `downsampled_level_dim`, `level_dim` (int64 (2,)), `name`, `save_path` (variable-length UTF-8
strings).  Read by Whole_Slide_Bag_FP (:862-865: `coords`, attrs `patch_level`, `patch_size`).

Two back ends, the same file either way: `h5py` where it is importable (the reference's own dependency), else the HDF5 C library through
ctypes (libhdf5 >= 1.10: NUHTC_HDF5_LIB, the loader's search path, then the usual install prefixes).  Neither present: `available()` is False
and read / write raise -- the `.npz` twin (nuhtc_amd.slides.save_coords) is then the only coordinate file."""
import ctypes
import ctypes.util
import glob
import os
import sys

import numpy as np

ATTR_ORDER = ('patch_size', 'patch_level', 'downsample', 'downsampled_level_dim', 'level_dim', 'name', 'save_path')   # WholeSlideImage.py:483-489

_hid = ctypes.c_int64          # hid_t of HDF5 >= 1.10
_hsz = ctypes.c_uint64         # hsize_t
_LIB = None
_TRIED = False


def _candidates():
    env = os.environ.get('NUHTC_HDF5_LIB')
    if env:
        yield env
    found = ctypes.util.find_library('hdf5') or ctypes.util.find_library('hdf5_serial')
    if found:
        yield found
    for pat in ('/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*', '/usr/lib/x86_64-linux-gnu/libhdf5.so*', '/usr/lib64/libhdf5.so*', '/usr/local/lib/libhdf5.so*',
                os.path.join(sys.prefix, 'lib', 'libhdf5.so*'), '/opt/conda/lib/libhdf5.so*'):
        for p in sorted(glob.glob(pat), key=len):
            yield p


def _lib():
    """The HDF5 C library with the prototypes this module uses, or None."""
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    for cand in _candidates():
        try:
            lib = ctypes.CDLL(cand)
            lib.H5open.restype = ctypes.c_int
            if lib.H5open() < 0:
                continue
            maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
            lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
            if (maj.value, mnr.value) < (1, 10):          # hid_t was a 32-bit int before 1.10
                continue
        except (OSError, AttributeError):
            continue
        H, I, S, P, CP = _hid, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p
        HP = ctypes.POINTER(_hsz)
        proto = {
            'H5Eset_auto2': (I, [H, P, P]), 'H5Fopen': (H, [CP, ctypes.c_uint, H]), 'H5Fcreate': (H, [CP, ctypes.c_uint, H, H]), 'H5Fclose': (I, [H]),
            'H5Dopen2': (H, [H, CP, H]), 'H5Dcreate2': (H, [H, CP, H, H, H, H, H]), 'H5Dget_space': (H, [H]), 'H5Dget_type': (H, [H]),
            'H5Dget_create_plist': (H, [H]), 'H5Dread': (I, [H, H, H, H, H, P]), 'H5Dwrite': (I, [H, H, H, H, H, P]), 'H5Dclose': (I, [H]),
            'H5Screate': (H, [I]), 'H5Screate_simple': (H, [I, HP, HP]), 'H5Sget_simple_extent_ndims': (I, [H]),
            'H5Sget_simple_extent_dims': (I, [H, HP, HP]), 'H5Sget_simple_extent_npoints': (ctypes.c_int64, [H]), 'H5Sclose': (I, [H]),
            'H5Pcreate': (H, [H]), 'H5Pset_chunk': (I, [H, I, HP]), 'H5Pget_chunk': (I, [H, I, HP]), 'H5Pget_layout': (I, [H]), 'H5Pset_fill_time': (I, [H, I]), 'H5Pclose': (I, [H]),
            'H5Acreate2': (H, [H, CP, H, H, H, H]), 'H5Awrite': (I, [H, H, P]), 'H5Aread': (I, [H, H, P]), 'H5Aopen': (H, [H, CP, H]),
            'H5Aopen_by_idx': (H, [H, CP, I, I, _hsz, H, H]), 'H5Aget_name': (ctypes.c_ssize_t, [H, S, CP]), 'H5Aget_type': (H, [H]),
            'H5Aget_space': (H, [H]), 'H5Aclose': (I, [H]), 'H5Aexists': (I, [H, CP]),
            'H5Tcopy': (H, [H]), 'H5Tset_size': (I, [H, S]), 'H5Tset_cset': (I, [H, I]), 'H5Tget_class': (I, [H]), 'H5Tget_size': (S, [H]),
            'H5Tis_variable_str': (I, [H]), 'H5Tclose': (I, [H]), 'H5free_memory': (I, [P]),
        }
        try:
            for name, (res, args) in proto.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            lib._ids = {n: _hid.in_dll(lib, n + '_g').value for n in ('H5T_NATIVE_INT64', 'H5T_NATIVE_DOUBLE', 'H5T_STD_I64LE', 'H5T_IEEE_F64LE', 'H5T_C_S1')}
            lib._ids['H5P_DATASET_CREATE'] = _hid.in_dll(lib, 'H5P_CLS_DATASET_CREATE_ID_g').value
        except (AttributeError, ValueError):
            continue
        lib.H5Eset_auto2(0, None, None)      # no error stack on stderr: failures are raised below
        lib._path = cand
        _LIB = lib
        break
    return _LIB


def _h5py():
    try:
        import h5py
        return h5py
    except Exception:
        return None


def available():
    return _h5py() is not None or _lib() is not None


def backend():
    return 'h5py' if _h5py() is not None else ('libhdf5:' + _lib()._path if _lib() is not None else None)


class H5Error(RuntimeError):
    pass


def _ck(v, what):
    if v < 0:
        raise H5Error(f'HDF5: {what} failed')
    return v


_UNLIMITED = (1 << 64) - 1      # H5S_UNLIMITED
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3
_VARIABLE = ctypes.c_size_t(-1).value      # H5T_VARIABLE


def _dims(lib, space):
    nd = _ck(lib.H5Sget_simple_extent_ndims(space), 'H5Sget_simple_extent_ndims')
    d, m = (_hsz * max(nd, 1))(), (_hsz * max(nd, 1))()
    if nd:
        _ck(lib.H5Sget_simple_extent_dims(space, d, m), 'H5Sget_simple_extent_dims')
    return [int(v) for v in d[:nd]], [None if v == _UNLIMITED else int(v) for v in m[:nd]]


def _read_attr(lib, attr):
    typ, space = _ck(lib.H5Aget_type(attr), 'H5Aget_type'), _ck(lib.H5Aget_space(attr), 'H5Aget_space')
    try:
        shape, _ = _dims(lib, space)
        n = int(np.prod(shape)) if shape else 1
        cls = lib.H5Tget_class(typ)
        if cls == _H5T_INTEGER:
            buf = np.empty(n, np.int64)
            _ck(lib.H5Aread(attr, lib._ids['H5T_NATIVE_INT64'], buf.ctypes.data_as(ctypes.c_void_p)), 'H5Aread')
            out = buf.reshape(shape) if shape else buf[0]
        elif cls == _H5T_FLOAT:
            buf = np.empty(n, np.float64)
            _ck(lib.H5Aread(attr, lib._ids['H5T_NATIVE_DOUBLE'], buf.ctypes.data_as(ctypes.c_void_p)), 'H5Aread')
            out = buf.reshape(shape) if shape else buf[0]
        elif cls == _H5T_STRING:
            mem = _ck(lib.H5Tcopy(lib._ids['H5T_C_S1']), 'H5Tcopy')
            try:
                if lib.H5Tis_variable_str(typ) > 0:
                    _ck(lib.H5Tset_size(mem, _VARIABLE), 'H5Tset_size')
                    lib.H5Tset_cset(mem, 1)      # H5T_CSET_UTF8 (ASCII files convert)
                    ptrs = (ctypes.c_void_p * n)()
                    if lib.H5Aread(attr, mem, ptrs) < 0:      # an ASCII-tagged string: read it as such
                        lib.H5Tset_cset(mem, 0)
                        _ck(lib.H5Aread(attr, mem, ptrs), 'H5Aread')
                    vals = []
                    for p in ptrs:
                        vals.append(ctypes.string_at(p).decode('utf-8', 'replace') if p else '')
                        if p:
                            lib.H5free_memory(p)
                else:
                    size = int(lib.H5Tget_size(typ))
                    _ck(lib.H5Tset_size(mem, size), 'H5Tset_size')
                    raw = ctypes.create_string_buffer(size * n)
                    _ck(lib.H5Aread(attr, typ, raw), 'H5Aread')
                    vals = [raw.raw[i * size:(i + 1) * size].split(b'\0')[0].decode('utf-8', 'replace') for i in range(n)]
            finally:
                lib.H5Tclose(mem)
            out = np.array(vals, dtype=object).reshape(shape) if shape else vals[0]
        else:
            out = None      # nothing the reference writes
        return out
    finally:
        lib.H5Sclose(space)
        lib.H5Tclose(typ)


def read_coords(path):
    """-> dict(coords=int64 (n, 2), attrs={...}, chunks=tuple | None, maxshape=tuple): the `coords` dataset of a reference .h5 and its attributes."""
    h5 = _h5py()
    if h5 is not None:
        with h5.File(path, 'r') as f:
            d = f['coords']
            attrs = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in d.attrs.items()}
            return dict(coords=np.asarray(d[:], np.int64).reshape(-1, 2), attrs=attrs, chunks=d.chunks, maxshape=d.maxshape)
    lib = _lib()
    if lib is None:
        raise H5Error('no HDF5 back end (neither h5py nor libhdf5 >= 1.10; set NUHTC_HDF5_LIB): use the .npz coordinate file')
    f = lib.H5Fopen(os.fsencode(path), 0, 0)      # H5F_ACC_RDONLY
    if f < 0:
        raise H5Error(f'cannot open {path} as HDF5')
    try:
        d = lib.H5Dopen2(f, b'coords', 0)
        if d < 0:
            raise H5Error(f'{path}: no dataset `coords`')
        try:
            space = _ck(lib.H5Dget_space(d), 'H5Dget_space')
            shape, maxshape = _dims(lib, space)
            lib.H5Sclose(space)
            typ = _ck(lib.H5Dget_type(d), 'H5Dget_type')
            cls = lib.H5Tget_class(typ)
            lib.H5Tclose(typ)
            if cls != _H5T_INTEGER or len(shape) != 2 or shape[1] != 2:
                raise H5Error(f'{path}: `coords` is not an integer (n, 2) dataset (shape {shape})')
            coords = np.empty(shape, np.int64)
            if coords.size:
                _ck(lib.H5Dread(d, lib._ids['H5T_NATIVE_INT64'], 0, 0, 0, coords.ctypes.data_as(ctypes.c_void_p)), 'H5Dread')
            plist = _ck(lib.H5Dget_create_plist(d), 'H5Dget_create_plist')
            chunks = None
            if lib.H5Pget_layout(plist) == 2:      # H5D_CHUNKED
                c = (_hsz * 2)()
                _ck(lib.H5Pget_chunk(plist, 2, c), 'H5Pget_chunk')
                chunks = (int(c[0]), int(c[1]))
            lib.H5Pclose(plist)
            attrs, i = {}, 0
            while True:
                a = lib.H5Aopen_by_idx(d, b'.', 0, 0, i, 0, 0)      # H5_INDEX_NAME, H5_ITER_INC
                if a < 0:
                    break
                try:
                    n = lib.H5Aget_name(a, 0, None)
                    buf = ctypes.create_string_buffer(int(n) + 1)
                    lib.H5Aget_name(a, int(n) + 1, buf)
                    attrs[buf.value.decode()] = _read_attr(lib, a)
                finally:
                    lib.H5Aclose(a)
                i += 1
            return dict(coords=coords, attrs=attrs, chunks=chunks, maxshape=tuple(maxshape))
        finally:
            lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)


def _write_attr(lib, d, name, val):
    if isinstance(val, str):
        typ = _ck(lib.H5Tcopy(lib._ids['H5T_C_S1']), 'H5Tcopy')
        _ck(lib.H5Tset_size(typ, _VARIABLE), 'H5Tset_size')
        lib.H5Tset_cset(typ, 1)
        space = _ck(lib.H5Screate(0), 'H5Screate')      # H5S_SCALAR
        raw = ctypes.c_char_p(val.encode('utf-8'))
        a = _ck(lib.H5Acreate2(d, name.encode(), typ, space, 0, 0), 'H5Acreate2')
        rc = lib.H5Awrite(a, typ, ctypes.byref(raw))
        lib.H5Aclose(a); lib.H5Sclose(space); lib.H5Tclose(typ)
        _ck(rc, 'H5Awrite')
        return
    arr = np.asarray(val)
    is_f = arr.dtype.kind == 'f'
    arr = np.array(arr, dtype=np.float64 if is_f else np.int64, order='C')      # (ascontiguousarray would make a 0-d value 1-d)
    if arr.ndim == 0:
        space = _ck(lib.H5Screate(0), 'H5Screate')
    else:
        dims = (_hsz * arr.ndim)(*arr.shape)
        space = _ck(lib.H5Screate_simple(arr.ndim, dims, None), 'H5Screate_simple')
    ftype, mtype = (lib._ids['H5T_IEEE_F64LE'], lib._ids['H5T_NATIVE_DOUBLE']) if is_f else (lib._ids['H5T_STD_I64LE'], lib._ids['H5T_NATIVE_INT64'])
    a = _ck(lib.H5Acreate2(d, name.encode(), ftype, space, 0, 0), 'H5Acreate2')
    rc = lib.H5Awrite(a, mtype, arr.ctypes.data_as(ctypes.c_void_p))
    lib.H5Aclose(a); lib.H5Sclose(space)
    _ck(rc, 'H5Awrite')


def write_coords(path, coords, attrs):
    """A coordinate file with the layout save_hdf5 gives it (chunks (1, 2), maxshape (None, 2), attributes on the dataset)."""
    coords = np.ascontiguousarray(np.asarray(coords, np.int64).reshape(-1, 2))
    h5 = _h5py()
    if h5 is not None:
        with h5.File(path, 'w') as f:
            d = f.create_dataset('coords', shape=coords.shape, maxshape=(None, 2), chunks=(1, 2), dtype=coords.dtype)
            d[:] = coords
            for k, v in attrs.items():
                d.attrs[k] = v
        return path
    lib = _lib()
    if lib is None:
        raise H5Error('no HDF5 back end (neither h5py nor libhdf5 >= 1.10; set NUHTC_HDF5_LIB)')
    f = lib.H5Fcreate(os.fsencode(path), 2, 0, 0)      # H5F_ACC_TRUNC
    if f < 0:
        raise H5Error(f'cannot create {path}')
    try:
        dims, maxd, chunk = (_hsz * 2)(len(coords), 2), (_hsz * 2)(_UNLIMITED, 2), (_hsz * 2)(1, 2)
        space = _ck(lib.H5Screate_simple(2, dims, maxd), 'H5Screate_simple')
        plist = _ck(lib.H5Pcreate(lib._ids['H5P_DATASET_CREATE']), 'H5Pcreate')
        _ck(lib.H5Pset_chunk(plist, 2, chunk), 'H5Pset_chunk')
        lib.H5Pset_fill_time(plist, 0)      # H5D_FILL_TIME_ALLOC, h5py's setting: `h5dump -p` of the file equals that of one h5py wrote
        d = lib.H5Dcreate2(f, b'coords', lib._ids['H5T_STD_I64LE'], space, 0, plist, 0)
        lib.H5Pclose(plist); lib.H5Sclose(space)
        _ck(d, 'H5Dcreate2')
        try:
            if len(coords):
                _ck(lib.H5Dwrite(d, lib._ids['H5T_NATIVE_INT64'], 0, 0, 0, coords.ctypes.data_as(ctypes.c_void_p)), 'H5Dwrite')
            for k, v in attrs.items():
                _write_attr(lib, d, k, v)
        finally:
            lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)
    return path
