"""Golden coordinate file written by the REFERENCE's own code with the real h5py (build container only; test infrastructure).

    /opt/conda/bin/python3.9 oracle/ref_harness/make_h5_golden.py      # the image's Python 3.9 has h5py 3.3.0 (HDF5 1.10.6); the 3.10 interpreter has none
        -> tests/golden/coords_reference.h5, tests/golden/coords_reference.json

The reference's `save_hdf5` (tools/wsi_core/wsi_utils.py:66-85) is imported from where it lies and called the way
`WholeSlideImage.process_contours` calls it (tools/wsi_core/WholeSlideImage.py:388-406): mode 'w' with the first contour's coordinates and the
attribute dict `process_contour` builds (:481-492), then mode 'a' per further contour (resize + write at the end).  `cv2` -- imported at the top
of wsi_utils.py and of util_classes.py, used by neither `save_hdf5` nor anything executed here -- is absent from that interpreter and is
replaced by an empty module for the import.  The .json holds what h5py itself reports for the file (values, dtypes, chunks, maxshape,
attributes): tests/test_h5coords.py reads the .h5 through nuhtc_amd.h5coords and compares."""
import json
import os
import sys
import types

import numpy as np

REF_TOOLS = '/root/reference/tools'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')


def main():
    import h5py
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    sys.path.insert(0, REF_TOOLS)
    from wsi_core.wsi_utils import save_hdf5

    # three "contours" worth of tile origins as process_contour returns them (np.array of the kept coordinate candidates: int64 (n, 2), x then y)
    level_dim = (98304, 73728)
    rng = np.random.default_rng(5)
    parts = []
    for x0, y0, nx, ny in ((2048, 4096, 9, 7), (50176, 20480, 4, 11), (90112, 69632, 3, 2)):
        xs, ys = np.meshgrid(np.arange(x0, x0 + nx * 192, 192), np.arange(y0, y0 + ny * 192, 192), indexing='ij')
        cand = np.array([xs.flatten(), ys.flatten()]).transpose()
        parts.append(cand[rng.random(len(cand)) > 0.25])
    attr = {'patch_size': 256, 'patch_level': 0, 'downsample': (1.0, 1.0), 'downsampled_level_dim': tuple(np.array(level_dim)), 'level_dim': level_dim,
            'name': 'TCGA-A1-0001', 'save_path': '/data/wsi_infer/patches'}
    path = os.path.join(OUT, 'coords_reference.h5')
    if os.path.exists(path):
        os.remove(path)
    save_hdf5(path, {'coords': parts[0]}, {'coords': attr}, mode='w')
    for p in parts[1:]:
        save_hdf5(path, {'coords': p}, mode='a')

    with h5py.File(path, 'r') as f:            # what h5py reports (and what Whole_Slide_Bag_FP would take: WholeSlideImage.py:862-865)
        d = f['coords']
        rep = dict(h5py=h5py.__version__, hdf5=h5py.version.hdf5_version, keys=list(f.keys()), dtype=str(d.dtype), shape=list(d.shape), chunks=list(d.chunks),
                   maxshape=[None if v is None else int(v) for v in d.maxshape], coords=d[:].tolist(),
                   attrs={k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in d.attrs.items()},
                   attr_dtypes={k: (str(v.dtype) if hasattr(v, 'dtype') else type(v).__name__) for k, v in d.attrs.items()},
                   patch_level=int(d.attrs['patch_level']), patch_size=int(d.attrs['patch_size']))
    assert rep['coords'] == np.concatenate(parts, 0).tolist()
    json.dump(rep, open(os.path.join(OUT, 'coords_reference.json'), 'w'), indent=1)
    print(path, os.path.getsize(path), 'bytes;', rep['shape'], rep['dtype'], rep['chunks'], rep['maxshape'], rep['attr_dtypes'])


if __name__ == '__main__':
    main()
