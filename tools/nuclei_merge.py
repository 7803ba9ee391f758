#!/usr/bin/env python
"""Same command line as the reference's tools/nuclei_merge.py:221-230: cross-tile duplicate removal on a GeoJSON list.

    python tools/nuclei_merge.py --geojson slide.geojson [--overlap_threshold 0.01] [--merge_strategy probability|area]
                                 [--output_name NAME] [--uniform_classification]
"""
import json
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuhtc_amd.contours import merge_features  # noqa: E402


def build_parser():
    """tools/nuclei_merge.py:221-230 of the reference (tests/test_cli_parity.py)."""
    p = ArgumentParser(allow_abbrev=False)
    p.add_argument('--geojson', help='geojson file name')
    p.add_argument('--output_name', default=None, type=str, help='output geojson file name')
    p.add_argument('--overlap_threshold', type=float, default=0.01, help='area overlap percentage threshold to be removed')
    p.add_argument('--merge_strategy', default='probability', help="'probability' or 'area'")
    p.add_argument('--uniform_classification', action='store_true')
    # ---- not in the reference
    p.add_argument('--host', action='store_true', help='merge with the host polygon code even when a GPU is there')
    p.add_argument('--device', type=int, default=0, help='GPU of the merge')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def main():
    args = parse_args()
    datadir = os.path.dirname(args.geojson) or '.'
    name = os.path.basename(args.geojson).split('.geojson')[0]
    with open(os.path.join(datadir, name + '.geojson')) as f:
        data = json.load(f)
    feats = None
    if args.merge_strategy == 'probability' and not args.host:
        # a file of traced rings (what tools/infer_wsi.py writes) is merged on the GPU: the rings are filled back into mask crops and the
        # polygon IoU of the reference is measured on them exactly (nuhtc_amd.contours.merge_features_device); anything else -- user-drawn
        # polygons, strategy 'area', no GPU -- takes the host's polygon code, which gives the same kept features
        try:
            import torch
            if torch.cuda.is_available():
                from nuhtc_amd.contours import merge_features_device
                feats = merge_features_device(data, args.overlap_threshold, device=args.device)
        except ImportError:
            feats = None
    path = 'GPU (rings filled into mask crops, nuhtc_merge_overlap)' if feats is not None else 'host polygons'
    if feats is None:
        feats = merge_features(data, args.overlap_threshold, args.merge_strategy)
    if args.uniform_classification:
        for ft in feats:
            ft['properties']['classification'] = {'name': 'uniform', 'color': [255, 255, 0]}
    out = os.path.join(datadir, (args.output_name or name + '_merged') + '.geojson')
    with open(out, 'w') as f:
        json.dump(feats, f)
    print(f'{len(data)} features -> {len(feats)} after merge ({path}); wrote {out}')


if __name__ == '__main__':
    main()
