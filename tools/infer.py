#!/usr/bin/env python
"""Same command line as the reference's tools/infer.py:17-38: one image per inference_detector call, result overlay PNGs.

    python tools/infer.py <img_dir> <config> <checkpoint> [--device cuda:0] [--score-thr 0.3] [--output demo/imgs_infer]
"""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuhtc_amd.apis import inference_detector, init_detector, save_result  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('img', help='Image file or directory of *.png')
    p.add_argument('config', help='Config file')
    p.add_argument('checkpoint', help='Checkpoint file')
    p.add_argument('--device', default='cuda:0', help='Device used for inference')
    p.add_argument('--score-thr', type=float, default=0.3, help='bbox score threshold')
    p.add_argument('--async-test', action='store_true', help='accepted for compatibility (ignored)')
    p.add_argument('--output', type=str, default='demo/imgs_infer', help='specify the directory to save visualization results.')
    return p.parse_args()


def main():
    args = parse_args()
    model = init_detector(args.config, args.checkpoint, device=args.device, max_batch=1)
    model.CLASSES = ('T', 'I', 'C', 'D', 'E')
    imgs = sorted(glob.glob(os.path.join(args.img, '*.png'))) if os.path.isdir(args.img) else [args.img]
    os.makedirs(args.output, exist_ok=True)
    for img in imgs:
        result = inference_detector(model, img)
        out = os.path.join(args.output, os.path.basename(img))
        save_result(model, img, result, score_thr=args.score_thr, out_file=out)
        print(img, '->', out, sum(len(b) for b in result[0]), 'instances')


if __name__ == '__main__':
    main()
