#!/usr/bin/env python
"""Same command line as the reference's tools/infer.py:17-38: one image per inference_detector call.

    python tools/infer.py <img_dir> <config> <checkpoint> [--device cuda:0] [--score-thr 0.35] [--output demo/imgs_infer]

With --output every image gets an overlay file of that name in the directory (`save_result`, :57-66).  Without it the reference
opens a matplotlib window per image (`show_result_pyplot`, :55); there is no display here, so the instance counts are printed.
"""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    """tools/infer.py:17-38 of the reference, flag for flag and default for default (tests/test_cli_parity.py)."""
    p = argparse.ArgumentParser(allow_abbrev=False)
    p.add_argument('img', help='Image file')
    p.add_argument('config', help='Config file')
    p.add_argument('checkpoint', help='Checkpoint file')
    p.add_argument('--device', default='cuda:0', help='Device used for inference')
    p.add_argument('--score-thr', type=float, default=0.35, help='bbox score threshold')
    p.add_argument('--async-test', action='store_true', help='whether to set async options for async inference.')
    p.add_argument('--output', type=str, default=None, help='specify the directory to save visualization results.')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def main():
    args = parse_args()
    from nuhtc_amd.apis import inference_detector, init_detector, save_result
    if args.async_test:
        print('--async-test: accepted; inference is synchronous per image (the reference awaits each image in turn too, :68-71)')
    model = init_detector(args.config, args.checkpoint, device=args.device, max_batch=1)
    model.CLASSES = ('T', 'I', 'C', 'D', 'E')
    # the reference globs f"{args.img}/*png" (:50); a single file is accepted as well
    imgs = sorted(glob.glob(f'{args.img}/*png')) if os.path.isdir(args.img) else [args.img]
    if args.output is not None:
        os.makedirs(args.output, exist_ok=True)
    for img in imgs:
        result = inference_detector(model, img)
        n = sum(len(b) for b in result[0])
        if args.output is None:
            print(img, n, 'instances,', int(sum((b[:, 4] >= args.score_thr).sum() for b in result[0])), f'with score >= {args.score_thr} (no display: pass --output for overlays)')
        else:
            out = os.path.join(args.output, os.path.basename(img))
            print(f'Save results to {out}')
            save_result(model, img, result, score_thr=args.score_thr, out_file=out)
            print(img, '->', out, n, 'instances')


if __name__ == '__main__':
    main()
