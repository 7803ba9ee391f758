#!/usr/bin/env python
"""Same command line as the reference's tools/infer_patch.py:106-190: segment nuclei in the images listed in a CSV and
save one COCO document (images + RLE annotations with scores).

    python tools/infer_patch.py --csv labels.csv --config <config> --checkpoint <ckpt> --output nuclei_coco.json
                                [--image-col image_path --score-thr 0.35 --device cuda:0 --mag 40 --batch-size 16
                                 --mask-nms-thr 0.05 --vis-dir DIR --vis-samples 10]

Per image (reference :247-290): inference on the RGB array (the ndarray branch of inference_detector, i.e. the WSI channel
handling), mask-NMS at --mask-nms-thr in score order, one annotation per kept instance (category = class id, bbox/area from
the RLE, `score`).  Images of one batch must share a size (the engine is built per tile size)."""
import argparse
import csv
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    """tools/infer_patch.py:106-190 of the reference, flag for flag and default for default (tests/test_cli_parity.py)."""
    p = argparse.ArgumentParser(description='Segment nuclei from images and save to COCO format', allow_abbrev=False)
    p.add_argument('--csv', type=str, required=True, help='CSV file with an image path column')
    p.add_argument('--image-col', type=str, default='image_path')
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--checkpoint', type=str, required=True)
    p.add_argument('--output', type=str, default='nuclei_coco.json')
    p.add_argument('--score-thr', type=float, default=0.35)
    p.add_argument('--device', type=str, default='cuda:1')
    p.add_argument('--mag', type=int, default=40)
    p.add_argument('--batch-size', type=int, default=16)
    p.add_argument('--num-workers', type=int, default=8, help='accepted (images are read in-process)')
    p.add_argument('--vis-dir', type=str, default=None)
    p.add_argument('--vis-samples', type=int, default=10)
    p.add_argument('--mask-nms-thr', type=float, default=0.05)
    return p


def parse_args(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = build_parser().parse_args(argv)
    if not any(t == '--device' or t.startswith('--device=') for t in argv):
        # the reference's default is its authors' second GPU; an un-flagged run on a box without one takes the first instead of failing
        import torch
        if torch.cuda.device_count() < 2:
            print("--device not given and there is no cuda:1 on this machine: using cuda:0")
            a.device = 'cuda:0'
    return a


def main():
    a = parse_args()
    from PIL import Image, ImageDraw
    from nuhtc_amd import evaluation, outputs
    from nuhtc_amd.apis import concat_results, inference_detector, init_detector
    with open(a.csv, newline='') as f:
        rows = list(csv.DictReader(f))
    if rows and a.image_col not in rows[0]:
        raise ValueError(f"CSV must contain '{a.image_col}' column")
    paths = [r[a.image_col] for r in rows]
    if a.vis_dir and a.vis_samples > 0:                       # the reference then processes only a random sample (:199-203)
        rng = np.random.default_rng()
        paths = [paths[i] for i in rng.permutation(len(paths))[:min(a.vis_samples, len(paths))]]
        os.makedirs(a.vis_dir, exist_ok=True)
    from nuhtc_amd.config import Config, set_test_scale_factor
    cfg = Config.fromfile(a.config)
    set_test_scale_factor(cfg, a.mag)                  # reference infer_patch.py: scale_factor = 80 / mag
    model = init_detector(cfg, a.checkpoint, device=a.device, max_batch=a.batch_size)
    model.CLASSES = ('T', 'I', 'C', 'D', 'E')[:model.opts['num_classes']]
    doc = {'images': [], 'annotations': [], 'categories': [{'id': 0, 'name': 'nucleus', 'supercategory': 'nucleus'}]}
    nuclei_id = vis_count = 0
    for i0 in range(0, len(paths), a.batch_size):
        chunk = paths[i0:i0 + a.batch_size]
        imgs = [np.array(Image.open(p).convert('RGB')) for p in chunk]
        infos = [{'id': i0 + k + 1, 'file_name': os.path.basename(p), 'img_path': p, 'height': im.shape[0], 'width': im.shape[1]}
                 for k, (p, im) in enumerate(zip(chunk, imgs))]
        doc['images'].extend(infos)
        results = inference_detector(model, imgs)
        for info, res in zip(infos, results):
            boxes, labels, masks = concat_results(res)
            if len(masks) == 0:
                continue
            masks, idx = evaluation.mask_nms(masks, boxes[:, 4], thr=a.mask_nms_thr)
            anns = [outputs.coco_annotation(m, labels[j], info['id'], nuclei_id + k, score=boxes[j, 4])
                    for k, (m, j) in enumerate(zip(masks, idx))]
            nuclei_id += len(anns)
            doc['annotations'].extend(anns)
            if a.vis_dir and vis_count < a.vis_samples:
                im = Image.open(info['img_path']).convert('RGB')
                dr = ImageDraw.Draw(im)
                for an in anns:
                    x, y, w, h = an['bbox']
                    dr.rectangle([x, y, x + w, y + h], outline='green', width=1)
                    dr.text((x, y), f"{an['score']:.2f}", fill='black')
                im.save(os.path.join(a.vis_dir, f"{vis_count:04d}_{info['file_name']}"))
                vis_count += 1
    out_dir = os.path.dirname(a.output)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    with open(a.output, 'w') as f:
        json.dump(doc, f, indent=2)
    print(f"Total images processed: {len(doc['images'])}\nTotal nuclei: {nuclei_id}\nOutput saved to: {a.output}")


if __name__ == '__main__':
    main()
