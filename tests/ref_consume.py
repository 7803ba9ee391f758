"""TEST INFRASTRUCTURE: the reference's per-tile consumption loop of `inference_detector` results, restated from the oracle's pieces,
so that a GPU test can rebuild every output document of the CLIs from the engine's own results and compare object for object.

    tools/infer_wsi.py:486-546   concat class lists, margin / min-area filter, mask_nms(thr 0.05), mask2inst (first cv2 contour, closed),
                                 `len(con) >= 3`, + tile origin                       -> oracle.model.tile_filter_and_mask_nms,
                                                                                          oracle.contour.find_contours_tree
    tools/infer_wsi.py:549-655   QuPath / point / DSA / COCO / SQL entries of the tile  -> oracle.writers.tile_documents, oracle.rle
    tools/infer_patch.py:247-290 mask_nms only, one RLE annotation per kept instance   -> oracle.writers.infer_patch_annotation

Nothing here is imported by the product."""
import numpy as np

from oracle import contour as OC
from oracle import model as O
from oracle import rle as R
from oracle import writers as W


def mask2inst(mask):
    """tools/infer_wsi.py:51-54: `cv2.findContours(m, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]` with its first point appended."""
    c, _ = OC.find_contours_tree(np.asarray(mask, bool))
    c = np.asarray(c[0], np.int64).reshape(-1, 2)
    return np.concatenate([c, c[[0]]], 0)


def wsi_documents(results, coords, P, classes=W.MAIN_CLASSES, margin=2, min_area=10, thr=0.05, first_annidx=0):
    """results: list of (bbox_results, segm_results) per tile (inference_detector), coords (n, 2) tile origins.  Returns the documents the
    reference would have written for the slide (dict of lists: geojson, pointjson, dsajson, imgs, annts, sql_rows) and, per kept
    detection, (tile, closed ring, label, score, mask) for further checks."""
    out = dict(geojson=[], pointjson=[], dsajson=[], imgs=[], annts=[], sql_rows=[])
    kept = []
    nid = 0
    for t, (bbox_res, segm_res) in enumerate(results):
        annidx = first_annidx + t
        boxes, labels, masks = O.tile_filter_and_mask_nms(bbox_res, segm_res, size=P, margin=margin, min_area=min_area, thr=thr)
        if len(masks) == 0:
            continue
        cons = [mask2inst(m) for m in masks]
        sel = np.array([len(c) >= 3 for c in cons])                                   # :536
        coord = np.asarray(coords[t], np.int64)
        seg_contours = [cons[i].reshape(1, -1, 2) + coord for i in range(len(cons)) if sel[i]]
        if not seg_contours:
            continue
        bb = boxes[sel, :4] + np.tile(coord, 2)
        lab, sc, mk = labels[sel], boxes[sel, 4], masks[sel]
        d = W.tile_documents(seg_contours, bb, lab, sc, annidx, P, P, rle_inst=[R.encode(m.tolist()) for m in mk], classes=classes, nuclei_id=nid)
        nid += len(seg_contours)
        out['geojson'] += d['geojson']; out['pointjson'] += d['pointjson']; out['dsajson'] += d['dsajson']
        out['imgs'].append(d['img_dict']); out['annts'] += d['annts']; out['sql_rows'] += d['sql_rows']
        kept += [(annidx, seg_contours[i][0], int(lab[i]), float(sc[i]), mk[i]) for i in range(len(seg_contours))]
    return out, kept


def patch_annotations(results, first_image_id=1, thr=0.05):
    """tools/infer_patch.py:247-290: per image mask_nms in score order, one annotation per kept instance, ids running over the file."""
    annts, nid = [], 0
    for k, (bbox_res, segm_res) in enumerate(results):
        if not sum(len(c) for c in segm_res):
            continue
        size = max(segm_res[c][0].shape[0] for c in range(len(segm_res)) if len(segm_res[c]))
        boxes, labels, masks = O.tile_filter_and_mask_nms(bbox_res, segm_res, size=size, margin=-(1 << 30), min_area=0, thr=thr)
        for i in range(len(masks)):
            annts.append(W.infer_patch_annotation(masks[i].tolist(), labels[i], boxes[i, 4], first_image_id + k, nid + i))
        nid += len(masks)
    return annts
