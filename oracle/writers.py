"""TEST INFRASTRUCTURE ONLY (checker; never imported by the product).

The reference's output writers are inline script code in tools/infer_wsi.py:541-693 (QuPath GeoJSON, point GeoJSON, DSA, COCO,
SQLite) and tools/infer_patch.py:262-290 (COCO of single patches).  They cannot be run here (pycocotools, cv2, openslide, mmcv are
absent), so this module restates them statement by statement -- the same dict literals in the same key order, the same numpy
expressions -- as functions over one tile's post-NMS detections; tests/test_oracle_outputs.py feeds identical detections to these
and to the product's writers (nuhtc_amd.contours / nuhtc_amd.outputs) and compares the documents and SQLite rows field by field.
COCO RLE comes from oracle/rle.py (the product uses nuhtc_amd/cocomask.py)."""
import numpy as np

from . import rle as R

MAIN_CLASSES = ('T', 'I', 'C', 'D', 'E')                                                          # infer_wsi.py:423
INST_RNG_COLORS = [[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255]]             # :425
INST_FILLCOLOR = ["rgba(255, 0, 0, 0)", "rgba(0, 255, 0, 0)", "rgba(0, 0, 255, 0)", "rgba(255, 255, 0, 0)", "rgba(255, 0, 255, 0)"]   # :426
INST_LINECOLOR = ["rgb(255, 0, 0)", "rgb(0, 255, 0)", "rgb(0, 0, 255)", "rgb(255, 255, 0)", "rgb(255, 0, 255)"]                     # :427


def tile_documents(seg_contours, bbox, labels, fg_scores, annidx, h, w, rle_inst=None, classes=MAIN_CLASSES, nuclei_id=0):
    """One tile's entries for every output mode, from its detections after the filter + mask-NMS + contour selection (:533-546):
    seg_contours: list of (1, n, 2) int arrays in slide coordinates (`mask2inst(...) .reshape(1, -1, 2) + coord`), bbox (k, 4) in
    slide coordinates, labels (k,), fg_scores (k,); rle_inst: per detection {'size', 'counts' (str)} of the instance inside its tile.
    Returns dict(geojson, pointjson, dsajson, img_dict, annts, sql_rows)."""
    k = len(seg_contours)
    geojson = [{                                                                                   # :549-566
        "type": "Feature",
        "geometry": {"type": "Polygon", "coordinates": seg_contours[i].tolist()},
        "properties": {"objectType": "annotation", "label": int(labels[i]), "score": float(fg_scores[i]),
                       "classification": {"name": classes[labels[i]], "color": INST_RNG_COLORS[labels[i]]}, "isLocked": False}
    } for i in range(k)]
    pointjson = [{                                                                                 # :567-583
        "type": "Feature",
        "geometry": {"type": "Point", "coordinates": [(bbox[i][0] + bbox[i][2]) / 2, (bbox[i][1] + bbox[i][3]) / 2]},
        "properties": {"objectType": "annotation", "label": int(labels[i]), "score": float(fg_scores[i]),
                       "classification": {"name": classes[labels[i]], "color": INST_RNG_COLORS[labels[i]]}, "isLocked": False}
    } for i in range(k)]
    dsajson = [{                                                                                   # :588-598
        "fillColor": INST_FILLCOLOR[labels[i]], "lineColor": INST_LINECOLOR[labels[i]], "lineWidth": 2, "type": "polyline", "closed": True,
        "points": np.pad(seg_contours[i][0], ((0, 0), (0, 1),), 'constant', constant_values=0).tolist(),
        "label": {"value": classes[labels[i]]}
    } for i in range(k)]
    # scipy.stats.mode(labels)[0]: the smallest of the most frequent values (:609)
    vals, cnt = np.unique(np.asarray(labels), return_counts=True)
    mode = int(vals[np.argmax(cnt)]) if k else 0
    img_dict = {'file_name': f'{annidx}.png', 'height': h, 'width': w, 'id': annidx, 'n_objects': k, 'type': classes[mode]}   # :602-610
    annts = []
    if rle_inst is not None:
        for i in range(k):                                                                         # :613-627
            r = rle_inst[i]
            bb = R.rle_to_bbox(r['size'][0], r['size'][1], R.rle_fr_string(r['counts']))
            annts.append({'bbox': bb, 'area': bb[2] * bb[3], 'image_id': annidx, 'category_id': int(labels[i]), 'id': nuclei_id + i,
                          'iscrowd': 0, 'segmentation': r})
    sql_rows = []
    for i in range(k):                                                                             # :637-655 (values of the INSERT, in its column order)
        c = seg_contours[i]
        xs, ys = c[0, :, 0], c[0, :, 1]
        sql_rows.append((annidx, i, 'polyline', classes[labels[i]], float(fg_scores[i]), INST_LINECOLOR[labels[i]], int(xs.min()), int(ys.min()),
                         int(xs.max()), int(ys.max()), int((xs.max() - xs.min()) * (ys.max() - ys.min())),
                         ','.join(xs.astype(str)), ','.join(ys.astype(str)), 1))
    return dict(geojson=geojson, pointjson=pointjson, dsajson=dsajson, img_dict=img_dict, annts=annts, sql_rows=sql_rows)


def dsa_file(dsajson_li):                                                                          # :667-674
    return {'description': 'Seg with NuHTC automatically', 'elements': dsajson_li, 'name': 'NuHTC'}


def coco_file(img_li, annt_li, classes=MAIN_CLASSES):                                               # :429-432,675-678
    return {'images': img_li, 'annotations': annt_li, 'categories': [{'id': i, 'name': classes[i]} for i in range(len(classes))]}


def infer_patch_annotation(mask, label, score, image_id, nuclei_id):
    """tools/infer_patch.py:271-288: one annotation of a patch (RLE of the whole patch mask, bbox from the RLE, bbox area)."""
    h, w, cnts = R.rle_encode(mask)
    bb = R.rle_to_bbox(h, w, cnts)
    return {'id': nuclei_id, 'bbox': bb, 'area': bb[2] * bb[3], 'image_id': image_id, 'category_id': int(label), 'iscrowd': 0,
            'segmentation': {'size': [h, w], 'counts': R.rle_to_string(cnts)}, 'score': float(score)}
