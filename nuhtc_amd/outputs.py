"""Writers for the reference's non-QuPath output modes of tools/infer_wsi.py (`--mode dsa|coco|sql`, :587-693) and the
COCO document of tools/infer_patch.py:216-224,271-290.  They consume closed contour rings (nuhtc_amd.contours.mask_to_ring,
slide coordinates) and per-tile boolean masks; QuPath GeoJSON lives in nuhtc_amd.contours."""
import json
import sqlite3

import numpy as np

from . import cocomask

FILL_COLORS = ["rgba(255, 0, 0, 0)", "rgba(0, 255, 0, 0)", "rgba(0, 0, 255, 0)", "rgba(255, 255, 0, 0)", "rgba(255, 0, 255, 0)"]
LINE_COLORS = ["rgb(255, 0, 0)", "rgb(0, 255, 0)", "rgb(0, 0, 255)", "rgb(255, 255, 0)", "rgb(255, 0, 255)"]   # infer_wsi.py:421-423


# ----------------------------------------------------------------------------- Digital Slide Archive (HistomicsUI)
def dsa_element(ring, label, classes):
    """One polyline element (infer_wsi.py:588-598): the ring's points padded with z = 0."""
    ring = np.asarray(ring)
    pts = np.pad(ring, ((0, 0), (0, 1)), 'constant', constant_values=0).tolist()
    return {"fillColor": FILL_COLORS[label % len(FILL_COLORS)], "lineColor": LINE_COLORS[label % len(LINE_COLORS)], "lineWidth": 2,
            "type": "polyline", "closed": True, "points": pts, "label": {"value": classes[label]}}


def dsa_document(elements):
    return {'description': 'Seg with NuHTC automatically', 'elements': elements, 'name': 'NuHTC'}   # infer_wsi.py:668-674


# ----------------------------------------------------------------------------- COCO
def coco_categories(classes):
    return [{'id': i, 'name': n} for i, n in enumerate(classes)]                                  # infer_wsi.py:425-428


def coco_tile_image(annidx, h, w, labels, classes):
    """Image record of one tile in `--mode coco` (infer_wsi.py:602-610): `type` is the most frequent class of the tile."""
    labels = np.asarray(labels, dtype=int)
    mode = int(np.bincount(labels).argmax()) if len(labels) else 0
    return {'file_name': f'{annidx}.png', 'height': int(h), 'width': int(w), 'id': int(annidx), 'n_objects': int(len(labels)),
            'type': classes[mode]}


def coco_annotation(mask, label, image_id, ann_id, score=None):
    """Annotation with an RLE segmentation; `area` is the bounding-box area, as the reference computes it.  Key order as the reference
    writes it: without a score the tile annotation of infer_wsi.py:613-626 (bbox first, id fifth), with a score the patch annotation of
    infer_patch.py:308-317 (id first, score last)."""
    rle = cocomask.encode(mask)
    bbox = cocomask.to_bbox(rle)
    if score is None:
        return {'bbox': bbox, 'area': bbox[2] * bbox[3], 'image_id': int(image_id), 'category_id': int(label), 'id': int(ann_id),
                'iscrowd': 0, 'segmentation': rle}
    return {'id': int(ann_id), 'bbox': bbox, 'area': bbox[2] * bbox[3], 'image_id': int(image_id), 'category_id': int(label),
            'iscrowd': 0, 'segmentation': rle, 'score': float(score)}


def write_json(path, obj, **kw):
    with open(path, 'w') as f:
        json.dump(obj, f, **kw)


def write_text_list(path, body):
    """A JSON list whose elements are already text (contours.ring_features_text ...): "[" + body + "]", the bytes json.dump writes."""
    with open(path, 'wb') as f:
        f.write(b'[')
        f.write(memoryview(body))
        f.write(b']')


# ----------------------------------------------------------------------------- SQLite contour table
class SqlContourWriter:
    """`--mode sql` (infer_wsi.py:637-660,683-693): one row per contour + an R-tree over the bounding boxes.  The
    reference never creates the table its INSERT targets, so the schema below is the column list of that INSERT with an
    integer primary key (the R-tree statement selects `id`)."""

    def __init__(self, path):
        self.conn = sqlite3.connect(path)
        self.c = self.conn.cursor()
        self.c.execute('DROP TABLE IF EXISTS contour;')
        self.c.execute('CREATE TABLE contour (id INTEGER PRIMARY KEY AUTOINCREMENT, annidx INTEGER, elementidx INTEGER, type TEXT, '
                       '"group" TEXT, score REAL, color TEXT, xmin INTEGER, ymin INTEGER, xmax INTEGER, ymax INTEGER, bbox_area INTEGER, '
                       'coords_x TEXT, coords_y TEXT, keep INTEGER);')

    def add(self, annidx, elementidx, ring, label, score, classes):
        ring = np.asarray(ring)
        xs, ys = ring[:, 0], ring[:, 1]
        self.c.execute('INSERT INTO contour (annidx, elementidx, type, "group", score, color, xmin, ymin, xmax, ymax, bbox_area, '
                       'coords_x, coords_y, keep) VALUES (?,?,?,?,?,?,?,?,?,?,?,?,?,?);',
                       (int(annidx), int(elementidx), 'polyline', classes[label], float(score), LINE_COLORS[label % len(LINE_COLORS)],
                        int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max()), int((xs.max() - xs.min()) * (ys.max() - ys.min())),
                        ','.join(str(int(v)) for v in xs), ','.join(str(int(v)) for v in ys), 1))

    def close(self):
        self.conn.commit()
        self.c.execute('DROP TABLE IF EXISTS rtree;')
        self.c.execute('CREATE VIRTUAL TABLE rtree USING rtree(id, xmin, xmax, ymin, ymax);')
        self.c.execute('INSERT INTO rtree (id, xmin, xmax, ymin, ymax) SELECT id, xmin, xmax, ymin, ymax FROM contour;')
        self.conn.commit()
        self.conn.close()
