"""dev: where the slide loop (nuhtc_amd.wsi.infer_tiles) spends its host time -- submit (tile slice + H2D enqueue + launches), the wait for
the oldest batch, the unpacking of its records -- and what the GPU does meanwhile (busy fraction from the batches' events).
    python tools/dev/r05_wsi_loop_profile.py [grid=60]"""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch

from nuhtc_amd import pipeline, synth, weights, wsi
from nuhtc_amd.apis import init_detector

G = int(sys.argv[1]) if len(sys.argv) > 1 else 60
band, y0 = synth.nuclei_canvas_parallel(G, rows=(0, G), workers=16)
tiles = synth.CanvasTiles(band, y0, G, 0, G * G)
ck = f'/tmp/r05_loop_{os.getpid()}.pth'
torch.save(dict(state_dict=weights.bench_state_dict(0)), ck)
model = init_detector(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'), ck, device='cuda:0', max_batch=16, bind_host=True)
model.opts.update(margin=2, min_area=10, mask_nms_thr=0.05)
wsi.infer_tiles(model, tiles[0:64], tiles.coords[:64], 16, 4)
torch.cuda.synchronize()
T = dict(submit=0.0, slice=0.0, collect_wait=0.0, unpack=0.0, read=0.0)
orig_submit, orig_collect = pipeline.EnginePipeline.submit, pipeline.EnginePipeline.collect
orig_unpack = wsi._unpack_packed


def submit(self, t, *a, **k):
    t0 = time.perf_counter()
    r = orig_submit(self, t, *a, **k)
    T['submit'] += time.perf_counter() - t0
    return r


def collect(self):
    t0 = time.perf_counter()
    r = orig_collect(self)
    T['collect_wait'] += time.perf_counter() - t0
    return r


def unpack(*a, **k):
    t0 = time.perf_counter()
    r = orig_unpack(*a, **k)
    T['unpack'] += time.perf_counter() - t0
    return r


pipeline.EnginePipeline.submit, pipeline.EnginePipeline.collect, wsi._unpack_packed = submit, collect, unpack
orig_getitem = type(tiles).__getitem__


def getitem(self, i):
    t0 = time.perf_counter()
    r = orig_getitem(self, i)
    T['slice'] += time.perf_counter() - t0
    return r


type(tiles).__getitem__ = getitem
t0 = time.perf_counter()
rec = wsi.infer_tiles(model, tiles, tiles.coords, 16, 4)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nb = (G * G + 15) // 16
print(f'{G * G} tiles in {dt:.3f} s = {G * G / dt:.0f} tiles/s; per batch of 16: {dt / nb * 1e3:.2f} ms wall')
for k, v in T.items():
    print(f'  {k:13s} {v:.3f} s  {v / nb * 1e3:.3f} ms per batch  {v / dt * 100:.1f} % of the loop')
print('  (submit includes slice; rest = final join of the parts)', round(dt - T['submit'] - T['collect_wait'] - T['unpack'], 3))
t0 = time.perf_counter()
parts = wsi.pack_records(rec)
print('pack_records', round(time.perf_counter() - t0, 3), 's for', len(rec['score']), 'records')
os.remove(ck)
