#!/usr/bin/env python
"""Headline benchmark: 256x256 tiles/s of the htc_lite_swin tile-inference path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16]

One step = one nuhtc_infer call on one batch of `--batch` synthetic 256x256x3 tiles already resident in HBM
(the full path: pre-processing, Swin-T, FPN, RPN, proposals, 3-stage cascade, detection NMS, mask head, paste,
per-tile mask-NMS).  Workload = BASELINE.json configs[1] ("PanNuke fold1 batch_size=16 256x256 tiles, 1xMI355X")
with seeded synthetic weights (models/pannuke.pth is not distributed) and synthetic nuclei tiles.
With N > 1 (launched by torch.distributed.run, one rank per GPU) tiles are sharded across ranks (weak scaling:
every rank processes its own batches, no data-path collective); the per-tile detection records of the last step
are all-gathered once over RCCL, as the WSI path does before the host-side merge.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
DOMINANT = 'gemm_kernel<3>'      # all Swin-T linears (N % 96 == 0): 96 % of the path's FLOPs


def _instance_parity(ref, got):
    """(bbox_results, segm_results) of the oracle and of the HIP path for one tile -> instances matched one to one with
    the same class, box IoU >= 0.999 and score within 1e-3; mask IoU of the matched pairs (BASELINE metric: per-instance IoU)."""
    cat = lambda r: (np.concatenate(r[0], 0), np.concatenate([np.full(len(b), c) for c, b in enumerate(r[0])]), [m for cl in r[1] for m in cl])
    (rb, rl, rm), (gb, gl, gm) = cat(ref), cat(got)
    out = dict(ref=len(rb), hip=len(gb), matched=0, mask_iou_min=1.0, mask_iou_below=0)
    if len(rb) == 0 or len(gb) == 0:
        return out
    x1 = np.maximum(rb[:, None, 0], gb[None, :, 0]); y1 = np.maximum(rb[:, None, 1], gb[None, :, 1])
    x2 = np.minimum(rb[:, None, 2], gb[None, :, 2]); y2 = np.minimum(rb[:, None, 3], gb[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ar = (rb[:, 2] - rb[:, 0]) * (rb[:, 3] - rb[:, 1]); ag = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
    iou = inter / np.maximum(ar[:, None] + ag[None, :] - inter, 1e-12)
    iou[rl[:, None] != gl[None, :]] = -1
    used = set()
    for i in np.argsort(-rb[:, 4]):
        j = int(np.argmax(iou[i]))
        if iou[i, j] >= 0.999 and j not in used and abs(rb[i, 4] - gb[j, 4]) < 1e-3:
            used.add(j)
            out['matched'] += 1
            u = np.logical_or(rm[i], gm[j]).sum()
            v = np.logical_and(rm[i], gm[j]).sum() / u if u else 1.0
            out['mask_iou_min'] = min(out['mask_iou_min'], float(v))
            out['mask_iou_below'] += int(v < 0.999)
    return out


def cpu_baseline(sd, tiles, n, eng=None, mode=1):
    """Oracle (CPU restatement of the reference path, oracle/model.py) timed on the host cores: reported next to
    the GPU number, never the thing shipped.  With `eng`, the same sample also serves as the parity check of the run:
    the HIP path's instances against the oracle's (north star: IoU >= 0.999 per instance, identical class ids)."""
    from oracle import model as O
    orc = O.Oracle(sd)
    threads = torch.get_num_threads()
    orc(tiles[:1], 1)  # warm-up (builds oracle/libnuhtc_oracle.so on first use)
    t0 = time.perf_counter()
    ref = orc(tiles[:n], mode)
    dt = time.perf_counter() - t0
    out = dict(value=n / dt, unit='tiles/s', cores=threads, kind='port',
               sample=f'{n} synthetic nuclei tiles, one batch, oracle/model.py fp32 torch-cpu + C RoIAlign/NMS, {dt:.1f} s')
    if eng is not None:
        eng.infer_async(eng.to_device(tiles[:n]), mode)
        got = eng.results(n)
        tot = dict(ref=0, hip=0, matched=0, mask_iou_min=1.0, mask_iou_below=0)
        for r, g in zip(ref, got):
            q = _instance_parity(r, g)
            for k in ('ref', 'hip', 'matched', 'mask_iou_below'):
                tot[k] += q[k]
            tot['mask_iou_min'] = min(tot['mask_iou_min'], q['mask_iou_min'])
        out['parity'] = dict(tiles=n, instances_oracle=tot['ref'], instances_hip=tot['hip'],
                             matched_same_class_box_iou_ge_0999=tot['matched'], matched_mask_iou_min=round(tot['mask_iou_min'], 6),
                             matched_mask_iou_below_0999=tot['mask_iou_below'],
                             note='detections before the per-tile margin / mask-NMS filter; a matched mask below 0.999 is a pixel whose probability sits on the 0.5 threshold (fp32 summation order), an unmatched instance one at the score threshold or a rank swap in a greedy NMS where two scores agree to an ulp')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-tiles', type=int, default=8)
    ap.add_argument('--in-flight', type=int, default=0,
                    help='also report the streaming rate with this many batches in flight (e.g. 3; off by default so that the '
                         'rocprofv3 summary of the default command sees every kernel without co-running kernels)')
    ap.add_argument('--fixed-load', action='store_true',
                    help='SURVEY 8d fixed-load mode: 1064 given RoIs and exactly 64 detections per tile (nuhtc_infer_fixed_load) instead of '
                         'the free-running proposal / detection counts of the synthetic weights')
    ap.add_argument('--roi-size', default='12,40', help='--fixed-load: RoI side range in network pixels (SURVEY: 12,40; 40x nuclei: ~40,100)')
    ap.add_argument('--gemm-shapes', action='store_true', help='add the per-shape GEMM timings to the JSON line')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)')
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    torch.cuda.set_device(local_rank)
    sd = weights.bench_state_dict()
    eng = Engine(sd, device=local_rank, max_batch=args.batch, tile=(256, 256))
    B = args.batch
    # every rank gets its own tiles (tile index space sharded contiguously across ranks)
    tiles_np = synth.nuclei_tiles(B, 256, start=rank * B)
    tiles = eng.to_device(tiles_np)
    mode = hip.CH_SWAP   # tools/infer_wsi.py channel handling
    if args.fixed_load:
        rois = torch.from_numpy(synth.fixed_load_rois(B, size=tuple(float(v) for v in args.roi_size.split(',')))).to(tiles.device)
        step_fn = lambda e=eng: e.infer_fixed_load_async(tiles, rois, 64, mode)
    else:
        step_fn = lambda e=eng: e.infer_async(tiles, mode)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_fn()
    eng.check()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn()
    if dist is not None:
        # one gather of the per-tile detection records (boxes, labels, counts, keep flags) for the host-side merge
        rec = torch.cat([eng.boxes.reshape(B, -1), eng.labels.float(), eng.keep.float(), eng.counts.float()[:, None]], 1)
        gathered = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.check()
    total_tiles = args.steps * B * world
    counts = eng.counts[:B].cpu().numpy()
    roi_counts = eng.buffer('roi_counts')[:B].cpu().numpy()

    # streaming rate: the same K steps with `--in-flight` batches on the GPU at once (one engine + HIP stream each, as the WSI
    # path runs: nuhtc_amd.pipeline).  Reported beside `value`, which stays the one-batch-at-a-time rate the per-kernel
    # numbers above belong to.
    pipelined = None
    if args.in_flight > 1:
        engs = [eng] + [Engine(sd, device=local_rank, max_batch=args.batch, tile=(256, 256)) for _ in range(args.in_flight - 1)]
        streams = [torch.cuda.Stream() for _ in engs]

        def run(k):
            for i in range(k):
                with torch.cuda.stream(streams[i % len(engs)]):
                    engs[i % len(engs)].infer_async(tiles, mode)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        run(args.warmup * len(engs))
        sync_all()
        t0 = time.perf_counter()
        run(args.steps)
        sync_all()
        dtp = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dtp], device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtp = float(t.item())
        for e in engs:
            e.check()
        pipelined = {'in_flight': len(engs), 'value': total_tiles / dtp, 'unit': 'tiles/s', 'ms_per_step': dtp / args.steps * 1e3,
                     'note': 'same K steps, consecutive batches overlapped on separate HIP streams (one engine each)'}

    # live per-kernel timing (HIP events on the launch stream) over the same workload, separate steps so the
    # event records do not perturb the headline number
    hip.profile_enable(True)
    prof_steps = max(2, min(5, args.steps))
    for _ in range(prof_steps):
        step_fn()
    prof_raw = hip.profile_read()
    hip.profile_enable(False)
    prof, shapes = {}, {}
    for tag, v in prof_raw.items():          # GEMM tags carry the shape: "gemm_kernel<3>|N288|K96"
        k = tag.split('|')[0]
        a = prof.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        for f in a:
            a[f] += v[f]
        if '|' in tag:
            shapes[tag] = dict(ms_per_step=round(v['ms'] / prof_steps, 3), tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1))
    dom = prof[DOMINANT]
    dur_ms = dom['ms'] / dom['launches']
    achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
    tot_ms = sum(v['ms'] for v in prof.values())
    breakdown = {k: round(v['ms'] / prof_steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}

    traffic = None
    tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_traffic.json')
    if os.path.exists(tf):   # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command (committed summary)
        traffic = json.load(open(tf))['hbm_bytes_per_launch']
    if rank == 0:
        out = {
            'metric': 'tiles/sec (256x256) whole-node', 'value': total_tiles / dt, 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'htc_lite_swin PanNuke config, batch_size=16 256x256 tiles per GPU (BASELINE configs[1]), full path '
                                   'incl. proposals, cascade, masks, per-tile mask-NMS' + (' [fixed load: 1064 RoIs, 64 detections per tile]' if args.fixed_load else ''), 'batch_per_gpu': B,
                       'weights': 'seeded synthetic (weights.bench_state_dict); pannuke.pth not distributed',
                       'tiles': 'synthetic nuclei tiles (nuhtc_amd.synth), resident in HBM',
                       'mean_rois_per_tile': float(roi_counts.mean()), 'mean_dets_per_tile': float(counts.mean())},
            'roofline': {'bound': 'mfma', 'kernel': DOMINANT + ' (Swin-T linears, fp32 v_mfma_f32_32x32x2_f32)', 'achieved': achieved,
                         'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': dom['bytes'] / dom['launches'],
                         'avg_launch_ms': dur_ms, 'launches_per_step': dom['launches'] // prof_steps,
                         'share_of_step_kernel_time': dom['ms'] / tot_ms},
            'kernel_ms_per_step': breakdown,
        }
        if pipelined:
            out['pipelined'] = pipelined
        if args.gemm_shapes:
            out['gemm_shapes'] = shapes
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, tiles_np, min(args.cpu_tiles, B), eng, mode)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
