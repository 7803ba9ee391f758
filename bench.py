#!/usr/bin/env python
"""Headline benchmark: 256x256 tiles/s of the htc_lite_swin tile-inference path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16]

One step = one nuhtc_infer call on one batch of `--batch` synthetic 256x256x3 tiles already resident in HBM
(the full path: pre-processing, Swin-T, FPN, RPN, proposals, 3-stage cascade, detection NMS, mask head, paste,
per-tile mask-NMS).  The K timed steps run the way the slide loop runs them: `--in-flight` (default 4) batches on the GPU at
once, one engine + HIP stream each (nuhtc_amd.pipeline); `sequential` is the rate of the same K steps one batch at a time
(`--in-flight 0`), the mode every per-kernel figure of the line is measured in.  Workload = BASELINE.json configs[1] ("PanNuke fold1 batch_size=16 256x256 tiles, 1xMI355X")
with seeded synthetic weights (models/pannuke.pth is not distributed) and synthetic nuclei tiles.
With N > 1 (launched by torch.distributed.run, one rank per GPU) tiles are sharded across ranks (weak scaling:
every rank processes its own batches, no data-path collective); the detection records of the last step -- the layout the
WSI path ships: heads, ring vertices, bit-packed mask crops (nuhtc_amd.wsi.pack_records) -- are exchanged in ONE all-gather
over RCCL (nuhtc_amd.parallel.gather_blobs), inside the timed region, as before the host-side merge of a slide.
Prints ONE JSON line on rank 0.

Besides `value`: `roofline` (dominant kernel, live HIP-event timing), `roofline.pipeline_frac` (SURVEY 8d headline: executed
FLOP of the whole step / step time / fp32-MFMA peak), `kernel_groups` (per group: TFLOP/s / 157.3 for the dense groups,
algorithmic GB/s / 8000 for the gather / scan groups), `real_slide_roi_load` (the same step with 40-100 px RoIs, the size of
40x nuclei after the x2 resize), `cpu_baseline` (the oracle on the host cores, bounded sample) with the parity of the run.
"""
import argparse
import json
import os
import sys
import time

# (GPU_MAX_HW_QUEUES is left at the HIP runtime's default of 4: the four engines in flight run the throughput schedule, one stream
# each, and four streams on four queues -- one per pipe of the command processor -- measured 1895-1905 tiles/s against 1820-1837
# with 8 or 16 queues.  `config.hw_queues` records an exported value.)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
DOMINANT = 'gemm_kernel<3>'      # all Swin-T linears (N % 96 == 0): 96 % of the path's FLOPs
# kernel tags (csrc ProfScope) -> groups of SURVEY 8d; dense groups are priced against the fp32-MFMA roof, the others
# against HBM with their ALGORITHMIC bytes (what the op must read + write once)
GROUPS = {
    'dense_swin_linears': ('mfma', ['gemm_kernel<3>', 'swin_mlp', 'swin_lnqkv']),   # 96-column GEMMs + the fused stage-1 kernels (csrc/mlp.hip)
    'dense_convs_fcs': ('mfma', ['gemm_kernel<1>', 'gemm_kernel<2>', 'gemm_kernel<4>', 'patch_embed', 'attn_pool']),
    'attention': ('mfma', ['window_attn']),
    'layernorm_gathers': ('hbm', ['layernorm', 'merge_ln', 'preproc', 'sem_fuse', 'qkv_pad_rows']),
    'roi_gather': ('hbm', ['roi_feat7', 'roi_feat14']),
    'nms_scan': ('hbm', ['nms', 'rpn_select', 'cc_proposals', 'det_candidates', 'bbox_tail', 'paste', 'tile_post', 'build_rois']),
}


def cpu_baseline(sd, tiles, eng=None, mode=1, batch=4, timed=3, warm=2, warm_batches=1, host_cpus=None, sweep=(8, 16, 32, 64, 128), sweep_batch=None):
    """Oracle (CPU restatement of the reference path, oracle/model.py) timed on the host cores: reported next to the GPU
    number, never the thing shipped.  Bounded sample: one warm-up batch of 2 tiles, a THREAD SWEEP (one batch of `sweep_batch`
    tiles at each torch thread count of `sweep` the host has, plus all of them), then `timed` batches of `batch` tiles each at the
    best count, timed one by one, with the stage breakdown BASELINE.md section 3 asks for (backbone / FPN+RPN+semantic / proposals /
    cascade / mask / post).  With `eng`, the same tiles are the parity check of the run: the HIP path's instances against the oracle's
    with the strict comparator of the tests (tests/parity_util.py: exact counts, every tolerated disagreement proven to sit on a
    threshold)."""
    from oracle import model as O
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests'))
    import parity_util as P
    if host_cpus:       # the GPU legs ran on the GPU's NUMA node (hip.bind_host_thread); the CPU baseline gets every core the process started with
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), host_cpus)
            except OSError:
                pass
    orc = O.Oracle(sd)
    ncpu = len(os.sched_getaffinity(0))
    threads0 = torch.get_num_threads()
    for _ in range(warm_batches):
        orc(tiles[:warm], mode)        # warm-up (also builds oracle/libnuhtc_oracle.so on first use)
    # ---- thread sweep: torch's intra-op pool at P threads (the oracle's C RoIAlign / NMS and its Python RoI loops are single-threaded
    # whatever P is).  More threads than the dense stages can use cost more than they give: the rounds 1-4 figure (all 128 hardware
    # threads of the GPU box) was an accident of oversubscription.
    sb = sweep_batch or batch
    cand = sorted({p for p in sweep if p < ncpu} | {ncpu})
    swept, skipped = {}, []
    for pth in cand:                      # ascending; past the peak the rate only falls (0.03 tiles/s at 256 threads: two minutes per batch), so the sweep
        if swept and swept[max(swept)] < 0.6 * max(swept.values()):      # stops once a count has fallen below 60 % of the best so far
            skipped.append(pth)
            continue
        torch.set_num_threads(pth)
        t0 = time.perf_counter()
        orc(tiles[:sb], mode)
        swept[pth] = sb / (time.perf_counter() - t0)
    best = max(swept, key=swept.get)
    torch.set_num_threads(best)
    times, refs, vals, stage = [], [], [], {}
    n = min(batch * timed, len(tiles))
    for i in range(0, n, batch):
        t0 = time.perf_counter()
        r, it = orc(tiles[i:i + batch], mode, keep=True, timing=stage)
        times.append(time.perf_counter() - t0)
        refs += r
        vals += P.oracle_paste_values(O, it, tiles.shape[1:3])
    torch.set_num_threads(threads0)
    dt = sum(times)
    stage['other'] = max(0.0, dt - sum(stage.values()))
    order = ('preprocess', 'backbone', 'fpn_rpn_semantic', 'proposals', 'cascade', 'mask', 'post', 'other')
    dense = stage.get('backbone', 0) + stage.get('fpn_rpn_semantic', 0)
    out = dict(value=n / dt, unit='tiles/s', cores=best, threads=best, host_cpus=ncpu, kind='port',
               thread_sweep={str(k): round(v, 3) for k, v in swept.items()}, thread_sweep_skipped=skipped,
               stage_s_per_tile={k: round(stage.get(k, 0.0) / n, 4) for k in order},
               stage_share={k: round(stage.get(k, 0.0) / dt, 3) for k in order},
               dense_part_tiles_per_s=round(n / dense, 3) if dense else None,
               kind_note='"port" = oracle/model.py, a correctness tool: its dense stages are torch-cpu fp32 operators like the reference\'s, its proposal and RoI stages '
                         '(RPN NMS and RoIAlign in plain one-thread C, the attention RoI extractor as Python loops over levels and RoIs, paste) are not tuned and '
                         'dominate (`proposals` + `cascade`: see stage_share), more than they would in mmdet + mmcv -- the oracle is not optimised on purpose; dense_part_tiles_per_s is the figure comparable with the 2.6 tiles/s BASELINE.md measured for the reference\'s own dense modules on 8 vCPUs',
               sample=f'{timed} timed batches of {batch} synthetic nuclei tiles at {best} torch threads (best of the sweep {sorted(swept)}, one batch of {sb} each; not run, past the peak: {skipped}) after {warm_batches} warm-up batch(es) of {warm} '
                      f'(oracle/model.py, fp32 torch-cpu + C RoIAlign/NMS), {dt:.1f} s; per batch {[round(batch / t, 3) for t in times]} tiles/s' +
                      ('; the full protocol of SURVEY 8d (2 warm-up + 10 timed batches of 16 tiles)' if (batch, timed, warm, warm_batches) == (16, 10, 16, 2) else
                       '; bounded deviation from SURVEY 8d (2 warm-up + 10 timed batches of 16 tiles: minutes of host time; `bench.py --cpu-full`, profiles/r05_cpu_baseline_full.json)'))
    if eng is not None:
        got = []
        for i in range(0, n, eng.cfg.max_batch):
            k = eng.infer_async(eng.to_device(tiles[i:i + eng.cfg.max_batch]), mode)
            got += eng.results(k)
        tot = dict(n_ref=0, n_got=0, matched=0, mask_px_flipped=0, masks_below_0999=0)
        min_iou, max_dist, explained, failures = 1.0, 0.0, [], []
        for i, (r, g) in enumerate(zip(refs, got)):
            rep, fails = P.compare_strict(r, g, values=vals[i])
            for k in tot:
                tot[k] += rep[k]
            min_iou = min(min_iou, rep['min_mask_iou'])
            max_dist = max(max_dist, rep['max_px_threshold_dist'])
            explained += [f'tile {i}: {e}' for e in rep['explained']]
            failures += [f'tile {i}: {f}' for f in fails]
        out['parity'] = dict(tiles=n, instances_oracle=tot['n_ref'], instances_hip=tot['n_got'],
                             matched_same_class_box_iou_ge_0999=tot['matched'], mask_pixels_differing=tot['mask_px_flipped'],
                             matched_mask_iou_min=round(min_iou, 6), matched_mask_iou_below_0999=tot['masks_below_0999'],
                             farthest_flipped_pixel_from_threshold=max_dist, tolerated=explained, unexplained=failures,
                             passed=not failures,
                             note='detections before the per-tile margin / mask-NMS filter; every tolerated entry names the threshold it sits on (tests/parity_util.py)')
    return out


class PowerLog:
    """GPU power / clock / temperature at ~10 Hz from the amdgpu hwmon files in sysfs (no GPU call, a reader thread of this process).  A box
    shows every GPU of its host there: all are sampled and the busiest one -- the one this process ran on -- is reported."""

    def __init__(self, sysfs='/sys'):
        import glob
        import threading
        self.cards = []
        for card in sorted(glob.glob(os.path.join(sysfs, 'class/drm/card[0-9]*/device'))):
            hm = glob.glob(os.path.join(card, 'hwmon', 'hwmon*'))
            if hm and any(os.path.exists(os.path.join(hm[0], f)) for f in ('power1_average', 'power1_input')):
                lab = {}
                for p in glob.glob(os.path.join(hm[0], 'temp*_label')):
                    try:
                        lab[open(p).read().strip()] = p.replace('_label', '_input')
                    except OSError:
                        pass
                self.cards.append((card, hm[0], lab))
        self.rows = {c[0]: [] for c in self.cards}
        self.marks = []
        self.stop = threading.Event()
        self.t0 = time.time()
        self.th = threading.Thread(target=self._loop, daemon=True)
        if self.cards:
            self.th.start()

    @staticmethod
    def _read(path, scale=1.0):
        try:
            with open(path) as f:
                return float(f.read().split()[0]) * scale
        except (OSError, ValueError, IndexError):
            return None

    @staticmethod
    def _dpm(path):
        """MHz of the active level of a pp_dpm_* file ('1: 1250Mhz *'), None when unreadable."""
        try:
            with open(path) as f:
                for line in f:
                    if line.rstrip().endswith('*'):
                        return float(line.split(':')[1].lower().replace('mhz', '').replace('*', '').strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _loop(self):
        while not self.stop.is_set():
            for card, hm, lab in self.cards:
                pw = self._read(os.path.join(hm, 'power1_average'), 1e-6)
                if pw is None:
                    pw = self._read(os.path.join(hm, 'power1_input'), 1e-6)
                self.rows[card].append((time.time() - self.t0, pw, self._read(os.path.join(hm, 'freq1_input'), 1e-6),
                                        self._read(lab['junction'], 1e-3) if 'junction' in lab else None,
                                        self._read(lab['mem'], 1e-3) if 'mem' in lab else None, self._read(os.path.join(card, 'gpu_busy_percent')),
                                        self._dpm(os.path.join(card, 'pp_dpm_fclk')), self._dpm(os.path.join(card, 'pp_dpm_socclk')),
                                        self._dpm(os.path.join(card, 'pp_dpm_mclk'))))
            time.sleep(0.1)

    def mark(self, name):
        self.marks.append((name, time.time() - self.t0))

    def summary(self, csv_path=None, pci_bdf=None):
        """-> dict: per marked phase the mean / max power, mean shader clock and max temperatures; optionally the samples as CSV.
        `pci_bdf` ('0000:0a:00.0') names this process's GPU; without it (or without a match) the busiest card is taken -- which on a
        shared host may be somebody else's."""
        self.stop.set()
        if not self.cards:
            return {'available': False, 'note': 'no amdgpu hwmon files readable on this box'}
        self.th.join()
        best = None
        if pci_bdf:
            hit = [c for c in self.rows if os.path.realpath(c).lower().endswith(pci_bdf.lower())]
            best = hit[0] if hit else None
        matched = best is not None
        if best is None:
            best = max(self.rows, key=lambda c: sum((r[5] or 0) for r in self.rows[c]))
        rows = self.rows[best]
        hm = [c for c in self.cards if c[0] == best][0][1]
        out = {'available': True, 'device': os.path.realpath(best), 'device_chosen_by': 'pci bus id of this process\'s GPU' if matched else 'busiest card in sysfs', 'power_cap_w': self._read(os.path.join(hm, 'power1_cap'), 1e-6), 'sample_hz': 10, 'phases': {}}
        mean = lambda v: (sum(v) / len(v)) if v else None
        for i, (name, t_a) in enumerate(self.marks):
            t_b = self.marks[i + 1][1] if i + 1 < len(self.marks) else rows[-1][0] + 1
            if name.startswith('_'):
                continue
            seg = [r for r in rows if t_a <= r[0] < t_b]
            out['phases'][name] = {'seconds': round(t_b - t_a, 2), 'samples': len(seg),
                                   'power_w_mean': mean([r[1] for r in seg if r[1] is not None]), 'power_w_max': max([r[1] for r in seg if r[1] is not None], default=None),
                                   'sclk_mhz_mean': mean([r[2] for r in seg if r[2] is not None]),
                                   'temp_junction_c_max': max([r[3] for r in seg if r[3] is not None], default=None),
                                   'temp_mem_c_max': max([r[4] for r in seg if r[4] is not None], default=None),
                                   'fclk_mhz_mean': mean([r[6] for r in seg if r[6] is not None]), 'socclk_mhz_mean': mean([r[7] for r in seg if r[7] is not None]),
                                   'mclk_mhz_mean': mean([r[8] for r in seg if r[8] is not None])}
        if csv_path:
            with open(csv_path, 'w') as f:
                f.write('t_s,power_w,sclk_mhz,temp_junction_c,temp_mem_c,gpu_busy_pct,fclk_mhz,socclk_mhz,mclk_mhz,phase\n')
                for r in rows:
                    ph = [n for n, t in self.marks if t <= r[0]]
                    f.write(','.join('' if v is None else f'{v:.3f}' for v in r) + ',' + (ph[-1] if ph else '') + '\n')
        return out


def self_launch(n):
    """Run this script as `n` ranks (one per GPU) under torch.distributed.run as a CHILD process and relay its output: the JSON line
    of rank 0 goes to stdout, everything else to stderr.  Returns the job's exit code.  Nothing here initialises the GPU."""
    from nuhtc_amd import parallel
    return parallel.self_launch(n, __file__, sys.argv[1:], relay=lambda line: sys.stdout if line.startswith('{"metric"') else sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=4, help='cpu_baseline: tiles per timed oracle batch (3 timed batches)')
    ap.add_argument('--cpu-full', action='store_true', help="cpu_baseline by the full protocol of SURVEY 8d: 2 warm-up + 10 timed batches of 16 tiles (about 7 minutes of host time; the default line keeps the bounded sample)")
    ap.add_argument('--pipe', default='split', choices=['split', 'fp32'],
                    help="matrix pipe of the engine: 'split' (default: exact three-way bf16 operand split, six bf16 MFMAs per fp32 product step) "
                         "or 'fp32' (v_mfma_f32_32x32x2_f32)")
    ap.add_argument('--no-fp32-pipe', action='store_true', help='skip the secondary measurement of the same step on the fp32 MFMA kernels')
    ap.add_argument('--no-roi-load', action='store_true', help='skip the second workload (fixed load with 40-100 px RoIs)')
    ap.add_argument('--in-flight', type=int, default=4,
                    help='batches on the GPU at once in the timed region (one engine + HIP stream each, as the slide loop runs); 0 or 1 = '
                         'one batch at a time (the rocprofv3 / PMC passes of tools/dev/round_all.sh, so that their summaries see every '
                         'kernel without co-running kernels)')
    ap.add_argument('--fixed-load', action='store_true',
                    help='SURVEY 8d fixed-load mode: 1064 given RoIs and exactly 64 detections per tile (nuhtc_infer_fixed_load) instead of '
                         'the free-running proposal / detection counts of the synthetic weights')
    ap.add_argument('--roi-size', default='12,40', help='--fixed-load: RoI side range in network pixels (SURVEY: 12,40; 40x nuclei: ~40,100)')
    ap.add_argument('--gemm-shapes', action='store_true', help='add the per-shape GEMM timings to the JSON line')
    ap.add_argument('--no-settle', action='store_true', help='skip the untimed settle phase (profiling passes that serialise kernels)')
    ap.add_argument('--power-csv', default=None, help='write the 10 Hz power / clock / temperature samples of the run to this file (the summary is always on the line)')
    ap.add_argument('--no-force-collective', action='store_true', help='N = 1: do not form the one-rank RCCL communicator; the exchange short-circuits (rounds 1-4)')
    ap.add_argument('--no-host-bind', action='store_true', help="leave the process on whatever CPUs the scheduler picks instead of the GPU's NUMA node (A/B of hip.bind_host_thread)")
    ap.add_argument('--roi-sort', action='store_true', help='--fixed-load (dev): hand the RoIs over sorted by position (locality experiment)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher (one rank per GPU under
        # torch.distributed.run, the reference's pattern: tools/test.py:183,239 -> init_dist) BEFORE anything here touches the GPU,
        # relays rank 0's JSON line and exits with the job's code.  It never re-execs: the ranks are child processes.
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # stdout carries ONE JSON line.  Libraries write there too (RCCL prints a version banner on fd 1 when a communicator is created):
    # from here on fd 1 is stderr, and the line goes to a private copy of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or unset WORLD_SIZE and let bench.py launch them)')
    # test hooks for a box with fewer GPUs than ranks (tests/test_hip_api.py): every rank on device 0, gloo instead of RCCL
    if os.environ.get('NUHTC_ONE_DEVICE') == '1':
        local_rank = 0
    # the submitting thread runs on the NUMA node its GPU is attached to (Engine() would do it too; here it precedes the process group so
    # that RCCL's threads inherit the mask); the CPU baseline gets the original mask back
    cpus0 = os.sched_getaffinity(0)
    if args.no_host_bind:
        os.environ['NUHTC_HOST_AFFINITY'] = '0'
    from nuhtc_amd import hip as _hip
    host_bound = False
    if os.environ.get('NUHTC_HOST_AFFINITY', '1') != '0':
        try:
            host_bound = _hip.bind_host_thread(local_rank)
        except RuntimeError:
            pass
    cpus_gpu = os.sched_getaffinity(0)
    dist = None
    collective_note = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        backend = os.environ.get('NUHTC_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    elif not args.no_force_collective and os.environ.get('NUHTC_FORCE_COLLECTIVE', '1') != '0':
        # N = 1: the job still forms its RCCL communicator (of one rank) and the timed exchange goes through both all_gathers on device
        # buffers (nuhtc_amd.parallel.force_collective) -- the branch the N > 1 runs take, on the hardware this run has.  No scaling claim.
        import torch.distributed as _d
        from nuhtc_amd import parallel as _par
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        try:
            torch.cuda.set_device(local_rank)
            import datetime
            _d.init_process_group(os.environ.get('NUHTC_DIST_BACKEND', 'nccl'), init_method=f'tcp://127.0.0.1:{_par._free_port()}', rank=0, world_size=1,
                                  device_id=torch.device('cuda', local_rank), timeout=datetime.timedelta(seconds=120))
            os.environ['NUHTC_FORCE_COLLECTIVE'] = '1'
            dist = _d
        except Exception as e:                  # the communicator is evidence, not a dependency of the N = 1 line
            collective_note = f'process group of one rank not formed ({type(e).__name__}: {e}); the exchange short-circuits'
            print(collective_note, file=sys.stderr)

    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    plog = PowerLog() if rank == 0 else None          # 10 Hz power / clock / temperature beside every phase of the run (sysfs)
    mark = (lambda name: plog.mark(name)) if plog else (lambda name: None)
    mark('_setup')
    torch.cuda.set_device(local_rank)
    sd = weights.bench_state_dict()
    pipe = hip.PIPE_BF16_SPLIT if args.pipe == 'split' else hip.PIPE_FP32
    eng = Engine(sd, device=local_rank, max_batch=args.batch, tile=(256, 256), matrix_pipe=pipe)
    torch.cuda.set_stream(eng.stream)      # everything below runs on the first engine's own stream unless it says otherwise
    B = args.batch
    # every rank gets its own tiles (tile index space sharded contiguously across ranks)
    tiles_np = synth.nuclei_tiles(B, 256, start=rank * B)
    tiles = eng.to_device(tiles_np)
    mode = hip.CH_SWAP   # tools/infer_wsi.py channel handling
    if args.fixed_load:
        rois_np = synth.fixed_load_rois(B, size=tuple(float(v) for v in args.roi_size.split(',')))
        if args.roi_sort:
            key = ((rois_np[..., 1] + rois_np[..., 3]) / 2 // 32) * 1000 + (rois_np[..., 0] + rois_np[..., 2]) / 2
            rois_np = np.take_along_axis(rois_np, np.argsort(key, axis=1)[..., None], axis=1)
        rois = torch.from_numpy(np.ascontiguousarray(rois_np)).to(tiles.device)
        step_fn = lambda e=eng: e.infer_fixed_load_async(tiles, rois, 64, mode)
    else:
        step_fn = lambda e=eng: e.infer_async(tiles, mode)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A GPU that has been idle (a fresh box) runs its first second or so well below its sustained clocks: before the W
    # contractual warm-up steps, untimed steps are run until the step time has settled (three consecutive groups of 10 steps
    # within 1 % of each other, at most 8 s).  Nothing here is timed or counted.
    mark('settle')
    settle_steps, hist = 0, []
    t_settle = time.perf_counter()
    while not args.no_settle and time.perf_counter() - t_settle < 8.0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            step_fn()
        torch.cuda.synchronize()
        hist.append((time.perf_counter() - t1) / 10)
        settle_steps += 10
        if len(hist) >= 3 and max(hist[-3:]) <= 1.01 * min(hist[-3:]):
            break
    print('settle phase, ms per step by group of 10:', [round(h * 1e3, 2) for h in hist], file=sys.stderr)
    from nuhtc_amd import parallel, wsi

    def exchange(e):
        """The exchange of the WSI path: the last step's kept detections (engine `e` ran it) in the record layout a slide ships
        (heads, ring vertices, bit-packed mask crops), one all-gather (at N = 1 the packing runs too, the collective is a no-op)."""
        e.export_async(B)
        torch.cuda.current_stream().synchronize()
        parts = []
        g = e.export_read()
        if g['n']:
            wsi._unpack_packed(e, g, 0, np.zeros((B, 2), np.int64), parts)
        rec = wsi._records_from_parts(parts)
        parts = wsi.pack_records(rec, tile_base=rank * B) + [torch.tensor([rank], dtype=torch.int32)]
        return parallel.gather_blobs([t.to(tiles.device) for t in parts])

    # The timed region runs the K steps the way the slide loop runs them (nuhtc_amd.pipeline, tools/infer_wsi.py): `--in-flight`
    # batches on the GPU at once, one engine + HIP stream each, consecutive steps on consecutive engines.  A dense launch is an
    # MFMA-bound phase followed by an HBM-bound phase, and only another batch's kernels fill the idle resource.
    # `--in-flight 0 / 1` times one batch at a time instead (the profiling passes: per-kernel durations need that); the
    # one-batch-at-a-time rate of the same K steps is always measured right after and reported as `sequential`.
    depth = max(1, args.in_flight)
    # engines that run beside each other use the throughput schedule (what nuhtc_amd.pipeline.EnginePipeline sets); `eng`, the
    # engine of `sequential` and of every per-kernel figure, keeps the default (latency) schedule
    if depth > 1:
        engs = [Engine(sd, device=local_rank, max_batch=args.batch, tile=(256, 256), matrix_pipe=pipe, schedule=hip.SCHED_THROUGHPUT) for _ in range(depth)]
    else:
        engs = [eng]
    streams = [e.stream for e in engs]

    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % depth]):
                step_fn(engs[i % depth])
    for st in streams:
        if st != torch.cuda.current_stream():
            st.wait_stream(torch.cuda.current_stream())
    for e in engs:              # untimed: first-call allocations of these engines
        if e is not eng:
            step_fn(e)
    torch.cuda.synchronize()
    mark('warmup')
    run(args.warmup)
    last = engs[(args.steps - 1) % depth]          # the engine that will run step K
    for st in streams:
        if st != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_stream(st)
    exchange(last)      # untimed: the first call allocates the pinned export buffers and loads lazily-built device code
    for e in engs:
        e.check()
    sync_all()
    mark('timed_in_flight' if depth > 1 else 'timed_sequential')
    t0 = time.perf_counter()
    run(args.steps)
    for st in streams:
        if st != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_stream(st)
    t_ex0 = time.perf_counter()
    gathered = exchange(last)            # once, inside the timed region: the records of step K
    exchange_s = time.perf_counter() - t_ex0      # (includes the wait for the steps still in flight on the engine that ran step K)
    ranks_seen = sorted(int(g[-1][0]) for g in gathered)
    gathered_records = int(sum(g[0].shape[0] for g in gathered))
    gathered_bytes = int(sum(t.numel() * t.element_size() for g in gathered for t in g))
    sync_all()
    dt = time.perf_counter() - t0
    mark('_after_timed')
    dt = max_over_ranks(dt)
    for e in engs:
        e.check()
    total_tiles = args.steps * B * world
    counts = eng.counts[:B].cpu().numpy()
    roi_counts = eng.buffer('roi_counts')[:B].cpu().numpy()

    # What the K-step window above leaves out: it starts with the `depth` engines empty and ends when the last of them has drained, so
    # depth - 1 steps' worth of the window run on a partly filled GPU (K = 20 at the driver's settings).  `value_steady` is the same step
    # in the same run with the pipeline already full: 2 * depth untimed steps, then the time between the completion of step F - 1 and of
    # step F + S - 1 (HIP events on the streams those two steps ran on; S >= 100, a multiple of depth so both events sit on one stream),
    # with 2 * depth further steps queued behind so that the GPU stays full until the second event.  `value` stays the contractual figure.
    steady = None
    if depth > 1:
        F = 2 * depth
        S = -(-max(100, args.steps) // depth) * depth
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        mark('steady_in_flight')
        for i in range(F + S + F):
            with torch.cuda.stream(streams[i % depth]):
                step_fn(engs[i % depth])
                if i == F - 1:
                    ev_a.record()
                elif i == F + S - 1:
                    ev_b.record()
        sync_all()
        mark('_after_steady')
        ds = max_over_ranks(ev_a.elapsed_time(ev_b) * 1e-3)
        for e in engs:
            e.check()
        steady = {'value': S * B * world / ds, 'ms_per_step': ds / S * 1e3, 'steps': S, 'untimed_steps_before_and_after': F,
                  'fill_drain_equivalent_steps': round((dt - args.steps * ds / S) / (ds / S), 2)}

    # the step at the RoI sizes of a real slide, with the same engines in flight (throughput schedule): 40-100 px (40x nuclei after the
    # x2 resize) and 100-200 px (clumps / component proposals); the sequential figures and the RoI kernels' share follow further down
    roi_in_flight = {}
    if depth > 1 and not args.no_roi_load and not args.fixed_load:
        for key, size in (('40_100', (40.0, 100.0)), ('100_200', (100.0, 200.0))):
            rois_b = torch.from_numpy(synth.fixed_load_rois(B, size=size)).to(tiles.device)
            kf = max(8, min(40, args.steps))

            def run_fixed(k):
                for i in range(k):
                    with torch.cuda.stream(streams[i % depth]):
                        engs[i % depth].infer_fixed_load_async(tiles, rois_b, 64, mode)
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
            run_fixed(2 * depth)
            sync_all()
            mark('roi_load_in_flight_' + key)
            t0 = time.perf_counter()
            run_fixed(kf)
            sync_all()
            df = max_over_ranks(time.perf_counter() - t0)
            roi_in_flight[key] = {'value': kf * B * world / df, 'unit': 'tiles/s', 'ms_per_step': df / kf * 1e3, 'steps': kf, 'batches_in_flight': depth}
        mark('_after_roi_in_flight')
        for e in engs:
            e.check()

    # the same K steps one batch at a time (what the per-kernel numbers below belong to)
    sequential = None
    if depth > 1:
        for e in engs:
            e.close()
        engs = [eng]
        sync_all()
        mark('sequential')
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        sync_all()
        dts = time.perf_counter() - t0
        mark('_after_sequential')
        dts = max_over_ranks(dts)
        sequential = {'value': total_tiles / dts, 'unit': 'tiles/s', 'ms_per_step': dts / args.steps * 1e3, 'steps': args.steps,
                      'note': 'the same K steps one batch at a time; roofline / kernel_ms_per_step / kernel_groups are measured in this mode'}
    else:
        dts = dt

    # the shader clock this box holds under the sequential step (the chip lowers its clock under matrix load and boxes differ:
    # a per-kernel fraction is only comparable between runs together with this figure): a one-wave probe on its own stream
    # (s_memtime against the 100 MHz s_memrealtime) beside untimed steps
    mark('_clock_probe')
    k_clk = max(3, min(20, args.steps))
    # The probe needs a hardware queue of its own: a stream that shares a queue with the engine runs the probe alone, ahead of the
    # steps (it then reports the idle clock).  Streams are dealt round the runtime's queues in an order this script does not control,
    # so candidates (non-blocking streams, never the legacy null stream) are tried until probe + steps take no longer than the steps.
    shader_clock_ghz, probe_note = None, 'no candidate stream ran beside the engine (probe serialised with the steps on every one): not measured'
    t_steps = k_clk * dts / args.steps
    cands = [torch.cuda.Stream(device=tiles.device) for _ in range(6)]
    for ci, cst in enumerate(cands):
        probe = hip.ClockProbe(local_rank, stream=cst)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        probe.start(0.8 * t_steps * 1e3)
        for _ in range(k_clk):
            step_fn()
        ghz = probe.ghz()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if wall < 1.35 * t_steps:
            shader_clock_ghz, probe_note = ghz, f'candidate stream {ci}: probe + {k_clk} steps took {wall / t_steps:.2f} x the steps alone (concurrent)'
            break

    # live per-kernel timing (HIP events on the launch stream) over the same workload, separate steps so the
    # event records do not perturb the headline number
    mark('per_kernel_events')
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
    # round 5: the QKV / fc1 launches of Swin stages 2-4 carry their LayerNorm in the A path (tag suffix "|ln"), the proj / fc2 / patch-merging
    # launches leave the LayerNorm partials in their epilogue ("|stats") -- work round 4 booked under `layernorm`.  The launches of the same
    # kernel with neither duty are priced separately so that the fraction of the kernel itself stays comparable across rounds.
    dom_plain = dict(launches=0, ms=0.0, flops=0.0)
    dom_ln = dict(launches=0, ms=0.0, flops=0.0)
    for tag, v in prof_raw.items():
        if tag.split('|')[0] == DOMINANT:
            tgt = dom_ln if ('|ln' in tag or '|stats' in tag) else dom_plain
            for f in tgt:
                tgt[f] += v[f]
    dom_split = {name: ({'launches_per_step': d['launches'] // prof_steps, 'ms_per_step': round(d['ms'] / prof_steps, 3),
                         'achieved': d['flops'] / (d['ms'] * 1e-3) / 1e12, 'frac': d['flops'] / (d['ms'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS} if d['launches'] else None)
                 for name, d in (('without_layernorm_duty', dom_plain), ('with_layernorm_in_a_path_or_statistics_epilogue', dom_ln))}
    tot_ms = sum(v['ms'] for v in prof.values())
    breakdown = {k: round(v['ms'] / prof_steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}

    # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes of this command
    # (tools/dev/round_traffic.sh); the committed summary is used only while it belongs to the GEMM source of this build
    traffic, traffic_note = None, 'no PMC summary for this build (rocprofv3 --pmc passes are run separately: tools/dev/round_traffic.sh)'
    here = os.path.dirname(os.path.abspath(__file__))
    import glob
    import hashlib
    sha = hashlib.sha1(open(os.path.join(here, 'nuhtc_amd', 'csrc', 'gemm.hip'), 'rb').read()).hexdigest()
    for tf in sorted(glob.glob(os.path.join(here, 'profiles', 'r*_traffic.json')), reverse=True):       # newest round first
        tj = json.load(open(tf))
        if tj.get('gemm_hip_sha1') == sha:
            traffic = tj['hbm_bytes_per_launch']
            traffic_note = f"profiles/{os.path.basename(tf)} (FETCH_SIZE x2 + WRITE_SIZE passes, gemm.hip {sha[:10]})"
            break
    else:
        traffic_note = 'no profiles/r*_traffic.json belongs to this gemm.hip: not reported (tools/dev/round_all.sh refreshes it)'
    # per-group fractions: dense groups against the fp32-MFMA roof, gather / scan groups against HBM with their
    # algorithmic bytes; `ms` = sum of launch durations per step (the RPN branch runs beside the semantic branch, so the sum
    # over groups exceeds the step)
    # algorithmic bytes of the kernels whose work depends on device-side counts (RoIs, detections): from the last step's counts
    R = int(roi_counts.sum())
    D = int(counts.sum())
    rois2 = eng.buffer('rois')[:R].cpu().numpy() if not args.fixed_load else None   # the cascade's RoI list (after its in-place refinements)
    px = lambda wh, s: (np.ceil(wh[:, 0] / s) + 2) * (np.ceil(wh[:, 1] / s) + 2)
    if rois2 is not None and R:
        wh = rois2[:, 3:5] - rois2[:, 1:3]
        roi7 = 3.0 * float(((px(wh, 4) + px(wh, 8)) * 256 + 7 * 7 * 64 * 4).sum())   # footprint on x0+sem / x1 + the 7x7x64 output, 3 stages
    else:
        roi7 = 3.0 * R * (2 * 36 * 256 + 12544)
    dyn_bytes = {
        'roi_feat7': roi7,
        'roi_feat14': D * (2 * 36 * 256 + 14 * 14 * 64 * 4.0),
        'rpn_select': B * (21760 * 15 * 4 + 9768 * 20.0),            # objectness + deltas of every anchor position in, candidates out
        'nms': B * (9768 * 20 + 1000 * 20.0) + R * 5 * 24.0 + D * 24.0,
        'cc_proposals': B * (128 * 128 * 4 + 3 * 512 * 512 / 8.0),
        'bbox_tail': 3.0 * R * (20 * 4 + 5 * 4),
        'det_candidates': R * (16 * 4 + 5 * 4.0),
        'paste': D * (28 * 28 * 4 + 256 * 256 / 8.0),
        'tile_post': D * (256 * 256 / 8.0 + 32),
    }
    for t, by in dyn_bytes.items():
        if t in prof and prof[t]['bytes'] == 0:
            prof[t]['bytes'] = by * prof_steps
    groups, seen = {}, set()
    for gname, (bound, tags) in GROUPS.items():
        v = [prof[t] for t in tags if t in prof]
        seen.update(t for t in tags if t in prof)
        if not v:
            continue
        ms = sum(x['ms'] for x in v)
        fl, by = sum(x['flops'] for x in v), sum(x['bytes'] for x in v)
        g = {'bound': bound, 'ms_per_step': round(ms / prof_steps, 3)}
        if bound == 'mfma':
            g['tflops'] = round(fl / (ms * 1e-3) / 1e12, 2)
            g['frac_of_fp32_mfma'] = round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        else:
            g['algorithmic_gbps'] = round(by / (ms * 1e-3) / 1e9, 1)
            g['frac_of_hbm'] = round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
        groups[gname] = g
    other = [t for t in prof if t not in seen]
    if other:
        groups['other'] = {'ms_per_step': round(sum(prof[t]['ms'] for t in other) / prof_steps, 3), 'tags': other}
    step_flops = sum(v['flops'] for v in prof.values()) / prof_steps          # FLOP the engine really executes per step
    pipeline_frac = step_flops / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS
    pipeline_frac_seq = step_flops / (dts / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS

    # second workload, on the record every round: the same step at the RoI sizes of a real 40x slide (40-100 px after the x2
    # resize) instead of the synthetic weights' ~20 px boxes -- the 7x7 RoI features are the data-dependent part of the path
    roi_load = None
    if not args.no_roi_load and not args.fixed_load:
        def seq_fixed(size):
            rois_b = torch.from_numpy(synth.fixed_load_rois(B, size=size)).to(tiles.device)
            k2 = max(5, min(20, args.steps))
            for _ in range(2):
                eng.infer_fixed_load_async(tiles, rois_b, 64, mode)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(k2):
                eng.infer_fixed_load_async(tiles, rois_b, 64, mode)
            sync_all()
            d2 = max_over_ranks(time.perf_counter() - t0)
            hip.profile_enable(True)
            for _ in range(2):
                eng.infer_fixed_load_async(tiles, rois_b, 64, mode)
            p2 = hip.profile_read()
            hip.profile_enable(False)
            roi_ms = sum(v['ms'] for k, v in p2.items() if k.split('|')[0] in ('roi_feat7', 'roi_classify')) / 2
            return {'value': k2 * B * world / d2, 'unit': 'tiles/s', 'ms_per_step': d2 / k2 * 1e3, 'steps': k2, 'roi_feat7_ms_per_step': round(roi_ms, 3)}
        mark('roi_load_sequential')
        roi_load = {'workload': 'fixed load: 1064 given RoIs per tile with sides 40-100 network px, 64 detections per tile (nuhtc_infer_fixed_load)'}
        roi_load.update(seq_fixed((40.0, 100.0)))
        roi_load['note'] = '`value` here is the one-batch-at-a-time rate; `in_flight` is the same load with the engines of the headline in flight'
        if '40_100' in roi_in_flight:
            roi_load['in_flight'] = roi_in_flight['40_100']
        big = {'workload': 'the same with RoI sides 100-200 network px (clumps, component proposals)'}
        big.update(seq_fixed((100.0, 200.0)))
        if '100_200' in roi_in_flight:
            big['in_flight'] = roi_in_flight['100_200']
        roi_load['roi_100_200'] = big
        mark('_after_roi_load')
    # the same step on the fp32 MFMA kernels (NUHTC_PIPE_FP32), on the record beside the default pipe: same weights, same tiles
    fp32_pipe = None
    if args.pipe == 'split' and not args.no_fp32_pipe and not args.fixed_load:
        mark('fp32_pipe')
        e32 = Engine(sd, device=local_rank, max_batch=args.batch, tile=(256, 256), matrix_pipe=hip.PIPE_FP32)
        k3 = max(5, min(30, args.steps))
        sync_all()
        with torch.cuda.stream(e32.stream):          # (its own stream, like every engine of this run)
            for _ in range(3):
                e32.infer_async(tiles, mode)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(k3):
                e32.infer_async(tiles, mode)
            sync_all()
            d3 = time.perf_counter() - t0
            hip.profile_enable(True)
            for _ in range(2):
                e32.infer_async(tiles, mode)
            p3 = hip.profile_read()
            hip.profile_enable(False)
        d3 = max_over_ranks(d3)
        g3 = [v for k, v in p3.items() if k.split('|')[0] == DOMINANT]
        a3 = sum(v['flops'] for v in g3) / (sum(v['ms'] for v in g3) * 1e-3) / 1e12
        # do the two pipes decide alike?  detections of the same batch, slot by slot
        eng.infer_async(tiles, mode)
        sync_all()
        same_counts = bool(torch.equal(eng.counts[:B], e32.counts[:B]))
        nmax = int(eng.counts[:B].max())
        live = (torch.arange(nmax, device=eng.counts.device)[None, :] < eng.counts[:B, None])[..., None]    # slots past a tile's count hold stale rows
        dbox = float(((eng.boxes[:B, :nmax] - e32.boxes[:B, :nmax]).abs() * live).max()) if same_counts and nmax else None
        same_labels = bool(torch.equal(eng.labels[:B, :nmax] * live[..., 0], e32.labels[:B, :nmax] * live[..., 0])) if same_counts else False
        fp32_pipe = {'value': k3 * B * world / d3, 'unit': 'tiles/s', 'ms_per_step': d3 / k3 * 1e3, 'steps': k3,
                     'roofline_frac_dominant_kernel': a3 / PEAK_F32_MFMA_TFLOPS, 'achieved_tflops_dominant_kernel': a3,
                     'same_detection_counts_as_default_pipe': same_counts, 'same_labels': same_labels, 'max_abs_box_or_score_difference': dbox,
                     'note': 'NUHTC_PIPE_FP32: every matrix product on v_mfma_f32_32x32x2_f32 (the kernels of round 1)'}
        e32.close()
    if rank == 0:
        out = {
            'metric': 'tiles/sec (256x256) whole-node', 'value': total_tiles / dt, 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'settle_steps_before_warmup': settle_steps, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': ('fp32 operands, results and accumulation everywhere; matrix products on the bf16 matrix pipe with every fp32 operand split exactly into '
                           'three bf16 numbers (six exact bf16 products per fp32 product; the three dropped cross terms are at most 2^-23 |a b| together): measured error against fp64 at or '
                           'below the fp32 MFMA chain (tests/test_hip_dense.py::test_split_bf16_pipe_is_fp32_arithmetic)') if args.pipe == 'split' else
                          'fp32 MFMA (v_mfma_f32_32x32x2_f32) for every matrix product',
            'config': {'workload': 'htc_lite_swin PanNuke config, batch_size=16 256x256 tiles per GPU (BASELINE configs[1]), full path '
                                   'incl. proposals, cascade, masks, per-tile mask-NMS' + (' [fixed load: 1064 RoIs, 64 detections per tile]' if args.fixed_load else ''), 'batch_per_gpu': B,
                       'weights': 'seeded synthetic (weights.bench_state_dict); pannuke.pth not distributed',
                       'tiles': 'synthetic nuclei tiles (nuhtc_amd.synth), resident in HBM',
                       'batches_in_flight': depth, 'schedule': 'NUHTC_SCHED_THROUGHPUT for the engines in flight (one stream, 256-row tiles); NUHTC_SCHED_LATENCY for `sequential` and the per-kernel figures' if depth > 1 else 'NUHTC_SCHED_LATENCY', 'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'runtime default (4)'),
                       'mean_rois_per_tile': float(roi_counts.mean()), 'mean_dets_per_tile': float(counts.mean())},
            'roofline': {'bound': 'mfma', 'kernel': DOMINANT + (' (Swin-T linears: gemm_split_kernel<3,0>, 6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 fp32 product)' if args.pipe == 'split'
                                                                else ' (Swin-T linears, fp32 v_mfma_f32_32x32x2_f32)'), 'achieved': achieved,
                         'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS,
                         'peak_note': 'achieved = ALGORITHMIC fp32 FLOP (2 M N K) / launch time; peak = dense fp32 MFMA (the dtype of the path)',
                         'matrix_pipe': ({'instruction': 'v_mfma_f32_32x32x16_bf16', 'executed_tflops': 6 * achieved, 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': 6 * achieved / PEAK_BF16_MFMA_TFLOPS,
                                          'fp32_equivalent_ceiling_tflops': PEAK_BF16_MFMA_TFLOPS / 6} if args.pipe == 'split' else
                                         {'instruction': 'v_mfma_f32_32x32x2_f32', 'executed_tflops': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'frac': achieved / PEAK_F32_MFMA_TFLOPS}),
                         'traffic': traffic, 'traffic_source': traffic_note,
                         'pipeline_frac': pipeline_frac, 'pipeline_frac_sequential': pipeline_frac_seq, 'pipeline_gflop_per_tile': step_flops / B / 1e9,
                         'algorithmic_bytes_per_launch': dom['bytes'] / dom['launches'],
                         'shader_clock_ghz_under_step': shader_clock_ghz,
                         'shader_clock_probe': probe_note,
                         'shader_clock_note': 's_memtime / s_memrealtime of a one-wave probe running beside untimed sequential steps (2.4 GHz nominal); '
                                              'the dense launches are clock-limited, so fractions of different boxes compare only at equal clock',
                         'avg_launch_ms': dur_ms, 'launches_per_step': dom['launches'] // prof_steps,
                         'by_layernorm_duty': dom_split,
                         'layernorm_note': 'since round 5 the LayerNorms of Swin stages 2-4 run inside these launches (A path of QKV / fc1, statistics in the epilogue of proj / fc2 / patch merging): '
                                           '`frac` prices ALL launches of the tag with their algorithmic GEMM FLOP only; compare `gemm_kernel<3>` + `layernorm` of kernel_ms_per_step across rounds',
                         'share_of_step_kernel_time': dom['ms'] / tot_ms},
            'kernel_ms_per_step': breakdown,
            'kernel_groups': groups,
            'exchange': {'collective': 'all_gather (header) + all_gather (one packed byte buffer): nuhtc_amd.parallel.gather_blobs',
                         'backend': (dist.get_backend() if dist is not None else 'none: one rank, short-circuit'),
                         'communicator_ranks': (dist.get_world_size() if dist is not None else 1),
                         **({'note': collective_note} if collective_note else {}),
                         'ranks_seen': ranks_seen, 'records': gathered_records, 'bytes': gathered_bytes, 'seconds_inside_timed_region': round(exchange_s, 4),
                         'layout': 'head f64[n,9] | ring vertices i32[*,2] | crop boxes i64[n,6] | bit-packed mask crops i32[*] | rank id'},
        }
        if steady:
            out['value_steady'] = steady['value']
            out['value_steady_detail'] = {**steady, 'unit': 'tiles/s',
                                          'note': 'the same step, same run, pipeline already full (HIP events around S steps with 2 x batches_in_flight untimed steps before and after); '
                                                  '`value` is the contractual K-step window, which starts with the engines empty and ends when they have drained'}
            out['pipeline_fill_drain_steps'] = depth - 1
            out['value_note'] = ('`value` times K steps through a pipeline of `batches_in_flight` engines that starts empty and drains inside the timed region, and includes the '
                                 'exchange (at N = 1: two all_gathers on a one-rank RCCL communicator unless --no-force-collective); `value_steady` is the steady-state rate')
        if roi_load:
            out['real_slide_roi_load'] = roi_load
            if 'in_flight' in roi_load:
                # first-class beside `value`: the synthetic weights' RoIs are ~20 px, nuclei of a real 40x slide give 40-100 px RoIs after
                # the x2 resize, and the RoI stage is the one part of the path whose cost follows the data -- THIS is the PanNuke-like rate
                out['value_real_slide_roi_load'] = {'value': roi_load['in_flight']['value'], 'unit': 'tiles/s', 'ms_per_step': roi_load['in_flight']['ms_per_step'],
                                                    'batches_in_flight': roi_load['in_flight']['batches_in_flight'],
                                                    'workload': '1064 given RoIs of 40-100 network px and 64 detections per tile, same step, same engines in flight as `value`',
                                                    'ratio_to_value': roi_load['in_flight']['value'] / (total_tiles / dt)}
        if fp32_pipe:
            out['fp32_mfma_pipe'] = fp32_pipe
        if sequential:
            out['sequential'] = sequential
        if args.gemm_shapes:
            out['gemm_shapes'] = shapes
        mark('cpu_baseline_host')
        if world == 1 and not args.no_cpu_baseline:
            if args.cpu_full:
                out['cpu_baseline'] = cpu_baseline(sd, synth.nuclei_tiles(160, 256, start=0), eng, mode, batch=16, timed=10, warm=16, warm_batches=2, host_cpus=cpus0)
            else:
                out['cpu_baseline'] = cpu_baseline(sd, tiles_np, eng, mode, batch=args.cpu_batch, host_cpus=cpus0)
        bdf = None
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        except Exception:
            pass
        out['power'] = plog.summary(args.power_csv, bdf)
        node = None
        try:
            node = int(open(f'/sys/bus/pci/devices/{bdf}/numa_node').read())
        except (OSError, ValueError, TypeError):
            pass
        out['host'] = dict(submitting_thread_bound_to_gpu_numa_node=bool(host_bound), gpu_pci=bdf, gpu_numa_node=node, cpus_during_gpu_legs=len(cpus_gpu), cpus_of_process=len(cpus0),
                           note='hip.bind_host_thread (nuhtc_bind_host_thread): from the other socket every dispatch packet costs the command processor 1.4-2.9 us more '
                                '(0.3-0.4 ms per sequential step); the CPU baseline runs on the process\'s original CPUs')
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
