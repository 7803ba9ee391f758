"""Batches in flight: consecutive tile batches of a slide are independent, so the WSI path keeps `depth` of them on the
GPU at once, each on its own engine (weights + workspace, 123 MB + ~0.3 GB per tile of max_batch) and its own HIP stream.
While one batch is in the under-filled tail of a launch (stage-3/4 GEMMs, the single-block proposal / detection kernels)
the other batch's kernels fill the idle CUs, and the host's result unpacking of batch i overlaps the GPU work of batch
i+1.  Measured on MI355X at B=16 (bench.py): 11.4 ms per batch alone, ≈ 10 ms per batch with four in flight (1400 -> ≈ 1600-1690
tiles/s); two identical chains started together stay in phase and gain nothing.  Host batches are copied from pageable memory:
that copy blocks the submitting thread for ≈ 4 ms per batch, but staging through pinned buffers made the slide loop twice as slow
(the CPU writes 3 MB into pinned memory at ≈ 0.5 GB/s, and with the host running ahead the copies of four streams got in the way of
the kernels: 1431 -> 688 tiles/s; a staging buffer of ordinary memory page-locked with hipHostRegister: 613; the pageable copy on
an idle per-slot copy stream: 1223), so the plain copy on the slot's stream stays.

Round 3: the engines of a pipeline run the throughput schedule (include/nuhtc_hip.h: one stream per engine, 256-row tiles), and the
default depth is six: the loop's host waits for the oldest batch before it reuses that slot, so with `depth` slots only depth - 1
batches are on the GPU while it unpacks and resubmits.  Full path from host tiles (tools/dev/pipe_rate.py): 1.49k tiles/s with four
slots on the runtime's four hardware queues, 1.71k with six, 1.83k with six and GPU_MAX_HW_QUEUES=16 (the slide tools set it);
bench.py's loop never waits on the host and peaks at four slots on four queues (1.90k).

The reference has no counterpart (its DataLoader overlaps only the CPU tile reads with the GPU, tools/infer_wsi.py:466-476)."""
import collections

import numpy as np
import torch

from .engine import Engine


class EnginePipeline:
    def __init__(self, state_dict, device=0, depth=6, **engine_kw):
        if depth > 1:           # engines that run beside each other: throughput schedule (include/nuhtc_hip.h)
            from . import hip
            engine_kw.setdefault('schedule', hip.SCHED_THROUGHPUT)
        self.engines = [Engine(state_dict, device=device, **engine_kw) for _ in range(depth)]
        self.device = self.engines[0].device
        # every engine runs on the stream it created next to its side streams (nuhtc_stream: three different pipes of the command
        # processor by construction, include/nuhtc_hip.h)
        self.streams = [e.stream for e in self.engines]
        self.pending = collections.deque()      # (slot, B, event, user tag)
        self.next = 0

    @property
    def depth(self):
        return len(self.engines)

    def submit(self, tiles, channel_mode, tag=None, export=False):
        """Enqueue one batch (host ndarray / tensor, or device tensor) on the next slot; returns the slot's engine.
        Blocks only when that slot still holds an uncollected batch.  export=True also enqueues Engine.export_async."""
        slot = self.next
        if any(p[0] == slot for p in self.pending):
            raise RuntimeError('pipeline slot still holds an uncollected batch: call collect() first')
        self.next = (slot + 1) % self.depth
        eng, st = self.engines[slot], self.streams[slot]
        st.wait_stream(torch.cuda.current_stream(self.device))
        # the host source of an asynchronous H2D copy must outlive the copy: keep it (and the device batch) referenced
        # until the batch is collected, callers may hand in temporaries
        src = torch.from_numpy(np.ascontiguousarray(tiles)) if isinstance(tiles, np.ndarray) else tiles
        with torch.cuda.stream(st):
            dev = eng.to_device(src)
            B = eng.infer_async(dev, channel_mode)
            if export:                      # contours + gather of the kept detections into pinned host buffers, still asynchronous
                eng.export_async(B)
            ev = torch.cuda.Event()
            ev.record(st)
        self.pending.append((slot, B, ev, tag, (dev, src)))
        return eng

    def full(self):
        return len(self.pending) >= self.depth

    def close(self):
        for e in self.engines:
            e.close()

    def collect(self):
        """Oldest submitted batch: waits for it and returns (engine, B, stream, tag); read the engine's output tensors on
        `stream` (torch.cuda.stream(stream)) before submitting to that slot again."""
        slot, B, ev, tag, _ = self.pending.popleft()
        ev.synchronize()
        with torch.cuda.stream(self.streams[slot]):      # (the flags are read on the slot's own, by now idle, stream)
            self.engines[slot].check()
        return self.engines[slot], B, self.streams[slot], tag

    def drain(self):
        while self.pending:
            yield self.collect()
