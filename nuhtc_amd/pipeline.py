"""Batches in flight: consecutive tile batches of a slide are independent, so the WSI path keeps `depth` of them on the
GPU at once, each on its own engine (weights + workspace, 123 MB + ~0.3 GB per tile of max_batch) and its own HIP stream.
While one batch is in the under-filled tail of a launch (stage-3/4 GEMMs, the single-block proposal / detection kernels)
the other batch's kernels fill the idle CUs, and the host's result unpacking of batch i overlaps the GPU work of batch
i+1.  Measured on MI355X at B=16 (bench.py): 11.4 ms per batch alone, ≈ 10 ms per batch with four in flight (1400 -> ≈ 1600-1690
tiles/s); two identical chains started together stay in phase and gain nothing.  Host batches are copied from pageable memory:
that copy blocks the submitting thread for ≈ 4 ms per batch, but staging through pinned buffers made the slide loop twice as slow
(the CPU writes 3 MB into pinned memory at ≈ 0.5 GB/s, and with the host running ahead the copies of four streams got in the way of
the kernels: 1431 -> 688 tiles/s; a staging buffer of ordinary memory page-locked with hipHostRegister: 613; the pageable copy on
an idle per-slot copy stream: 1223), so the plain copy on the slot's stream stays.  Round 5 measured it once more with a ring of
page-locked buffers per slot allocated ONCE (per_slot + 1 host and device buffers, host copy into the ring, asynchronous copy on the
slot's stream, submit() never waiting for the GPU): 781-916 against 1859-1869 tiles/s on a 3600-tile slide, 896 against 1950 on the
10 000-tile slide (profiles/r05_upload_ab.txt) -- the host's copy INTO page-locked memory takes 12-16 ms per 3 MB batch on these hosts.

Round 3: the engines of a pipeline run the throughput schedule (include/nuhtc_hip.h: one stream per engine, 256-row tiles), and a
slot holds up to two exported batches (see EnginePipeline.per_slot): with one batch per slot the batches in flight finish together
and the GPU idles while the host refills the slots.  Slide level from host tiles (tools/bench_wsi.py, 10 000 tiles): 1.49k tiles/s at
the start of the round, 1.80k now with four engines and GPU_MAX_HW_QUEUES=16 (the slide tools set it; 1.57k on the runtime's four
queues, where the caller's stream shares a queue with an engine; 1.73k with three engines on four queues); bench.py's loop never
waits on the host and runs four engines on four queues at 1.90k.

The reference has no counterpart (its DataLoader overlaps only the CPU tile reads with the GPU, tools/infer_wsi.py:466-476)."""
import collections

import numpy as np
import torch

from .engine import Engine


class EnginePipeline:
    def __init__(self, state_dict, device=0, depth=4, per_slot=2, **engine_kw):
        if depth > 1:           # engines that run beside each other: throughput schedule (include/nuhtc_hip.h)
            from . import hip
            engine_kw.setdefault('schedule', hip.SCHED_THROUGHPUT)
        self.engines = [Engine(state_dict, device=device, **engine_kw) for _ in range(depth)]
        self.device = self.engines[0].device
        # every engine runs on the stream it created next to its side streams (nuhtc_stream: three different pipes of the command
        # processor by construction, include/nuhtc_hip.h)
        self.streams = [e.stream for e in self.engines]
        self.pending = collections.deque()      # (slot, B, event, user tag, keep-alive, export turn)
        self.next = 0
        self.last_turn = None                   # export buffer of the batch collect() returned last (Engine.export_read(turn))
        # Batches submitted with export=True leave the device through the engine's pinned host buffers (Engine.EXPORT_BUFFERS = 3), used
        # in turn: a slot may hold `per_slot` = 2 such batches, the second queued behind the first on the slot's stream, and the third
        # buffer is what lets the host resubmit to the slot BEFORE it has unpacked the batch it just collected.  The streams then never run dry while
        # the host waits for, unpacks and resubmits a batch -- with one batch per slot the batches in flight finish together and the
        # GPU idles until the host has refilled the slots (1.51k tiles/s against 1.9k for the same four engines fed without pause).
        # Batches whose results are read from the engine's own tensors (export=False) stay one per slot.
        self.per_slot = max(1, min(2, int(per_slot)))
        if Engine.EXPORT_BUFFERS < self.per_slot + 1:
            raise RuntimeError('EnginePipeline needs Engine.EXPORT_BUFFERS >= per_slot + 1 (a collected batch is read while the slot already holds per_slot new ones)')

    @property
    def depth(self):
        return len(self.engines)

    def _held(self, slot):
        return [p for p in self.pending if p[0] == slot]

    def submit(self, tiles, channel_mode, tag=None, export=False):
        """Enqueue one batch (host ndarray / tensor, or device tensor) on the next slot; returns the slot's engine.
        Raises when that slot cannot take another batch (collect() first: see full()).  export=True also enqueues
        Engine.export_async."""
        slot = self.next
        held = self._held(slot)
        if held and (not export or len(held) >= self.per_slot or any(p[5] is None for p in held)):
            raise RuntimeError('pipeline slot still holds an uncollected batch: call collect() first')
        self.next = (slot + 1) % self.depth
        eng, st = self.engines[slot], self.streams[slot]
        # A device batch was produced on the caller's stream: order the slot's stream behind it.  Host batches need no such wait, and
        # it is not free: the event is recorded on the caller's stream, which with the runtime's four hardware queues shares a queue
        # with one of the engines -- the marker then sits behind that engine's queued batches and the new batch waits for them
        # (the slide loop ran at 1.51k instead of 1.8k tiles/s).
        if isinstance(tiles, torch.Tensor) and tiles.is_cuda:
            st.wait_stream(torch.cuda.current_stream(self.device))
        # the host source of an asynchronous H2D copy must outlive the copy: keep it (and the device batch) referenced
        # until the batch is collected, callers may hand in temporaries
        src = torch.from_numpy(np.ascontiguousarray(tiles)) if isinstance(tiles, np.ndarray) else tiles
        turn = None
        with torch.cuda.stream(st):
            dev = eng.to_device(src)
            B = eng.infer_async(dev, channel_mode)
            if export:                      # contours + gather of the kept detections into pinned host buffers, still asynchronous
                turn = eng.export_async(B)
            ev = torch.cuda.Event()
            ev.record(st)
        self.pending.append((slot, B, ev, tag, (dev, src), turn))
        return eng

    def full(self, export=False):
        """No slot free for the next submit() (of that kind)."""
        held = self._held(self.next)
        if not held:
            return False
        return not export or len(held) >= self.per_slot or any(p[5] is None for p in held)

    def close(self):
        for e in self.engines:
            e.close()

    def collect(self):
        """Oldest submitted batch: waits for it and returns (engine, B, stream, tag); read the engine's output tensors on
        `stream` (torch.cuda.stream(stream)) before submitting to that slot again -- or, for a batch submitted with export=True,
        read `engine.export_read(pipeline.last_turn)`: the engine's own tensors may belong to the slot's next batch by then."""
        slot, B, ev, tag, _, turn = self.pending.popleft()
        ev.synchronize()
        self.last_turn = turn
        self.engines[slot]._read_turn = turn      # export_read() of the returned engine reads THIS batch, not the slot's next one
        if turn is None:
            with torch.cuda.stream(self.streams[slot]):      # (the flags are read on the slot's own, by now idle, stream)
                self.engines[slot].check()
        return self.engines[slot], B, self.streams[slot], tag

    def drain(self):
        while self.pending:
            yield self.collect()
