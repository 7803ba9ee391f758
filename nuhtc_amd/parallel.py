"""Multi-GPU layout of the WSI path: tiles are independent, so the tile list is sharded in contiguous
blocks across one process per GPU with no data-path collective; the only exchange is one variable-length gather of
per-detection records before the cross-tile merge (pattern of mmdet/apis/test.py:161-191 `collect_results_gpu`:
all_gather of the counts, then of buffers padded to the maximum) over RCCL/xGMI (backend "nccl") or gloo (CPU tests)."""
import os

import torch
import torch.distributed as dist


def force_collective():
    """NUHTC_FORCE_COLLECTIVE=1: a job of ONE rank still forms its process group and sends the exchange through the collectives
    (two all_gathers over a communicator of size 1) instead of short-circuiting them -- the way to run the RCCL branch of
    `gather_blobs` (device buffers in, device buffers out) on a one-GPU box.  Results are byte-equal to the short-circuit."""
    return os.environ.get('NUHTC_FORCE_COLLECTIVE') == '1'


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n, script, argv, relay=None):
    """Run `script argv` as `n` ranks (one per GPU) under torch.distributed.run as a CHILD process -- never an exec, and before anything in
    the caller has touched the GPU -- and return the job's exit code.  `relay(line)` decides where each output line goes (default: stdout).
    The reference's launcher is the same command typed by hand (tools/dist_test.sh -> torch.distributed.launch)."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(script)] + list(argv)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:
        (relay(line) if relay else sys.stdout).write(line)
        sys.stdout.flush()
    return p.wait()


def env_ranks():
    """(rank, local_rank, world) of this process from the launcher's environment, without forming the process group."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('NUHTC_ONE_DEVICE') == '1':       # test hook: several ranks on a one-GPU box (with NUHTC_DIST_BACKEND=gloo)
        local_rank = 0
    return rank, local_rank, world


def job_token():
    """A name all ranks of ONE launch share and no other launch does: the rendezvous port plus the launcher's pid (the ranks of a
    torch.distributed.run job are children of one agent process)."""
    return f"{os.environ.get('MASTER_PORT', '0')}.{os.getppid()}"


def host_phase_done(directory, rank, world, poll_s=0.2):
    """Rank 0 has finished a host-only phase of unbounded length (tissue segmentation and patching of a whole folder of slides): it drops
    a marker file, the other ranks poll for it.  Deliberately NOT a collective: a rank parked in an RCCL barrier is aborted by the
    communicator's watchdog after its timeout (10 minutes by default), and a folder of real slides takes longer than that -- so this runs
    BEFORE the process group exists.  If rank 0 dies the launcher ends the other ranks."""
    import time
    if world <= 1:
        return
    marker = os.path.join(directory, f'.host_phase_done.{job_token()}')
    if rank == 0:
        with open(marker, 'w') as f:
            f.write('done\n')
        return
    while not os.path.exists(marker):
        time.sleep(poll_s)


def host_phase_cleanup(directory, rank, world):
    """Remove rank 0's marker (call after a collective that every rank has passed)."""
    if world > 1 and rank == 0:
        try:
            os.remove(os.path.join(directory, f'.host_phase_done.{job_token()}'))
        except OSError:
            pass


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    rank, local_rank, world = env_ranks()
    if world == 1 and force_collective() and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if (world > 1 or force_collective()) and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = backend or os.environ.get('NUHTC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of `n` tiles for `rank` (keeps spatial locality for the merge); sizes differ by <= 1."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(rec, group=None):
    """rec: (n_i, F) tensor of this rank's records (any n_i >= 0) -> list over ranks of (n_r, F) tensors (on every rank).
    Two collectives: counts (1 int64 per rank), then the payload padded to the maximum count."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [rec]
    world = dist.get_world_size(group)
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:c] for b, c in zip(bufs, counts)]


def gather_blobs(parts, group=None):
    """The one exchange of the WSI path (north star: a single all-gather of the per-tile detections): `parts` is this rank's
    list of tensors -- any dtypes, any leading lengths (0 allowed), trailing shapes equal on all ranks -- and every rank gets
    back, per rank, the list of that rank's tensors.  All parts travel as ONE byte buffer: an all_gather of the small header
    (leading lengths; this is the "counts" exchange of mmdet/apis/test.py:161-191) followed by one all_gather of the buffers
    padded to the longest.
    Every rank receives every rank's records although only rank 0 merges: that is the collective the north star names, and at ~0.25 KB per
    detection (45 MB per rank for a 10 000-tile slide shard) the ring all-gather over xGMI is milliseconds; a gather to rank 0 alone would
    save the other ranks' receive buffers, nothing on the critical path."""
    parts = [p.contiguous() for p in parts]
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective()):
        return [parts]
    world = dist.get_world_size(group)
    if dist.get_backend(group) == 'gloo':       # gloo moves host memory: hand it host tensors
        parts = [p.cpu() for p in parts]
    dev = parts[0].device
    head = torch.tensor([p.shape[0] for p in parts], dtype=torch.int64, device=dev)
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    heads = [h.cpu().tolist() for h in heads]
    row_bytes = [p.element_size() * int(torch.tensor(p.shape[1:]).prod()) if p.dim() > 1 else p.element_size() for p in parts]

    def seg_bytes(lengths):            # every segment starts on an 8-byte boundary
        return [-(-n * rb // 8) * 8 for n, rb in zip(lengths, row_bytes)]
    total = max(max(sum(seg_bytes(h)) for h in heads), 8)
    buf = torch.zeros(total, dtype=torch.uint8, device=dev)
    off = 0
    for p, sb in zip(parts, seg_bytes(head.cpu().tolist())):
        nb = p.numel() * p.element_size()
        if nb:
            buf[off:off + nb] = p.reshape(-1).view(torch.uint8)
        off += sb
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = []
    for b, h in zip(bufs, heads):
        off, got = 0, []
        for p, n, rb, sb in zip(parts, h, row_bytes, seg_bytes(h)):
            seg = b[off:off + n * rb].view(p.dtype) if n else torch.zeros(0, dtype=p.dtype, device=dev)
            got.append(seg.reshape((n,) + tuple(p.shape[1:])))
            off += sb
        out.append(got)
    return out
