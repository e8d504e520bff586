"""Read sharding across the GPUs of one node and the gather of GAF text to rank 0.

Reads are independent units (the reference loops over them one by one, main.rs:56,174,257,297), so the
batch partitions into contiguous blocks, one per rank, with no exchange inside the DP.  The only
collective is the gather of the formatted records to rank 0 (RCCL over xGMI on the GPU box;
the same code runs over gloo in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, order is preserved."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def even_tiles(n_reads, max_tile, ramp=0):
    """Tile sizes for `n_reads` reads of one rank: as few tiles as `max_tile` allows, all of (almost) the same size — a
    rank's share of a fixed read set is not a multiple of the tile size, and whole tiles would leave the ranks with
    different numbers of launches (102 400 reads on 8 GPUs: 4/3/3/3/3/3/3/3 tiles of 4096 = a 6.25x ceiling).
    `ramp` > 0: the first `ramp` tiles are cut in two (a short one first), so that the handles of a stream start out of
    phase: the sweeps of one tile then run beside the small kernels of another from the first tile on."""
    if n_reads <= 0:
        return []
    nt = -(-n_reads // max_tile)
    sizes = [n_reads * (k + 1) // nt - n_reads * k // nt for k in range(nt)]
    out = []
    for k, s in enumerate(sizes):
        if k < ramp and s >= 4:
            a = s * (k + 1) // (ramp + 2)          # 1/3, 2/4 ... of the tile first
            out += [max(1, a), s - max(1, a)]
        else:
            out.append(s)
    return out


def gather_parts(data: bytes, rank, world, device="cpu", dst=0):
    """Gather of variable-length byte strings to `dst`: all_gather of the sizes + gather of the padded payload.  Returns
    the per-rank payloads as uint8 CPU tensors (rank order) and their sizes on dst, (None, sizes) elsewhere."""
    n = len(data)
    if world == 1:
        return [torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy())], [n]
    ln = torch.tensor([n], dtype=torch.int64, device=device)
    lens = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(lens, ln)
    sizes = [int(x.item()) for x in lens]
    mx = max(1, max(sizes))
    pad = torch.zeros(mx, dtype=torch.uint8)
    if n:
        pad[:n] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy())
    pad = pad.to(device)
    try:
        outs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, outs, dst=dst)
    except (RuntimeError, NotImplementedError):
        # a backend without gather: every rank receives every part (the same bytes arrive on `dst`)
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad)
    if rank != dst:
        return None, sizes
    return [o[:s].cpu() for o, s in zip(outs, sizes)], sizes


def gather_text(text: bytes, rank, world, device="cpu", dst=0):
    """`gather_parts` as bytes objects: the list (rank order) on dst, None elsewhere."""
    parts, _ = gather_parts(text, rank, world, device, dst)
    if parts is None:
        return None
    return [bytes(p.numpy().tobytes()) for p in parts]


class FixedGather:
    """Gather of one byte payload per step to `dst` with NO host synchronisation on the other ranks.

    `gather_parts` asks the device twice per step on every rank (`.item()` on the gathered sizes, `.cpu()` on the
    payloads): on a GPU box each of those spins or sleeps a host thread until the slowest rank arrives.  Here every step
    is ONE collective of a fixed-capacity buffer — an 8-byte length header followed by the payload, padded to `cap` bytes —
    issued with ``async_op=True`` from a ring of `depth` page-locked staging buffers:

    * a rank other than `dst` copies its text into the staging buffer, enqueues the host-to-device copy and the gather and
      returns; it touches the device again only to reuse a ring slot `depth` steps later (an event query; it waits only if
      the destination has fallen that far behind);
    * `dst` reads the lengths out of the headers after its own device-to-host copy of the gathered buffers, one event wait
      per step (it needs the bytes on the host; under hipDeviceScheduleBlockingSync that wait sleeps).

    A payload longer than `cap` cannot be announced to the other ranks without a second collective, so it is an error on the
    rank that holds it (the caller sizes `cap` from what it knows about its records: bench.py takes 3 x read length + 600
    bytes per read).  Over gloo (CPU tests) the same code runs with CPU tensors; `wait()` blocks there."""

    HDR = 8

    def __init__(self, rank, world, device, cap, dst=0, depth=4):
        self.rank, self.world, self.device, self.cap, self.dst, self.depth = rank, world, device, (int(cap) + 7) // 8 * 8, dst, depth      # (8-byte rows: the headers are read as int64)
        self.cuda = str(device) != "cpu"
        n = self.HDR + self.cap
        self.host = [torch.zeros(n, dtype=torch.uint8).pin_memory() if self.cuda else torch.zeros(n, dtype=torch.uint8) for _ in range(depth)]
        self.dev = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(depth)] if self.cuda else self.host
        self.outs = None
        self.land = None
        if rank == dst:
            self.outs = [[torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(world)] for _ in range(depth)]
            if self.cuda:
                self.land = [torch.zeros((world, n), dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.work = [None] * depth            # the slot's collective
        self.event = [None] * depth           # cuda: recorded behind the slot's last device operation
        self.step = 0
        self.results = []                     # dst: per step the list of per-rank payloads (uint8 CPU tensors)
        self.sizes = []
        self.pending = []                     # dst: slots whose bytes have not been taken to `results` yet
        self.host_waits = 0                   # times a slot had to be waited for before reuse (back-pressure)

    def _release(self, k):
        """Slot k may be overwritten once its collective (and, on dst, the copy behind it) is complete."""
        if self.work[k] is None:
            return
        if self.rank == self.dst:
            self._collect(k)
            return
        if self.cuda:
            if not self.event[k].query():
                self.host_waits += 1
                self.event[k].synchronize()
        else:
            self.work[k].wait()
        self.work[k] = None

    def _collect(self, k):
        """dst: take slot k's gathered bytes to the host and file them as the next step's result."""
        if self.work[k] is None:
            return
        if self.cuda:
            self.event[k].synchronize()
            rows = self.land[k]
        else:
            self.work[k].wait()
            rows = torch.stack(self.outs[k])
        parts, sizes = [], []
        for r in range(self.world):
            n = int(rows[r, :self.HDR].view(torch.int64)[0])
            if n < 0 or n > self.cap:
                raise RuntimeError("FixedGather: rank %d announced %d bytes for a capacity of %d" % (r, n, self.cap))
            parts.append(rows[r, self.HDR:self.HDR + n].clone())
            sizes.append(n)
        self.results.append(parts)
        self.sizes.append(sizes)
        self.work[k] = None

    def submit(self, data: bytes):
        n = len(data)
        if n > self.cap:
            raise ValueError("FixedGather: %d bytes do not fit the capacity of %d agreed on by the ranks" % (n, self.cap))
        k = self.step % self.depth
        self.step += 1
        if self.rank == self.dst:
            # slots are used round-robin and results are filed in step order: when every slot is in flight the oldest one is
            # this slot's previous use
            if len(self.pending) == self.depth:
                assert self.pending[0] == k
                self._collect(self.pending.pop(0))
        else:
            self._release(k)
        h = self.host[k]
        h[:self.HDR].view(torch.int64)[0] = n
        if n:
            h[self.HDR:self.HDR + n] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        if self.cuda:
            self.dev[k].copy_(h, non_blocking=True)
        outs = self.outs[k] if self.rank == self.dst else None
        self.work[k] = dist.gather(self.dev[k], outs, dst=self.dst, async_op=True)
        if self.cuda:
            self.work[k].wait()                   # (NCCL: the CURRENT STREAM waits for the collective; the host does not)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.land[k][r].copy_(outs[r], non_blocking=True)
            self.event[k] = torch.cuda.Event()
            self.event[k].record()
        if self.rank == self.dst:
            self.pending.append(k)

    def finish(self):
        """Completes every outstanding step.  dst: returns (results, sizes) — per step the per-rank payloads; the other ranks
        (None, None) after their last collectives have left the device."""
        if self.rank == self.dst:
            while self.pending:
                self._collect(self.pending.pop(0))
            return self.results, self.sizes
        for k in range(self.depth):
            self._release(k)
        return None, None
