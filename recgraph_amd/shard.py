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
