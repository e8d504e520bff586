"""Read sharding across the GPUs of one node and the final gather of GAF text.

Reads are independent units (the reference loops over them one by one, main.rs:56,174,257,297), so the
batch partitions into contiguous blocks, one per rank, with no exchange inside the DP.  The only
collective is the gather of the formatted records to rank 0 at the end (RCCL over xGMI on the GPU box;
the same code runs over gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, order is preserved."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_text(text: bytes, rank, world, device="cpu", dst=0):
    """Gather variable-length byte strings to `dst`.  Returns the list (rank order) on dst, None elsewhere."""
    if world == 1:
        return [text]
    t = torch.frombuffer(bytearray(text) if text else bytearray(1), dtype=torch.uint8)
    n = len(text)
    t = t[:n].to(device) if n else torch.zeros(0, dtype=torch.uint8, device=device)
    ln = torch.tensor([n], device=device, dtype=torch.int64)
    lens = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(lens, ln)
    sizes = [int(x.item()) for x in lens]
    mx = max(1, max(sizes))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:n] = t
    try:
        outs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, outs, dst=dst)
    except (RuntimeError, NotImplementedError):
        # a backend without gather: every rank receives every part (the same bytes arrive on `dst`)
        outs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad)
    if rank != dst:
        return None
    return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]
