"""Path-index sharding across the GPUs of one node and the ordered framebuffer exchange.

The reference has no distributed anything; this is new work shaped for MI355X's fully connected xGMI mesh:
  * shapes shard by CONTIGUOUS index range (painter's order is preserved inside a rank and across ranks),
  * every rank renders its shard into a private full-size premultiplied RGBA8 layer,
  * tile-sliced all-to-all: the frame is cut into `world` row slabs; rank r receives slab r of every layer
    (grouped send/recv = every GPU drives all of its xGMI links at once, 1/world of a layer per link — a ring would be
    per-link bound and world-1 steps deep),
  * rank r composites its `world` slabs in rank order with premultiplied "over" (associative, NOT commutative),
  * the finished slabs are gathered to rank 0.
torch.distributed is plumbing here: backend "nccl" is RCCL on ROCm; the CPU tests drive the same code over "gloo".
"""
import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, order-preserving split of [0, n_items) into `world` ranges whose sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def slab_rows(height: int, world: int):
    """Row ranges of the `world` slabs, aligned to the rasterizer's 16-pixel tiles."""
    tiles = (height + 15) // 16
    out = []
    for r in range(world):
        t0, t1 = shard_range(tiles, r, world)
        out.append((min(t0 * 16, height), min(t1 * 16, height)))
    return out


def _transport_needs_host_staging(tensor):
    """RCCL moves device memory directly. The gloo backend (used to validate the multi-rank flow on a single GPU) only sends host
    memory, so device tensors are staged through the host for it."""
    import torch.distributed as dist
    return tensor.is_cuda and dist.get_backend() == "gloo"


def exchange_layers(layer, rank: int, world: int, group=None):
    """layer: uint8 tensor [H, W, 4] (this rank's premultiplied layer). Returns (received [world, h_r, W, 4], (row0, row1))
    = slab `rank` of every rank's layer, in rank order."""
    import torch
    import torch.distributed as dist
    if _transport_needs_host_staging(layer):
        received, rows = exchange_layers(layer.cpu(), rank, world, group)
        return received.to(layer.device), rows
    h = layer.shape[0]
    rows = slab_rows(h, world)
    r0, r1 = rows[rank]
    received = torch.empty((world, r1 - r0) + tuple(layer.shape[1:]), dtype=layer.dtype, device=layer.device)
    ops = []
    for peer in range(world):
        p0, p1 = rows[peer]
        if peer == rank:
            received[rank].copy_(layer[r0:r1])
            continue
        if p1 > p0:
            ops.append(dist.P2POp(dist.isend, layer[p0:p1].contiguous(), peer, group))
        if r1 > r0:
            ops.append(dist.P2POp(dist.irecv, received[peer], peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return received, (r0, r1)


def gather_slabs(slab, rank: int, world: int, height: int, group=None):
    """Collects the composited slabs on rank 0 -> full [H, W, 4] image (None on the other ranks)."""
    import torch
    import torch.distributed as dist
    if _transport_needs_host_staging(slab):
        image = gather_slabs(slab.cpu(), rank, world, height, group)
        return image.to(slab.device) if image is not None else None
    rows = slab_rows(height, world)
    if rank == 0:
        image = torch.empty((height,) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
        image[rows[0][0]:rows[0][1]].copy_(slab)
        ops = []
        for peer in range(1, world):
            p0, p1 = rows[peer]
            if p1 > p0:
                ops.append(dist.P2POp(dist.irecv, image[p0:p1], peer, group))  # a row range of a contiguous image is contiguous: received in place
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return image
    if slab.shape[0] > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, slab.contiguous(), 0, group)]):
            req.wait()
    return None


def composite_over_reference(layers: np.ndarray) -> np.ndarray:
    """numpy statement of the ordered "over" (dst = src + dst * (1 - src.a), layer 0 at the bottom) with the same f32 operation
    order and RGBA8 rounding as k_composite / k_composite_tiles — used by the gloo tests, where no GPU exists, and as the checker of the
    C-ABI exchange. uint8 layers are RGBA8 unorm, float16 layers RGBA16F (CRH_FORMAT_RGBA16F). Not a product path."""
    acc = np.zeros(layers.shape[1:], dtype=np.float32)
    for layer in layers:
        src = layer.astype(np.float32) * np.float32(1.0 / 255.0) if layer.dtype == np.uint8 else layer.astype(np.float32)
        k = (np.float32(1.0) - src[..., 3:4]).astype(np.float32)
        acc = (src + acc * k).astype(np.float32)
    acc = np.clip(acc, 0.0, 1.0)
    return (acc * np.float32(255.0) + np.float32(0.5)).astype(np.int32).astype(np.uint8)
