"""One frame over several GPUs of a node: one process per GPU, each renders the image tiles it owns, rank 0 ends up with
the whole frame.

The reference splits a frame over worker threads by pixel index (src/scene/raytracer.cpp:258, 290-299); here the unit is
a tile of pixels and the workers are GPUs. The scene and its BVH are replicated per GPU (they are read-only and small
next to 288 GB of HBM), a pixel's random stream depends only on (seed, pixel, sample), and pixels outside a rank's tiles
come back as exact zeros, so the frame does not depend on the number of ranks. The only exchange is one gather per frame
of each rank's own pixels (width*height*16/world bytes per rank) to rank 0: RCCL over xGMI with the "nccl" backend, TCP
loopback with "gloo".
"""
import numpy as np


def tile_owner(width, height, tile, part_count):
    """[height, width] int array: the rank that renders each pixel (row-major tile index modulo part_count), the rule of
    RtRenderParams.partIndex / partCount in include/rt_amd.h."""
    tw, th = tile
    tiles_x = (width + tw - 1) // tw
    ys, xs = np.mgrid[0:height, 0:width]
    return ((ys // th) * tiles_x + xs // tw) % max(1, part_count)


def render_partition(scene, out, rank, world, tile=(32, 32), seed=56, samples_per_pass=0, **kw):
    """Renders this rank's tiles of the frame into `out` (a float32 [H, W, 4] torch tensor on the scene's device, or on
    the host for a host-side renderer); the rest of `out` is zero afterwards. Returns the RtStats of the call."""
    if out.dtype.itemsize != 4 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor")
    _, st = scene.render(seed=seed, part_index=rank, part_count=world, tile=tile, samples_per_pass=samples_per_pass,
                         out_ptr=out.data_ptr(), **kw)
    return st


class TileGather:
    """Moves every rank's own pixels to rank 0: each rank packs the pixels of its tiles (1/world of the frame, 16 B per
    pixel) into one contiguous buffer, one gather brings the buffers to rank 0 (point-to-point transfers into rank 0,
    no ring), rank 0 writes them to their places in its frame. Nothing is added, so the gathered frame holds exactly
    the bits each rank rendered."""

    def __init__(self, width, height, tile, rank, world, device):
        import torch
        self.rank, self.world = rank, world
        owner = tile_owner(width, height, tile, world).reshape(-1)
        counts = np.bincount(owner, minlength=world)
        self.max_pixels = int(counts.max())
        ranks = range(world) if rank == 0 else [rank]
        self.index = {r: torch.from_numpy(np.flatnonzero(owner == r)).to(device) for r in ranks}
        self.packed = torch.zeros((self.max_pixels, 4), dtype=torch.float32, device=device)
        self.inbox = ([torch.zeros_like(self.packed) for _ in range(world)] if rank == 0 and world > 1 else None)

    def gather(self, out, dst=0):
        if self.world <= 1:
            return out
        import torch.distributed as dist
        if dst != 0:
            raise ValueError("the frame is gathered on rank 0")
        flat = out.view(-1, 4)
        mine = self.index[self.rank]
        self.packed[: mine.numel()] = flat[mine]
        dist.gather(self.packed, self.inbox, dst=0)
        if self.rank == 0:
            for r in range(1, self.world):
                idx = self.index[r]
                flat[idx] = self.inbox[r][: idx.numel()]
        if out.is_cuda:
            # the renderer writes `out` from a stream of its own: the frame must be complete before the next one starts
            import torch
            torch.cuda.current_stream(out.device).synchronize()
        return out


def render_frame(scene, out, rank, world, gatherer=None, tile=(32, 32), **kw):
    """One frame: this rank's tiles, then the gather. `gatherer` is the TileGather of this (size, tile, world); pass it
    to keep its buffers across frames."""
    st = render_partition(scene, out, rank, world, tile=tile, **kw)
    if world > 1:
        if gatherer is None:
            gatherer = TileGather(out.shape[1], out.shape[0], tile, rank, world, out.device)
        gatherer.gather(out)
    return st
