"""One frame over several GPUs of a node: one process per GPU, each renders the image tiles it owns, rank 0 ends up with
the whole frame.

The reference splits a frame over worker threads by pixel index (src/scene/raytracer.cpp:258, 290-299); here the unit is
a tile of pixels and the workers are GPUs. The scene and its BVH are replicated per GPU (they are read-only and small
next to 288 GB of HBM), a pixel's random stream depends only on (seed, pixel, sample), and pixels outside a rank's tiles
come back as exact zeros, so the partial frames combine by addition into a frame that does not depend on the number of
ranks. The only exchange is that one reduce of width*height*4 floats per frame (RCCL over xGMI with the "nccl" backend,
plain TCP loopback with "gloo").
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


def gather_frame(out, world, dst=0):
    """Combines the ranks' partial frames on rank `dst` (in place). No-op for a single rank."""
    if world > 1:
        import torch.distributed as dist
        dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    return out


def render_frame(scene, out, rank, world, **kw):
    st = render_partition(scene, out, rank, world, **kw)
    gather_frame(out, world)
    return st
