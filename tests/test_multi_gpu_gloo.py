"""The N > 1 path on CPU: two processes (gloo), each renders its image tiles, rank 0 gathers. The renderer behind the
C ABI here is the CPU restatement (tests may use it); partition, gather and the rank-count independence of the result
are the product's host logic in raytracer_amd/multi_gpu.py, the same code bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import capi, multi_gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, scene_file, tile, result_path):
    from tests.oracle_binding import oracle_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hs = capi.HostScene.load(scene_file).build()
        o = hs.desc.contents.options
        scene = oracle_scene(hs.desc, threads=2)
        out = torch.zeros((o.height, o.width, 4), dtype=torch.float32)
        st = multi_gpu.render_partition(scene, out, rank, world, tile=tile, seed=7)
        own = multi_gpu.tile_owner(o.width, o.height, tile, world) == rank
        mine = out.numpy().copy()
        assert np.all(mine[~own] == 0), "pixels outside this rank's tiles must be exact zeros"
        assert st.cameraSamples == int(own.sum()) * o.numRays
        multi_gpu.TileGather(o.width, o.height, tile, rank, world, out.device).gather(out)
        samples = torch.tensor([float(st.cameraSamples)], dtype=torch.float64)
        dist.all_reduce(samples)
        if rank == 0:
            np.savez(result_path, frame=out.numpy(), samples=samples.numpy())
        scene.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("scene_name,tile", [("mesh1k_gi", (32, 32)), ("spheres_txt", (16, 8))])
def test_two_ranks_equal_one(tmp_path, scene_path, scene_name, tile):
    from tests.oracle_binding import oracle_scene
    scene_file = scene_path(scene_name)
    result = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), scene_file, tile, result), nprocs=2, join=True)
    got = np.load(result)
    hs = capi.HostScene.load(scene_file).build()
    o = hs.desc.contents.options
    whole = oracle_scene(hs.desc, threads=4)
    ref, st = whole.render(seed=7)
    whole.close()
    ref = np.asarray(ref, dtype=np.float32).reshape(o.height, o.width, 4)
    assert int(got["samples"][0]) == st.cameraSamples
    assert np.array_equal(got["frame"], ref), "the gathered two-rank frame must equal the one-rank frame bit for bit"


def _gather_worker(rank, world, port, width, height, tile, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank holds "its" pixels of a known frame and zeros elsewhere, as rt_render leaves them
        frame = torch.arange(height * width * 4, dtype=torch.float32).reshape(height, width, 4) + 1.0
        own = torch.from_numpy(multi_gpu.tile_owner(width, height, tile, world) == rank)
        out = torch.where(own[..., None], frame, torch.zeros_like(frame))
        g = multi_gpu.TileGather(width, height, tile, rank, world, out.device)
        for _ in range(2):  # the gatherer keeps its buffers across frames
            g.gather(out)
        if rank == 0:
            np.save(result_path, out.numpy())
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,tile", [(3, 70, 50, (32, 32)), (4, 33, 17, (8, 4)), (2, 5, 3, (32, 32))])
def test_tile_gather_ragged_sizes(tmp_path, world, width, height, tile):
    """Tiles that do not divide the image, more ranks than tiles, three and four ranks: rank 0 ends up with every pixel."""
    result = str(tmp_path / "frame.npy")
    mp.spawn(_gather_worker, args=(world, _free_port(), width, height, tile, result), nprocs=world, join=True)
    want = np.arange(height * width * 4, dtype=np.float32).reshape(height, width, 4) + 1.0
    assert np.array_equal(np.load(result), want)


def test_tile_owner_matches_library_rule():
    own = multi_gpu.tile_owner(70, 50, (32, 32), 3)
    assert own.shape == (50, 70)
    assert own[0, 0] == 0 and own[0, 32] == 1 and own[0, 64] == 2 and own[32, 0] == 0 and own[32, 32] == 1
    assert np.all(multi_gpu.tile_owner(9, 9, (4, 4), 1) == 0)


def _shared_scene_worker(rank, world, port, scene_file, shared_dir, tile, result_path):
    from raytracer_amd import desc_io
    from tests.oracle_binding import oracle_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # what bench.py does for N > 1: rank 0 parses the files and builds the BVH once, the others map the flattened arrays
        if rank == 0:
            hs = capi.HostScene.load(scene_file).build()
            desc_io.save_desc(hs.desc, shared_dir)
        dist.barrier()
        if rank != 0:
            hs = desc_io.load_desc(shared_dir)
        o = hs.desc.contents.options
        scene = oracle_scene(hs.desc, threads=2)
        out = torch.zeros((o.height, o.width, 4), dtype=torch.float32)
        multi_gpu.render_partition(scene, out, rank, world, tile=tile, seed=7)
        multi_gpu.TileGather(o.width, o.height, tile, rank, world, out.device).gather(out)
        if rank == 0:
            np.save(result_path, out.numpy())
        scene.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_scene_built_once_and_mapped_by_the_other_ranks(tmp_path, scene_path):
    """Three ranks, one scene build: the ranks that load the flattened arrays from files render the same pixels as ranks
    that would have parsed the scene themselves (textured scene: texels and texture coordinates travel too)."""
    from tests.oracle_binding import oracle_scene
    scene_file = scene_path("readme_small")
    result = str(tmp_path / "gathered.npy")
    mp.spawn(_shared_scene_worker, args=(3, _free_port(), scene_file, str(tmp_path / "flat"), (16, 16), result), nprocs=3, join=True)
    hs = capi.HostScene.load(scene_file).build()
    whole = oracle_scene(hs.desc, threads=4)
    ref, _ = whole.render(seed=7)
    whole.close()
    assert np.array_equal(np.load(result).view(np.uint32), np.asarray(ref, dtype=np.float32).view(np.uint32))
