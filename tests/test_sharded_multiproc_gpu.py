"""The REAL GPU shard engines in several processes: world_size 2 and 3 `gloo` process groups whose
ranks all sit on cuda:0 drive GpuShardEngine / GpuFlatShard / GpuAccumShard through the product drivers of
richdem_amd/sharded.py -- fill, flat-resolved directions, accumulation of one raster cut into row blocks -- with the
collectives staged through host memory (gloo moves host memory only; on a multi-GPU node the same drivers run over RCCL).
The rows every rank returns must equal the CPU oracle on the whole raster: tiling invariance, the reference's own test idea
(programs/parallel_priority_flood/test.py:44-118, parallel_d8_accum/test_small.sh).
Also: the one-exchange device protocol of the fill on RCCL at world_size 1 with a payload capacity that is too small (the
solve refuses on the device, the exchange is repeated with the exact size)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

H, W = 3000, 4000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dem():
    from richdem_amd.synth import fractal_dem

    return fractal_dem(W, H, seed=2)     # the S2 generator's terrain, 12 M cells


def _worker(rank, world, port, outdir, mode):
    import torch
    import torch.distributed as dist

    from richdem_amd.sharded import (GpuAccumShard, GpuFlatShard, GpuShardEngine, d8_flow_accum_sharded,
                                     d8_flow_directions_sharded, fill_depressions_sharded, row_split)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dem = np.load(os.path.join(outdir, "dem.npy"))
    r0, r1 = row_split(dem.shape[0], world)[rank]
    block = torch.from_numpy(np.ascontiguousarray(dem[r0:r1])).cuda()
    if mode == "small_payload":
        os.environ["RDGPU_SHARD_EDGE_CAP"] = "5"       # the solve refuses on the device, the exact-size exchange follows
    if mode == "host_solve":   # export to host, all-gather, rdgpu_fill_graph_solve on the host, finish (r02's exchange)
        fill_depressions_sharded(block, engine=GpuShardEngine(), comm_device="cpu")
    else:                      # ONE all-gather of the fixed-capacity payload (staged through the host by gloo), GPU solve
        fill_depressions_sharded(block, engine=GpuShardEngine())
    dirs = d8_flow_directions_sharded(block, -9999.0, flats=True)                # GpuFlatShard, repeated cut-row exchange
    assert isinstance(GpuFlatShard(), GpuFlatShard)
    area = torch.empty(block.shape, dtype=torch.float64, device="cuda")
    exchanges = d8_flow_accum_sharded(dirs, area, shard=GpuAccumShard())         # one exchange
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"fill{rank}.npy"), block.cpu().numpy())
    np.save(os.path.join(outdir, f"dirs{rank}.npy"), dirs.cpu().numpy())
    np.save(os.path.join(outdir, f"area{rank}.npy"), area.cpu().numpy())
    np.save(os.path.join(outdir, f"ex{rank}.npy"), np.array([exchanges]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "one_exchange"), (3, "one_exchange"), (3, "host_solve"), (2, "small_payload")])
def test_gpu_shard_engines_in_separate_processes(orc, tmp_path, world, mode):
    import torch.multiprocessing as mp

    dem = _dem()
    np.save(tmp_path / "dem.npy", dem)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    cat = lambda name: np.concatenate([np.load(tmp_path / f"{name}{r}.npy") for r in range(world)], axis=0)   # noqa: E731
    filled = orc.port.fill(dem, 8)
    assert np.array_equal(cat("fill"), filled)
    edirs = orc.port.flat_resolution(filled, np.float32(-9999))
    assert np.array_equal(cat("dirs"), edirs)
    assert np.array_equal(cat("area"), orc.port.d8_flow_accum(edirs, 255, np.float64))
    assert all(int(np.load(tmp_path / f"ex{r}.npy")[0]) == 1 for r in range(world))


def _rccl_worker(rank, world, port, outdir, cap):
    import torch
    import torch.distributed as dist

    from richdem_amd.sharded import fill_depressions_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if cap:
        os.environ["RDGPU_SHARD_EDGE_CAP"] = str(cap)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dem = np.load(os.path.join(outdir, "dem.npy"))
    block = torch.from_numpy(dem).cuda()
    fill_depressions_sharded(block)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "rccl_fill.npy"), block.cpu().numpy())
    dist.destroy_process_group()


def test_fill_blocks_through_a_payload_that_is_too_small(rd, orc, monkeypatch):
    """The fixed-capacity payload of the one-exchange fill: with room for 7 edge triples the device solve refuses
    (RDGPU_ERR_CAPACITY) and the exact-size exchange follows -- exercised through the single-process block driver's
    building blocks, which share graph_solve_dev with the RCCL driver."""
    import torch

    from richdem_amd import RdgpuError
    from richdem_amd.sharded import GpuShardEngine, graph_solve_dev, row_split

    dem = _dem()[:900, :1100].copy()
    t = torch.from_numpy(dem).cuda()
    blocks = [t[a:b] for a, b in row_split(t.shape[0], 3)]
    engs = [GpuShardEngine() for _ in blocks]
    keys, edges = zip(*[e.begin_dev(b, s > 0, s + 1 < 3, 8) for s, (e, b) in enumerate(zip(engs, blocks))])
    counts = torch.tensor([int(e.shape[0]) for e in edges], dtype=torch.int32, device="cuda")
    assert int(counts.max()) > 7
    small = torch.zeros((3, 7, 3), dtype=torch.int32, device="cuda")
    with pytest.raises(RdgpuError) as ei:
        graph_solve_dev(torch.stack(keys), small, counts, 8)
    assert ei.value.code == 4
    cap = int(counts.max())
    full = torch.zeros((3, cap, 3), dtype=torch.int32, device="cuda")
    for s, e in enumerate(edges):
        full[s, : e.shape[0]] = e
    levels = graph_solve_dev(torch.stack(keys), full, counts, 8)
    for s, e in enumerate(engs):
        e.finish_dev(levels[s].contiguous())
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), orc.port.fill(dem, 8))


@pytest.mark.parametrize("cap", [0, 5])
def test_rccl_world_one_single_exchange(orc, tmp_path, cap):
    """fill_depressions_sharded over RCCL (world_size 1, the only size one GPU allows): the device protocol with its ONE
    all-gather; cap = 5 forces the repeat-with-exact-size path through the same collective code."""
    import torch.multiprocessing as mp

    dem = _dem()[:700, :900].copy()
    np.save(tmp_path / "dem.npy", dem)
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), cap), nprocs=1, join=True)
    assert np.array_equal(np.load(tmp_path / "rccl_fill.npy"), orc.port.fill(dem, 8))
