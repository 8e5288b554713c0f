"""The N>1 path on CPU: world_size 2 and 3 gloo process groups run the real exchange
(richdem_amd.sharded.fill_depressions_sharded: all-gather + rdgpu_fill_graph_solve, product code) around
a numpy MODEL of the shard-local engine (tests/shard_model.py); the GPU engine itself is covered by the
tiling-invariance tests in test_fill_gpu.py.  Results must equal the single-process oracle exactly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_dem(case):
    from richdem_amd.synth import fractal_dem, fractal_dem_int

    if case == "f32":
        return fractal_dem(61, 47, seed=5)
    if case == "i32_flats":
        return fractal_dem_int(50, 41, 6, 0.05)
    if case == "bowl":
        yy, xx = np.mgrid[0:36, 0:40]
        d = np.hypot(yy - 18, xx - 20).astype(np.float32)
        d[::5, ::3] -= 9
        return d
    raise KeyError(case)


def _worker(rank, world, port, case, topo, outdir):
    import torch.distributed as dist

    from richdem_amd.sharded import fill_depressions_sharded, row_split
    from shard_model import NumpyShardEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dem = _make_dem(case)
    r0, r1 = row_split(dem.shape[0], world)[rank]
    block = np.ascontiguousarray(dem[r0:r1]).copy()
    fill_depressions_sharded(block, topology=topo, engine=NumpyShardEngine(), comm_device="cpu")
    np.save(os.path.join(outdir, f"block{rank}.npy"), block)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case,topo", [("f32", "D8"), ("i32_flats", "D8"), ("bowl", "D4")])
def test_gloo_sharded_fill_matches_oracle(orc, tmp_path, world, case, topo):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, topo, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"block{r}.npy") for r in range(world)], axis=0)
    dem = _make_dem(case)
    exp = orc.port.fill(dem, 8 if topo == "D8" else 4)
    assert got.dtype == dem.dtype
    assert np.array_equal(got, exp)


def test_graph_solve_single_shard_and_errors(rd):
    from richdem_amd.sharded import graph_solve

    lv = graph_solve(np.zeros((1, 2, 7), np.uint32), [np.zeros((0, 3), np.uint32)], 8)
    assert lv.shape == (1, 2, 7) and not lv.any()
    with pytest.raises(rd.RdgpuError):
        graph_solve(np.zeros((2, 2, 7), np.uint32), [np.array([[99, 1, 5]], np.uint32), np.zeros((0, 3), np.uint32)], 8)


def test_model_engine_matches_oracle_single_process(orc):
    """The model engine + product graph solve, shard after shard in one process (no process group)."""
    from richdem_amd.sharded import graph_solve, row_split
    from shard_model import NumpyShardEngine

    dem = _make_dem("f32")
    for world in (2, 4, 5):
        blocks = [np.ascontiguousarray(dem[a:b]).copy() for a, b in row_split(dem.shape[0], world)]
        engs, keys, edges = [], [], []
        for s, blk in enumerate(blocks):
            e = NumpyShardEngine()
            k, ed = e.begin(blk, s > 0, s + 1 < world, 8)
            engs.append(e); keys.append(k); edges.append(ed)
        levels = graph_solve(np.stack(keys), edges, 8)
        for s, e in enumerate(engs):
            e.finish(levels[s])
        assert np.array_equal(np.concatenate(blocks, axis=0), orc.port.fill(dem, 8)), world


def _accum_dirs(case):
    import oracle

    dem = oracle.port.fill(_make_dem("f32"))
    dirs = oracle.port.flat_resolution(dem, np.float32(-9999))
    dirs[20:24, 30:34] = 255                      # a NoData hole straddling nothing in particular
    if case == "loop":                            # hand-made: a 2-cycle across the first cut of a 3-way split
        h = dirs.shape[0]
        y = h // 3 + (1 if h % 3 else 0)          # first row of the second block (row_split)
        dirs[y - 1, 40], dirs[y, 40] = 7, 3       # south / north: they point at each other
    return dirs


def _accum_worker(rank, world, port, case, protocol, outdir):
    import torch.distributed as dist

    from richdem_amd.sharded import d8_flow_accum_sharded, row_split
    from shard_model import NumpyAccumShard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dirs = _accum_dirs(case)
    r0, r1 = row_split(dirs.shape[0], world)[rank]
    block = np.ascontiguousarray(dirs[r0:r1])
    area = np.zeros(block.shape, np.float64)
    rounds = d8_flow_accum_sharded(block, area, 255, shard=NumpyAccumShard(), protocol=protocol)
    np.save(os.path.join(outdir, f"area{rank}.npy"), area)
    np.save(os.path.join(outdir, f"rounds{rank}.npy"), np.array([rounds]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case,protocol", [(2, "dem", "links"), (3, "dem", "links"), (3, "dem", "rounds"), (3, "loop", "links")])
def test_gloo_sharded_accumulation_matches_oracle(orc, tmp_path, world, case, protocol):
    """d8_flow_accum_sharded (product code: the one-exchange link protocol, its solve, the fallback to the outbox /
    all-gather / inject loop) around a Python model of the shard engine: tiling invariance of D8 accumulation
    (reference parallel_d8_accum/test_small.sh).  Loop-free directions take exactly ONE exchange; a direction loop
    across a cut is detected by the solve and handled by the iterated protocol with the reference's partial sums."""
    import torch.multiprocessing as mp

    mp.spawn(_accum_worker, args=(world, _free_port(), case, protocol, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"area{r}.npy") for r in range(world)], axis=0)
    dirs = _accum_dirs(case)
    assert np.array_equal(got, orc.port.d8_flow_accum(dirs, 255, np.float64))
    rounds = int(np.load(tmp_path / "rounds0.npy")[0])
    if case == "dem" and protocol == "links":
        assert rounds == 1
    else:
        assert rounds >= 2   # flow really crossed the cuts, exchange by exchange


def test_accum_link_solve_chain_and_cycle():
    """The cut-row forest on hand-made links: a path that crosses three cuts accumulates along the chain; two entries
    that lead into each other are reported as a loop."""
    import torch

    from richdem_amd.sharded import accum_link_solve

    world, w = 4, 5
    boxes = torch.zeros((world, 2, w), dtype=torch.int64)
    links = torch.full((world, 2, w), -1, dtype=torch.int32)
    DOWN = -(1 << 31)
    boxes[0, 1, 2] = (1 << 56) | 7          # rank 0 sends 7 down to column 2 of rank 1's first row
    links[1, 0, 2] = DOWN | 3               # what enters there leaves rank 1 downwards at column 3
    boxes[1, 1, 3] = (1 << 56) | 10         # rank 1's own cells also send 10 to that cell
    links[2, 0, 3] = DOWN | 1
    boxes[3, 0, 4] = (2 << 56) | 5          # rank 3 sends 5 up to column 4 of rank 2's last row
    links[2, 1, 4] = 0                      # ... which leaves rank 2 upwards at column 0
    inflow = accum_link_solve(boxes, links, world, w)
    assert inflow[1, 0, 2] == 7 and inflow[2, 0, 3] == 17 and inflow[3, 0, 1] == 17
    assert inflow[2, 1, 4] == 5 and inflow[1, 1, 0] == 5 and int(inflow.sum()) == 7 + 17 + 17 + 5 + 5
    links[1, 1, 0] = DOWN | 4               # rank 1's last row cell 0 sends it back down to column 4 of rank 2's first row ...
    links[2, 0, 4] = 0                      # ... whose path leaves upwards at column 0 again: a loop across the cut
    assert accum_link_solve(boxes, links, world, w) is None


def _flat_dem(case):
    import oracle
    from richdem_amd.synth import fractal_dem_int

    if case == "lakes":
        return oracle.port.fill(fractal_dem_int(57, 46, 8, 0.04)), np.int32(-9999)
    if case == "snake":
        comb = np.full((30, 41), 9, np.int32)
        comb[1:-1, 1:-1] = 5
        for x in range(4, 38, 4):
            comb[2:-2, x] = 9
            if (x // 4) % 2:
                comb[1:4, x] = 5
            else:
                comb[-4:-1, x] = 5
        comb[14, 0] = 1
        return comb, np.int32(-1)
    if case == "raw":
        rng = np.random.default_rng(3)
        return rng.integers(0, 3, (40, 37)).astype(np.int32), np.int32(-1)
    raise KeyError(case)


def _flat_worker(rank, world, port, case, outdir):
    import torch
    import torch.distributed as dist

    from richdem_amd.sharded import flat_resolution_sharded, row_split
    from shard_model import NumpyFlatShard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dem, nd = _flat_dem(case)
    r0, r1 = row_split(dem.shape[0], world)[rank]
    dirs = flat_resolution_sharded(torch.from_numpy(np.ascontiguousarray(dem[r0:r1])), nd, shard_factory=NumpyFlatShard)
    np.save(os.path.join(outdir, f"dirs{rank}.npy"), dirs.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["lakes", "snake", "raw"])
def test_gloo_sharded_flat_resolution_matches_oracle(orc, tmp_path, world, case):
    """flat_resolution_sharded's ghost-row gather, cut-row exchange loop and flat-height solve (product
    code) around a numpy model of the shard engine == barnes_flat_resolution_d8 of the whole raster."""
    import torch.multiprocessing as mp

    mp.spawn(_flat_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"dirs{r}.npy") for r in range(world)], axis=0)
    dem, nd = _flat_dem(case)
    assert np.array_equal(got, orc.port.flat_resolution(dem, nd))


def test_flat_model_blocks_single_process(orc):
    """Same model, every block driven by one process (flat_resolution_blocks), more shard counts."""
    import torch

    from richdem_amd.sharded import flat_resolution_blocks
    from shard_model import NumpyFlatShard

    for case in ("lakes", "snake"):
        dem, nd = _flat_dem(case)
        for world in (1, 4, 7):
            got, ex = flat_resolution_blocks(torch.from_numpy(dem), nd, world, shard_factory=NumpyFlatShard,
                                             solve=NumpyFlatShard().solve)
            assert np.array_equal(got.numpy(), orc.port.flat_resolution(dem, nd)), (case, world, ex)
            if case == "snake" and world == 7:
                assert ex[0] > 3      # the towards levels crossed the cuts repeatedly


def test_row_split_rejects_more_ranks_than_rows(rd):
    from richdem_amd.sharded import _check_block, row_split

    assert row_split(10, 3) == [(0, 3), (3, 6), (6, 10)]
    with pytest.raises(rd.RdgpuError, match="cannot be split"):
        row_split(2, 3)
    with pytest.raises(rd.RdgpuError, match="at least one row"):
        _check_block(np.zeros((0, 5), np.float32), "fill_depressions_sharded")


def test_accum_link_solve_random_forests():
    """Random loop-free link forests (mixed upward and downward crossings, chains over many cuts, several entries landing on
    one cell) against a plain recursive accumulation."""
    import sys

    import torch

    from richdem_amd.sharded import accum_link_solve

    sys.setrecursionlimit(20000)
    rng = np.random.default_rng(9)
    DOWN = -(1 << 31)
    for trial in range(25):
        world, w = int(rng.integers(2, 9)), int(rng.integers(1, 40))
        boxes = np.zeros((world, 2, w), np.int64)
        links = np.full((world, 2, w), -1, np.int64)
        # a potential per node that strictly decreases along every link keeps the graph loop free
        pot = rng.permutation(world * 2 * w).reshape(world, 2, w)
        for r in range(world):
            for k in range(2):
                for x in range(w):
                    if rng.random() < 0.6:
                        down = bool(rng.integers(2))
                        if (down and r + 1 >= world) or (not down and r == 0):
                            continue
                        col = int(rng.integers(w))
                        dst = (r + 1, 0, col) if down else (r - 1, 1, col)
                        if pot[dst] < pot[r, k, x]:
                            links[r, k, x] = (DOWN | col) if down else col
                    if rng.random() < 0.5:
                        boxes[r, k, x] = (int(rng.integers(1, 4)) << 56) | int(rng.integers(1, 10 ** 6))
        low = boxes & ((1 << 56) - 1)
        i0 = np.zeros_like(low)
        i0[1:, 0] = low[:-1, 1]
        i0[:-1, 1] = low[1:, 0]
        ups = {}
        for r in range(world):
            for k in range(2):
                for x in range(w):
                    L = int(links[r, k, x])
                    if L != -1:
                        dst = (r + 1, 0, L & 0x7FFFFFFF) if L < 0 else (r - 1, 1, L)
                        ups.setdefault(dst, []).append((r, k, x))
        memo = {}

        def total(node):
            if node not in memo:
                memo[node] = int(i0[node]) + sum(total(u) for u in ups.get(node, []))
            return memo[node]

        exp = np.array([[[total((r, k, x)) for x in range(w)] for k in range(2)] for r in range(world)], np.int64)
        got = accum_link_solve(torch.from_numpy(boxes), torch.from_numpy(links.astype(np.int32)), world, w)
        assert got is not None and np.array_equal(got.numpy(), exp), trial
