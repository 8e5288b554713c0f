"""GPU parity tests: d8_flow_directions, d8_flow_accum, FA_D8 through the C-ABI vs the oracle/reference goldens."""
import numpy as np
import pytest

from conftest import gen_cases
from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu


def test_d8_flow_accum_reference_golden(rd, fixtures):
    names = sorted({k.split("/")[1] for k in fixtures.files if k.startswith("accum/")})
    assert len(names) == 24  # reference tests/tests.cpp:135-146
    for nm in names:
        dirs, nd, exp = fixtures[f"accum/{nm}/d8"], int(fixtures[f"accum/{nm}/nodata"]), fixtures[f"accum/{nm}/out"]
        got = rd.d8_flow_accum(dirs, nd, np.int32)
        assert got.dtype == np.int32 and np.array_equal(got, exp), nm
        assert np.array_equal(rd.d8_flow_accum(dirs, nd, np.float64), exp.astype(np.float64)), nm
        assert np.array_equal(rd.d8_flow_accum(dirs, nd, np.float32), exp.astype(np.float32)), nm


def test_generated_reference_outputs(rd, generated):
    for name in gen_cases(generated):
        dem, nd = generated[f"{name}/dem"], generated[f"{name}/nodata"]
        for tag, src in (("raw", dem), ("filled", generated[f"{name}/fill_d8"])):
            assert np.array_equal(rd.d8_flow_directions(src, nd), generated[f"{name}/{tag}/d8_flowdirs"]), (name, tag)
            fr = generated[f"{name}/{tag}/flat_resolved_dirs"]
            assert np.array_equal(rd.d8_flow_accum(fr, 255, np.float64), generated[f"{name}/{tag}/d8_flow_accum_f64"]), (name, tag)
            assert np.array_equal(rd.FlowAccumulation(src, "D8", nodata=nd), generated[f"{name}/{tag}/fa_d8"]), (name, tag)


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (2, 2), (3, 3), (5, 70), (16, 64), (17, 65), (130, 67), (400, 523)])
@pytest.mark.parametrize("dtype", [np.float32, np.int32, np.uint8, np.int16, np.uint16, np.uint32, np.float64])
def test_d8_flowdirs_shapes_dtypes(rd, orc, shape, dtype):
    h, w = shape
    z = fractal_dem(w, h, seed=h * 7 + w)
    if dtype in (np.float32, np.float64):
        dem = z.astype(dtype)
        nd = dtype(-9999)
    else:
        dem = np.floor((z - z.min()) * 0.1).astype(dtype)   # quantised: many ties exercise the tie rule
        nd = dtype(3)
    if h > 4 and w > 4:
        dem[h // 2, w // 3] = nd
    got = rd.d8_flow_directions(dem, nd)
    assert np.array_equal(got, orc.port.d8_flowdirs(dem, nd))


def test_d8_flowdirs_tie_rule_exhaustive(rd, orc):
    """All 3x3 neighbourhoods over a 3-letter alphabet (3^9 = 19683 cases) laid out as one raster of
    isolated 3x3 blocks: pins the cardinal-before-diagonal tie rule of d8_FlowDir (d8_flowdirs.hpp:63-71)."""
    n = 3 ** 9
    cols = 192
    rows = (n + cols - 1) // cols
    dem = np.full((rows * 4 + 1, cols * 4 + 1), 9, np.int32)
    for k in range(n):
        digs = [(k // 3 ** i) % 3 for i in range(9)]
        r, c = divmod(k, cols)
        dem[r * 4 + 1:r * 4 + 4, c * 4 + 1:c * 4 + 4] = np.array(digs).reshape(3, 3)
    got = rd.d8_flow_directions(dem, np.int32(-1))
    assert np.array_equal(got, orc.port.d8_flowdirs(dem, np.int32(-1)))


def _accum_case(rd, orc, dem, nd):
    dirs = orc.port.flat_resolution(dem, nd)
    for dt in (np.int32, np.float64, np.float32):
        assert np.array_equal(rd.d8_flow_accum(dirs, 255, dt), orc.port.d8_flow_accum(dirs, 255, dt))
    raw = orc.port.d8_flowdirs(dem, nd)   # still has NO_FLOW cells
    assert np.array_equal(rd.d8_flow_accum(raw, 255, np.float64), orc.port.d8_flow_accum(raw, 255, np.float64))
    assert np.array_equal(rd.FlowAccumulation(dem, "D8", nodata=nd), orc.port.fa_d8(dem, nd))


def test_accum_fractal_and_flats(rd, orc):
    z = fractal_dem(700, 500, seed=17)
    _accum_case(rd, orc, z, np.float32(-9999))
    _accum_case(rd, orc, orc.port.fill(z), np.float32(-9999))
    zi = fractal_dem_int(400, 300, 18, 0.05)
    _accum_case(rd, orc, orc.port.fill(zi), np.int32(-9999))
    hole = z.copy(); hole[100:140, 200:260] = -9999.0; hole[:, :4] = -9999.0
    _accum_case(rd, orc, hole, np.float32(-9999))


def test_accum_direction_loops_match_reference_semantics(rd, orc):
    """Hand-made directions with a 2-cycle and a chain hanging below it: the reference never completes
    those cells (d8_methods.hpp:104-131); the engine must leave the same partial values."""
    dirs = np.zeros((6, 8), np.uint8)
    dirs[:] = 5                      # everything flows east ...
    dirs[2, 3], dirs[2, 4] = 5, 1    # ... except a 2-cycle (2,3) <-> (2,4)
    dirs[4, 2] = 255
    for dt in (np.int32, np.float64):
        assert np.array_equal(rd.d8_flow_accum(dirs, 255, dt), orc.port.d8_flow_accum(dirs, 255, dt))


def test_fa_d8_weights(rd, orc):
    z = fractal_dem(300, 260, seed=23)
    rng = np.random.default_rng(1)
    wi = rng.integers(0, 5, z.shape).astype(np.float64)         # integer-valued weights: exact
    assert np.array_equal(rd.FlowAccumulation(z, "D8", nodata=-9999, weights=wi), orc.port.fa_d8(z, np.float32(-9999), wi))
    wf = rng.random(z.shape)                                     # general weights: summation order differs
    got, exp = rd.FlowAccumulation(z, "D8", nodata=-9999, weights=wf), orc.port.fa_d8(z, np.float32(-9999), wf)
    assert np.allclose(got, exp, rtol=1e-12, atol=0)
    # north_star tolerance: within 1 ULP after a float32 cast
    g32, e32 = got.astype(np.float32), exp.astype(np.float32)
    assert (np.abs(g32.view(np.int32).astype(np.int64) - e32.view(np.int32).astype(np.int64)) <= 1).all()
    with pytest.raises(rd.RdgpuError, match="same dimensions"):
        rd.FlowAccumulation(z, "D8", weights=np.ones((3, 3)))


def test_fa_d8_unit_weights_take_the_counting_path_with_the_same_result(rd, orc, monkeypatch):
    """All weights 1 (FA_D8's default) is detected on the device and counted through the tile links; the weighted engine
    on the same input gives the same doubles, NoData cells and cells draining into NoData included."""
    z = fractal_dem(900, 700, seed=29)
    z[300:340, 100:180] = -9999.0
    z[:, -3:] = -9999.0
    for dem in (z, orc.port.fill(z)):
        exp = orc.port.fa_d8(dem, np.float32(-9999))
        fast = rd.FlowAccumulation(dem, "D8", nodata=-9999)
        monkeypatch.setenv("RDGPU_ACCUM_UNIT", "0")
        slow = rd.FlowAccumulation(dem, "D8", nodata=-9999)
        monkeypatch.delenv("RDGPU_ACCUM_UNIT")
        assert np.array_equal(fast, exp) and np.array_equal(slow, exp)
    # r05: weights=None goes through rdgpu_fa_d8_unit_<T> (the caller KNOWS the weights are ones: nothing read, nothing uploaded);
    # an explicit array of ones through the plain entry (detected on the device), and the HBM-resident pair -- same doubles
    import torch

    dem = orc.port.fill(z)
    exp = orc.port.fa_d8(dem, np.float32(-9999))
    assert np.array_equal(rd.FlowAccumulation(dem, "D8", nodata=-9999, weights=np.ones(dem.shape)), exp)
    t = torch.from_numpy(dem).cuda()
    a1 = torch.ones(dem.shape, dtype=torch.float64, device="cuda")
    a2 = torch.full(dem.shape, -7.0, dtype=torch.float64, device="cuda")        # (output only: whatever it held is ignored)
    rd.fa_d8_dev(t, -9999.0, a1)
    rd.fa_d8_dev(t, -9999.0, a2, unit_weights=True)
    torch.cuda.synchronize()
    assert np.array_equal(a1.cpu().numpy(), exp) and np.array_equal(a2.cpu().numpy(), exp)
    one_off = np.ones(z.shape)
    one_off[450, 350] = 1.0 + 2.0 ** -40    # a single weight that is not 1: the weighted path, exactly representable sums
    assert np.array_equal(rd.FlowAccumulation(z, "D8", nodata=-9999, weights=one_off), orc.port.fa_d8(z, np.float32(-9999), one_off))


def test_large_accumulation_exact(rd, orc):
    """3000x3000 filled DEM: long cross-XCD flow paths; exact equality with the oracle for both engines
    (packed-u64 unit path and f64 release/acquire path)."""
    z = fractal_dem(3000, 3000, seed=29)
    zf = rd.FillDepressions(z)
    nd = np.float32(-9999)
    dirs = orc.port.flat_resolution(zf, nd)
    exp = orc.port.d8_flow_accum(dirs, 255, np.float64)
    assert np.array_equal(rd.d8_flow_accum(dirs, 255, np.float64), exp)
    assert exp.max() > 1e5
    assert np.array_equal(rd.FlowAccumulation(zf, "D8", nodata=nd), orc.port.fa_d8(zf, nd))
    assert np.array_equal(rd.FlowAccumulation(z, "D8", nodata=nd), orc.port.fa_d8(z, nd))


def _accum_blocks_on_one_gpu(rd, dirs_np, world, dtype):
    """What the ranks of d8_flow_accum_sharded do, block after block on one GPU (no process group)."""
    import torch

    from richdem_amd.sharded import GpuAccumShard, row_split

    dirs = torch.from_numpy(dirs_np).cuda()
    spl = row_split(dirs.shape[0], world)
    blocks = [dirs[a:b].contiguous() for a, b in spl]
    shards = []
    for s, blk in enumerate(blocks):
        sh = GpuAccumShard()
        sh.begin(blk, 255, blocks[s - 1][-1] if s > 0 else None, blocks[s + 1][0] if s + 1 < world else None)
        shards.append(sh)
    rounds = 0
    while True:
        outs = [sh.outbox() for sh in shards]
        rounds += 1
        if not any(bool((o != 0).any().item()) for o in outs):
            break
        for s, sh in enumerate(shards):
            sh.inject(outs[s - 1][1] if s > 0 else None, outs[s + 1][0] if s + 1 < world else None)
        assert rounds < 10000
    areas = []
    for sh, blk in zip(shards, blocks):
        a = torch.empty(blk.shape, dtype=dtype, device="cuda")
        sh.finish(a)
        areas.append(a)
    return torch.cat(areas, 0).cpu().numpy(), rounds


def _accum_blocks_one_exchange(rd, dirs_np, world, dtype):
    """The one-exchange protocol block after block on one GPU: begin_local, (outbox, links) of every block stacked as the
    all-gather would, accum_link_solve, add_paths, finish.  Returns (area, True) or (None, False) when a loop is reported."""
    import torch

    from richdem_amd.sharded import GpuAccumShard, accum_link_solve, row_split

    dirs = torch.from_numpy(dirs_np).cuda()
    spl = row_split(dirs.shape[0], world)
    blocks = [dirs[a:b].contiguous() for a, b in spl]
    shards, boxes, links, pend = [], [], [], []
    for s, blk in enumerate(blocks):
        sh = GpuAccumShard()
        sh.begin_local(blk, 255, blocks[s - 1][-1] if s > 0 else None, blocks[s + 1][0] if s + 1 < world else None)
        shards.append(sh)
        boxes.append(sh.outbox())
        lk, pn = sh.links()
        links.append(lk)
        pend.append(pn)
    w = dirs.shape[1]
    inflow = None
    if int(torch.cat(pend).sum().item()) == 0:
        inflow = accum_link_solve(torch.stack(boxes), torch.stack(links), world, w)
    if inflow is None:
        for sh in shards:
            sh.abort()
        return None, False
    areas = []
    for s, (sh, blk) in enumerate(zip(shards, blocks)):
        sh.add_paths(inflow[s, 0] if s > 0 else None, inflow[s, 1] if s + 1 < world else None)
        a = torch.empty(blk.shape, dtype=dtype, device="cuda")
        sh.finish(a)
        areas.append(a)
    return torch.cat(areas, 0).cpu().numpy(), True


def test_sharded_accumulation_one_exchange(rd, orc):
    """programs/parallel_d8_accum's protocol: ONE exchange whatever the number of cut crossings -- 2..64 blocks (down to
    single-row blocks between two cuts), all output types, NoData holes, raw directions with NO_FLOW cells; a direction
    loop (inside a block or across a cut) is reported, not mis-accumulated."""
    import torch

    z = orc.port.fill(fractal_dem(400, 333, seed=55))
    dirs = orc.port.flat_resolution(z, np.float32(-9999))
    dirs[100:110, 50:70] = 255
    for world in (2, 3, 9, 64):
        for dt, ndt in ((torch.float64, np.float64), (torch.int32, np.int32), (torch.float32, np.float32)):
            got, ok = _accum_blocks_one_exchange(rd, dirs, world, dt)
            assert ok and np.array_equal(got, orc.port.d8_flow_accum(dirs, 255, ndt)), (world, dt)
    raw = orc.port.d8_flowdirs(fractal_dem(90, 64, seed=56), np.float32(-9999))
    for world in (32, 64):   # 64 blocks of 64 rows: every block is a single row between two cuts
        got, ok = _accum_blocks_one_exchange(rd, raw, world, torch.float64)
        assert ok and np.array_equal(got, orc.port.d8_flow_accum(raw, 255, np.float64))
    loop = dirs.copy()
    loop[199, 40], loop[200, 40] = 7, 3          # a 2-cycle across the cut of a 2-way split (rows 0..199 | 200..399)
    assert _accum_blocks_one_exchange(rd, loop, 2, torch.float64) == (None, False)
    loop = dirs.copy()
    loop[50, 40], loop[51, 40] = 7, 3            # the same inside the first block
    assert _accum_blocks_one_exchange(rd, loop, 2, torch.float64) == (None, False)
    got, _ = _accum_blocks_on_one_gpu(rd, loop, 2, torch.float64)   # ... and the iterated protocol reproduces the reference
    assert np.array_equal(got, orc.port.d8_flow_accum(loop, 255, np.float64))


def test_sharded_accumulation_tiling_invariance(rd, orc):
    """Row-block shards of d8_flow_accum give the whole-raster answer exactly (the reference's distributed
    test idea, programs/parallel_d8_accum/test_small.sh), for 2..9 blocks and all output types."""
    import torch

    z = orc.port.fill(fractal_dem(400, 333, seed=55))
    dirs = orc.port.flat_resolution(z, np.float32(-9999))
    dirs[100:110, 50:70] = 255
    for world in (2, 3, 9):
        for dt, ndt in ((torch.float64, np.float64), (torch.int32, np.int32), (torch.float32, np.float32)):
            got, rounds = _accum_blocks_on_one_gpu(rd, dirs, world, dt)
            assert np.array_equal(got, orc.port.d8_flow_accum(dirs, 255, ndt)), (world, dt)
            assert rounds >= 2
    # 2-row shards, raw (unresolved) directions with NO_FLOW cells, fixtures
    raw = orc.port.d8_flowdirs(fractal_dem(90, 64, seed=56), np.float32(-9999))
    got, _ = _accum_blocks_on_one_gpu(rd, raw, 32, torch.float64)
    assert np.array_equal(got, orc.port.d8_flow_accum(raw, 255, np.float64))


def test_sharded_pipeline_one_rank_group(rd, orc):
    """fill -> d8 directions -> accumulation through the collective entry points on a 1-rank RCCL group."""
    import os

    import torch
    import torch.distributed as dist

    from richdem_amd.sharded import d8_flow_accum_sharded, d8_flow_directions_sharded, fill_depressions_sharded

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29535")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        z = fractal_dem(600, 450, seed=57)
        t = torch.from_numpy(z).cuda()
        fill_depressions_sharded(t)
        dirs = d8_flow_directions_sharded(t, -9999.0, flats=True)
        area = torch.empty(t.shape, dtype=torch.float64, device="cuda")
        d8_flow_accum_sharded(dirs, area)
        ez = orc.port.fill(z)
        ed = orc.port.flat_resolution(ez, np.float32(-9999))
        assert np.array_equal(t.cpu().numpy(), ez)
        assert np.array_equal(dirs.cpu().numpy(), ed)
        assert np.array_equal(area.cpu().numpy(), orc.port.d8_flow_accum(ed, 255, np.float64))
    finally:
        dist.destroy_process_group()


def test_sharded_directions_halo(rd, orc):
    """d8 directions of row blocks with a 1-row halo equal the whole-raster directions."""
    import torch

    z = fractal_dem(300, 257, seed=58)
    z[120:130, 40:60] = -9999.0
    exp = orc.port.d8_flowdirs(z, np.float32(-9999))
    from richdem_amd.sharded import row_split

    t = torch.from_numpy(z).cuda()
    for world in (2, 5):
        outs = []
        for s, (a, b) in enumerate(row_split(z.shape[0], world)):
            lo, hi = max(a - 1, 0), min(b + 1, z.shape[0])
            halo = t[lo:hi].contiguous()
            d = torch.empty(halo.shape, dtype=torch.uint8, device="cuda")
            rd.d8_flow_directions_dev(halo, -9999.0, d)
            outs.append(d[a - lo : a - lo + (b - a)])
        assert np.array_equal(torch.cat(outs, 0).cpu().numpy(), exp), world


def test_multi_device_accumulation_entry_on_one_gpu(rd, orc, monkeypatch):
    """rdgpu_d8_flow_accum_multi_<A> (one process, a list of devices: the reference's parallel_d8_accum driver as a library
    call).  With device 0 listed several times the row blocks go through exactly the multi-device code -- a worker thread
    per device, ghost rows, the one exchange through the host's link solve, add_paths, per-block finish -- and the result
    equals the single-device call on every cell; direction loops fall back to one device and keep the reference's partial
    sums; RDGPU_DEVICES routes the plain host entry the same way."""
    import ctypes

    from richdem_amd._lib import check, lib
    from richdem_amd.synth import fractal_dem

    z = orc.port.fill(fractal_dem(530, 410, seed=77))
    dirs = orc.port.flat_resolution(z, np.float32(-9999))
    dirs[100:104, 200:230] = 255                                   # a NoData hole
    h, w = dirs.shape
    for dt, suf in ((np.float64, "f64"), (np.int32, "i32"), (np.float32, "f32")):
        exp = orc.port.d8_flow_accum(dirs, 255, dt)
        for devs in ([0], [0, 0], [0] * 7, [0] * 64):
            out = np.empty((h, w), dt)
            arr = (ctypes.c_int * len(devs))(*devs)
            check(getattr(lib(), f"rdgpu_d8_flow_accum_multi_{suf}")(dirs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint8(255), w, h,
                                                                      out.ctypes.data_as(ctypes.c_void_p), arr, len(devs)), "multi")
            assert np.array_equal(out, exp), (suf, len(devs))
    # r05: the default exchange stays on the devices (boxes and links pulled by hipMemcpyPeerAsync behind events, the forest
    # over the cut-row cells solved on devices[0] by pointer doubling, the inflows pushed back); RDGPU_MULTI_HOST_STAGED=1 is
    # the r02-r04 exchange through host vectors and the host's Kahn order: the same doubles, also on rivers that cross many cuts
    snake = np.zeros((400, 90), np.uint8)
    snake[:, :] = 7                                                # south ...
    snake[:, 1::2] = 3                                             # ... and north in alternate columns
    snake[-1, 0:-1:2] = 5                                          # turn east at the bottom / top: one path through every cut, 90 times
    snake[0, 1:-1:2] = 5
    snake[0, -1] = 3 if (snake.shape[1] - 1) % 2 else 7
    for ddirs in (dirs, snake):
        hh, ww = ddirs.shape
        want = orc.port.d8_flow_accum(ddirs, 255, np.float64)
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("RDGPU_MULTI_HOST_STAGED", env)
            out = np.empty((hh, ww), np.float64)
            arr = (ctypes.c_int * 9)(*([0] * 9))
            check(lib().rdgpu_d8_flow_accum_multi_f64(ddirs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint8(255), ww, hh,
                                                      out.ctypes.data_as(ctypes.c_void_p), arr, 9), "multi")
            if env:
                monkeypatch.delenv("RDGPU_MULTI_HOST_STAGED")
            assert np.array_equal(out, want), (ddirs.shape, env, float(out.max()), float(want.max()))
    # raw directions with loops: reported by the blocks / the link solve, then one device
    rng = np.random.default_rng(5)
    loops = rng.integers(0, 9, (260, 300)).astype(np.uint8)
    exp = orc.port.d8_flow_accum(loops, 255, np.float64)
    out = np.empty(loops.shape, np.float64)
    arr = (ctypes.c_int * 5)(0, 0, 0, 0, 0)
    check(lib().rdgpu_d8_flow_accum_multi_f64(loops.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint8(255), 300, 260,
                                              out.ctypes.data_as(ctypes.c_void_p), arr, 5), "multi")
    assert np.array_equal(out, exp)
    monkeypatch.setenv("RDGPU_DEVICES", "0,0,0")
    assert np.array_equal(rd.d8_flow_accum(dirs, 255, np.float64), orc.port.d8_flow_accum(dirs, 255, np.float64))
    assert np.array_equal(rd.d8_flow_accum(loops, 255, np.float64), exp)
    monkeypatch.delenv("RDGPU_DEVICES")
    arr = (ctypes.c_int * 2)(0, 99)
    assert lib().rdgpu_d8_flow_accum_multi_f64(dirs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint8(255), w, h,
                                               out.ctypes.data_as(ctypes.c_void_p), arr, 2) != 0


def test_an_exit_with_more_than_255_in_links(rd, orc):
    """Tile links: an exit cell can be handed flow by every exit of the neighbouring tiles whose path ends at it.  Here
    the centre tile of a 3 x 3 arrangement receives flow through all 256 cells of its four edges and funnels it through
    one corner: 256+ in-links on one node of the link forest (an 8-bit pending count wrapped there until r03 -- found by
    the full-size digest test on FA_D8, 1899 cells of 1.6e9)."""
    T = 64
    d = np.zeros((3 * T, 3 * T), np.uint8)
    d[:, :] = 5                                   # default: east
    d[0:T, T:2 * T] = 7                           # top tile: south, into the centre
    d[2 * T:, T:2 * T] = 3                        # bottom tile: north, into the centre
    d[T:2 * T, 2 * T:] = 1                        # right tile: west, into the centre
    d[0:T, 2 * T:] = 1                            # top-right corner tile: west, into the top tile
    d[2 * T:, 0:T] = 5                            # bottom-left: east, into the bottom tile
    c = d[T:2 * T, T:2 * T]                       # the centre tile: east along the rows, south down the last column
    c[:, :] = 5
    c[:, T - 1] = 7
    c[T - 1, T - 1] = 6                           # ... and out through the corner, south-east
    d[2 * T:, 2 * T:] = 5                         # the tile it leaves into drains off the raster
    for dt in (np.float64, np.int32):
        exp = orc.port.d8_flow_accum(d, 255, dt)
        assert np.array_equal(rd.d8_flow_accum(d, 255, dt), exp), dt
    assert exp[2 * T - 1, 2 * T - 1] > 3 * T * T  # everything the four side tiles and the centre hold goes through that corner
