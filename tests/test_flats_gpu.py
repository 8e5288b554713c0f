"""GPU parity tests: barnes_flat_resolution_d8(alter=false) through the C-ABI vs the oracle
(the reference has no golden file for flat resolution -- SURVEY.md section 4; parity is pinned by the
outputs of the unmodified reference in tests/golden/ref_generated.npz and by the oracle)."""
import numpy as np
import pytest

from conftest import gen_cases
from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu


def canon(labels):
    """Canonical partition ids: 1 + lowest cell index of each label (0 stays 0)."""
    flat = labels.ravel()
    out = np.zeros_like(flat)
    ids, first = np.unique(flat, return_index=True)
    lut = dict(zip(ids.tolist(), first.tolist()))
    nz = flat != 0
    out[nz] = np.vectorize(lambda v: lut[v] + 1, otypes=[flat.dtype])(flat[nz]) if nz.any() else 0
    return out.reshape(labels.shape)


def check(rd, orc, dem, nd):
    got = rd.barnes_flat_resolution_d8(dem, nd)
    exp = orc.port.flat_resolution(dem, nd)
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{len(bad)} dirs differ; first {bad[:5].tolist()} got {got[tuple(bad[0])]} exp {exp[tuple(bad[0])]}")
    dirs, mask, labels = rd.resolve_flats(dem, nd)
    _, emask, elabels = orc.port.resolve_flats(dem, nd)
    assert np.array_equal(dirs, exp)
    assert np.array_equal(mask, emask), "flat_mask differs"
    assert np.array_equal(labels, canon(elabels)), "flat partition differs"
    return got


def test_generated_reference_outputs(rd, generated):
    for name in gen_cases(generated):
        dem, nd = generated[f"{name}/dem"], generated[f"{name}/nodata"]
        for tag, src in (("raw", dem), ("filled", generated[f"{name}/fill_d8"])):
            got = rd.barnes_flat_resolution_d8(src, nd)
            assert np.array_equal(got, generated[f"{name}/{tag}/flat_resolved_dirs"]), (name, tag)
            _, mask, labels = rd.resolve_flats(src, nd)
            assert np.array_equal(mask, generated[f"{name}/{tag}/flat_mask"]), (name, tag)
            assert np.array_equal(labels, canon(generated[f"{name}/{tag}/flat_labels"])), (name, tag)


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (3, 3), (4, 9), (16, 64), (17, 65), (100, 130), (300, 421)])
@pytest.mark.parametrize("scale", [1.0, 0.1, 0.02])
def test_filled_integer_dems(rd, orc, shape, scale):
    h, w = shape
    dem = orc.port.fill(fractal_dem_int(w, h, seed=3 * h + w, scale=scale))
    check(rd, orc, dem, np.int32(-9999))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.uint16, np.uint8, np.uint32])
def test_dtypes(rd, orc, dtype):
    z = fractal_dem(260, 190, seed=41)
    dem = np.floor((z - z.min()) * 0.08).astype(dtype)
    dem = orc.port.fill(dem) if dtype != np.float64 else orc.port.fill(dem.astype(np.float32)).astype(np.float64)
    nd = dtype(250) if dtype == np.uint8 else dtype(60000) if dtype in (np.uint16, np.uint32) else dtype(-9999)
    check(rd, orc, dem, nd)


def test_unfilled_dems_with_undrainable_flats(rd, orc):
    """Raw (unfilled) DEMs: pits and flats without outlets must stay NO_FLOW (flat_resolution.hpp:491-500)."""
    check(rd, orc, fractal_dem_int(300, 200, 51, 0.05), np.int32(-9999))
    check(rd, orc, fractal_dem(200, 150, 52), np.float32(-9999))
    rng = np.random.default_rng(5)
    check(rd, orc, rng.integers(0, 3, (150, 170)).astype(np.int32), np.int32(-1))
    check(rd, orc, np.zeros((40, 50), np.float32), np.float32(-1))          # one flat, drained by the edge cells
    mesa = np.zeros((60, 60), np.int32); mesa[20:40, 20:40] = 5              # mesa: no high edges
    check(rd, orc, mesa, np.int32(-1))
    bowl = np.full((60, 60), 5, np.int32); bowl[20:40, 20:40] = 0             # sunken flat: no outlet
    check(rd, orc, bowl, np.int32(-1))


def test_reference_data_dems(rd, orc, fixtures):
    """The reference's un-asserted flat inputs data/*.dem."""
    names = sorted({k.split("/")[1] for k in fixtures.files if k.startswith("data/")})
    assert "multi_flat" in names and "garbrecht" in names
    for nm in names:
        dem, nd = fixtures[f"data/{nm}/dem"], fixtures[f"data/{nm}/nodata"]
        check(rd, orc, dem, nd)
        check(rd, orc, orc.port.fill(dem), nd)


def test_nodata_holes_and_equal_elevation_bridges(rd, orc):
    dem = orc.port.fill(fractal_dem_int(240, 200, 61, 0.04))
    dem[50:70, 80:120] = -9999
    dem[:, :3] = -9999
    check(rd, orc, dem, np.int32(-9999))
    # two NO_FLOW pockets of one elevation joined only through cells that have flow:
    # they share ONE label (and one flat_height) in the reference
    d = np.full((12, 30), 9, np.int32)
    d[3:9, 3:27] = 5
    d[5:7, 5:10] = 5; d[5:7, 20:25] = 5
    d[4:8, 12:18] = 4          # a lower notch in the middle gives the 5-cells around it a direction
    d[6, 14:16] = 3; d[7:, 14] = 2
    check(rd, orc, d, np.int32(-1))


def test_large_lake_many_bfs_levels(rd, orc):
    """A 1500x1100 DEM quantised so hard that lakes are hundreds of cells across."""
    dem = orc.port.fill(fractal_dem_int(1500, 1100, 71, 0.01))
    check(rd, orc, dem, np.int32(-9999))
    st = rd.lib  # noqa: F841


def test_full_pipeline_fill_dirs_accum(rd, orc):
    """rd_flood_for_flowdirs-style chain (apps/rd_d8_flowdirs.cpp:12-25 + d8_flow_accum) entirely on the GPU path."""
    z = fractal_dem(1200, 900, seed=81)
    nd = np.float32(-9999)
    filled = rd.FillDepressions(z)
    dirs = rd.barnes_flat_resolution_d8(filled, nd)
    area = rd.d8_flow_accum(dirs, 255, np.float64)
    efilled = orc.port.fill(z)
    edirs = orc.port.flat_resolution(efilled, nd)
    assert np.array_equal(filled, efilled)
    assert np.array_equal(dirs, edirs)
    assert (dirs[1:-1, 1:-1] != 0).all()      # a filled DEM has no undrainable flats
    assert np.array_equal(area, orc.port.d8_flow_accum(edirs, 255, np.float64))


def test_stencil_relaxation_engine_still_agrees(rd, orc, monkeypatch):
    """RDGPU_FLAT_BITS=0 selects the stencil relaxation (k_flat_relax) instead of the bitmap search, on a single device
    and in the row-block shards: same directions, same flat_mask."""
    import torch

    from richdem_amd.sharded import flat_resolution_blocks

    dem = orc.port.fill(fractal_dem_int(700, 500, 72, 0.02))
    nd = np.int32(-9999)
    exp = orc.port.flat_resolution(dem, nd)
    monkeypatch.setenv("RDGPU_FLAT_BITS", "0")
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), exp)
    got, _ = flat_resolution_blocks(torch.from_numpy(dem).cuda(), nd, 5)
    assert np.array_equal(got.cpu().numpy(), exp)
    monkeypatch.delenv("RDGPU_FLAT_BITS")
    got, _ = flat_resolution_blocks(torch.from_numpy(dem).cuda(), nd, 5)
    assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("switch", ["RDGPU_FLAT_ASYNC=0", "RDGPU_FLAT_ASYNC=100000", "RDGPU_FLAT_AWAY_BESIDE=0", "RDGPU_FLAT_ASYNC_BLOCKS=3",
                                    "RDGPU_RFE_LEAN=0", "RDGPU_RFE_OVERLAP=0", "RDGPU_RFE_AWAY_BESIDE=1", "RDGPU_FLAT_ASYNC_FAIL=1",
                                    "RDGPU_FLAT_Q=0", "RDGPU_FLAT_PLANES=0", "RDGPU_FLAT_PLANES_MAX=40", "RDGPU_FLAT_CLASS_BITMAPS=1", "RDGPU_FLAT_STATIC=0"])
def test_search_schedules_give_the_same_levels(rd, orc, monkeypatch, switch):
    """The bitmap search in rounds to the end, with its asynchronous tail from the first batch on (k_relax_bits_async),
    with the away search after instead of beside the towards tail, on three resident blocks; ResolveFlatsEpsilon with the
    older label path, on one stream, on three; a tail that is declared failed and finished in rounds from every tile
    (RDGPU_FLAT_ASYNC_FAIL): the fixed point does not depend on the schedule -- directions and
    epsilon-resolved elevations equal the oracle's under every switch, on lakes that span many 64 x 64 tiles."""
    k, v = switch.split("=")
    dem = orc.port.fill(fractal_dem_int(1300, 900, 73, 0.01))
    nd = np.int32(-9999)
    edirs = orc.port.flat_resolution(dem, nd)
    femp = orc.port.fill(fractal_dem(1100, 700, seed=74) * np.float32(0.02)).astype(np.float32)
    ffl = np.floor(femp).astype(np.float32)            # float DEM with wide lakes
    eeps = orc.port.resolve_flats_epsilon(ffl, np.float32(-9999))
    monkeypatch.setenv(k, v)
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), edirs)
    st = rd.flat_stats()                                # rdgpu_flat_get_async_stats: what the resident wavefronts did
    if switch == "RDGPU_FLAT_ASYNC=100000":             # both searches hand over after the first batch of rounds
        assert 1 <= st["tail_launches"] <= 2 and st["tail_failures"] == 0 and st["tail_visits"] >= st["tail_live_tiles"] > 0, st
    if switch == "RDGPU_FLAT_ASYNC=0":
        assert st["tail_launches"] == 0 and st["tail_visits"] == 0, st
    assert rd.ResolveFlats(ffl, nodata=np.float32(-9999)).tobytes() == eeps.tobytes()
    monkeypatch.delenv(k)
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), edirs)
    assert rd.ResolveFlats(ffl, nodata=np.float32(-9999)).tobytes() == eeps.tobytes()


def test_level_planes_long_channels_and_overflow(rd, orc, monkeypatch):
    """r06 (csrc/flat_planes.inc): the level fields as 16 bit planes per 64 x 64 tile.  A serpentine channel of one elevation
    -- breadth-first levels in the thousands, crossing every tile many times, several 256-level segments per visit -- and
    rasters narrower / shorter than a tile; then the same channel with the planes' range cut to 300 levels
    (RDGPU_FLAT_PLANES_MAX): the search overflows and the int engine takes over, same directions."""
    h, w = 330, 410
    dem = np.full((h, w), 50, np.int32)
    for k, y in enumerate(range(2, h - 2, 4)):       # walls every 4 rows, a gap at alternating ends
        dem[y, 2:w - 2] = 10
        x = w - 3 if k % 2 == 0 else 2
        if y + 4 < h - 2:
            dem[y:y + 5, x] = 10
    dem[2, 1] = 5; dem[2, 0] = 1                     # the outlet
    nd = np.int32(-9999)
    exp = orc.port.flat_resolution(dem, nd)
    assert (exp[dem == 10] != 0).all()
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), exp)
    for shape in ((70, 9), (9, 70), (64, 64), (65, 129), (1, 200), (200, 1), (3, 3)):
        rng = np.random.default_rng(shape[0] * 7 + shape[1])
        small = orc.port.fill(rng.integers(0, 4, shape).astype(np.int32))
        assert np.array_equal(rd.barnes_flat_resolution_d8(small, nd), orc.port.flat_resolution(small, nd)), shape
    monkeypatch.setenv("RDGPU_FLAT_PLANES_MAX", "300")
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), exp)
    monkeypatch.setenv("RDGPU_FLAT_ASYNC", "100000")   # ... and when the overflow happens in the asynchronous tail
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), exp)


def test_directions_from_masks_modulo_8(rd, orc, monkeypatch):
    """r05: the directions-only entry derives the directions from the two level planes alone, as masks modulo 8 (adjacent
    cells of a flat differ by at most one level in either field), cells next to a low edge getting theirs in the
    classification pass; RDGPU_FLAT_Q=0 is the r02-r04 pass that also reads and compares the DEM.
    Both equal the oracle: plateaus whose levels run far past 4 and past 256, a flat with an island of higher ground (high
    edges inside), flats without an outlet (no direction), NoData beside flats, flats wider than a tile and crossing the tile
    borders at every offset, under both search schedules."""
    rng = np.random.default_rng(91)
    cases = {}
    cases["terraces"] = orc.port.fill(fractal_dem_int(900, 700, 81, 0.01))
    lake = np.full((300, 520), 50, np.int32)
    lake[0, :] = lake[-1, :] = lake[:, 0] = lake[:, -1] = 90
    lake[150, 0] = 10                                           # one outlet: levels up to ~520
    lake[100:140, 200:260] = 70                                 # an island: high edges in the middle of the flat
    cases["lake with an island, one outlet"] = lake
    pit = np.full((130, 170), 40, np.int32)
    pit[0, :] = pit[-1, :] = pit[:, 0] = pit[:, -1] = 60
    cases["a flat without an outlet"] = pit
    nod = orc.port.fill(fractal_dem_int(500, 400, 82, 0.02))
    nod[rng.random(nod.shape) < 0.03] = -9999
    nod[200:230, 100:300] = -9999
    cases["NoData beside flats"] = nod
    for dx in (0, 1, 31, 63):
        a = np.full((200, 260), 20, np.int32)
        a[:, :dx + 3] = 100 - np.arange(dx + 3)[None, :]
        a[:, -5:] = 5
        cases[f"flat crossing tiles, offset {dx}"] = a
    fl = np.floor(orc.port.fill(fractal_dem(800, 600, seed=83)) * np.float32(0.02)).astype(np.float32)
    cases["float terraces"] = fl
    for name, dem in cases.items():
        nd = dem.dtype.type(-9999)
        exp = orc.port.flat_resolution(dem, nd)
        for sw in (None, "RDGPU_FLAT_Q=0", "RDGPU_FLAT_ASYNC=0", "RDGPU_FLAT_ASYNC=100000"):
            if sw:
                monkeypatch.setenv(*sw.split("="))
            got = rd.barnes_flat_resolution_d8(dem, nd)
            if sw:
                monkeypatch.delenv(sw.split("=")[0])
            assert np.array_equal(got, exp), (name, sw, int((got != exp).sum()))


def test_device_entries_on_a_side_stream_of_the_caller(rd, orc):
    """The _dev_ entries of flat resolution, ResolveFlatsEpsilon and FA_D8 fork internal side streams from the CALLER's
    stream and join them again: on a non-default torch stream, with work queued before and after, the results are the
    oracle's."""
    import torch

    dem = orc.port.fill(fractal_dem(900, 700, seed=75) * np.float32(0.03)).astype(np.float32)
    dem = np.floor(dem).astype(np.float32)
    nd = np.float32(-9999)
    edirs = orc.port.flat_resolution(dem, nd)
    eeps = orc.port.resolve_flats_epsilon(dem, nd)
    efa = orc.port.fa_d8(eeps, nd)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        z = torch.from_numpy(dem).cuda(non_blocking=True)
        z2 = z * 1.0                                             # (queued work the calls must come after)
        dirs = torch.empty(z.shape, dtype=torch.uint8, device="cuda")
        rd.d8_flow_directions_dev(z2, nd, dirs, flats=True)
        e = z2.clone()
        rd.resolve_flats_epsilon_dev(e, nd)
        acc = torch.ones(z.shape, dtype=torch.float64, device="cuda")
        rd.fa_d8_dev(e, nd, acc)
        acc2 = acc + 0.0                                         # (... and work that must come after them)
    st.synchronize()
    assert np.array_equal(dirs.cpu().numpy(), edirs)
    assert e.cpu().numpy().tobytes() == eeps.tobytes()
    assert np.array_equal(acc2.cpu().numpy(), efa)


def test_open_water_tiles(rd, orc):
    """Flats that cover whole 64x64 tiles (the bitmap engine's chamfer path for tiles in which every cell takes part):
    a lake floor with outlets on different sides, with and without islands next to the open tiles, tile-aligned and not,
    and one deep enough that the levels of a tile straddle several flushes of the general path around it."""
    rng = np.random.default_rng(5)
    for (h, w, outlets, islands) in [(300, 400, [(150, 0)], 0), (330, 290, [(0, 100), (329, 200), (100, 289)], 6),
                                     (200, 700, [(199, 650)], 3), (520, 530, [(260, 0), (0, 265)], 12)]:
        dem = np.full((h, w), 50, np.int32)
        dem[3:-3, 3:-3] = 7                                  # the lake floor: NO_FLOW cells only, many full tiles
        for _ in range(islands):
            y, x = int(rng.integers(10, h - 20)), int(rng.integers(10, w - 20))
            dem[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = 60
        for (y, x) in outlets:                               # notches in the rim: the low edges
            if y in (0, h - 1):
                dem[0 if y == 0 else h - 3:3 if y == 0 else h, x] = np.arange(3, 0, -1) if y != 0 else np.arange(1, 4)
            else:
                dem[y, 0 if x == 0 else w - 3:3 if x == 0 else w] = np.arange(1, 4) if x == 0 else np.arange(3, 0, -1)
        check(rd, orc, dem, np.int32(-1))
        got = rd.barnes_flat_resolution_d8(dem, np.int32(-1))
        assert (got[4:-4, 4:-4][dem[4:-4, 4:-4] == 7] != 0).all()   # the whole floor drains


def test_sources_on_tile_edges(rd, orc):
    """Flats whose low/high edges sit exactly on 64x32 tile borders (the relaxation must wake the
    neighbouring tile of a source)."""
    for (h, w, x0, y0) in [(70, 140, 63, 31), (70, 140, 64, 32), (100, 200, 127, 63), (40, 70, 0, 0)]:
        dem = np.full((h, w), 10, np.int32)
        dem[y0:y0 + 3, :] = 5          # a horizontal flat band starting on a tile row boundary
        dem[:, x0:x0 + 2] = 5          # and a vertical one on a tile column boundary
        dem[y0 + 1, 0] = 1             # outlets
        dem[0, x0] = 1
        check(rd, orc, dem, np.int32(-1))
        check(rd, orc, dem.T.copy(), np.int32(-1))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_alter_true(rd, orc, dtype):
    """barnes_flat_resolution_d8(alter=true): the DEM is raised by flat_mask nextafterf steps inside drainable
    flats (d8_flats_alter_dem, flat_resolution.hpp:545-582) -- bit-exact, including negative values."""
    base = np.floor(fractal_dem(400, 300, seed=91) * 0.05).astype(np.float32)
    for shift in (0.0, -40.0, -1e-38):
        dem = (orc.port.fill(base) + np.float32(shift)).astype(dtype)
        edem, edirs = orc.port.flat_resolution_alter(dem, dtype(-9999))
        got = dem.copy()
        dirs = rd.barnes_flat_resolution_d8(got, dtype(-9999), alter=True)
        assert got.tobytes() == edem.tobytes(), (dtype, shift)
        assert np.array_equal(dirs, edirs), (dtype, shift)
        assert (got != dem).any()
    with pytest.raises(rd.RdgpuError):
        rd.barnes_flat_resolution_d8(np.zeros((5, 5), np.int32)[:, ::2], -1, alter=True)   # not contiguous: cannot alter in place


@pytest.mark.parametrize("dtype,offset", [(np.int32, -20), (np.int16, -20), (np.int8, -60), (np.uint8, 0), (np.uint16, 0),
                                          (np.int64, -20), (np.int32, 1 << 26), (np.uint32, (1 << 31) + 5),
                                          (np.int64, -(1 << 40)), (np.uint64, 1 << 50)])
def test_alter_true_integer_dems(rd, orc, dtype, offset):
    """Integer element types: the reference's step is (T)nextafterf((float)e, numeric_limits<T>::infinity() == 0): one
    towards zero per flat_mask increment, one FLOAT spacing for magnitudes of 2^24 and more -- bit-exact (the restatement
    is pinned to the compiled reference on these inputs in test_oracle_pinning.py)."""
    z = fractal_dem(260, 190, seed=92)
    dem = (np.floor((z - z.min()) * 0.05).astype(np.int64) + offset).astype(dtype)
    nd = dtype(0) if np.dtype(dtype).kind == "u" else dtype(-128 if dtype is np.int8 else -9999)
    if np.dtype(dtype).itemsize <= 4:
        dem = orc.port.fill(dem)
    edem, edirs = orc.port.flat_resolution_alter(dem, nd)
    got = dem.copy()
    dirs = rd.barnes_flat_resolution_d8(got, nd, alter=True)
    assert got.tobytes() == edem.tobytes() and np.array_equal(dirs, edirs)
    assert (got != dem).any()


@pytest.mark.parametrize("n,seed", [(10000, 2), (40000, 3)], ids=["10k", "config4_40k"])
def test_full_size_chain_invariants(rd, n, seed):
    """BASELINE config sizes are too big for the oracle in seconds; the whole chain is checked through
    size-independent invariants at 10000 x 10000 and at the bench size 40000 x 40000 on the device-resident path: after fill + flat resolution no
    interior cell is left without a direction, every cell drains off the raster exactly once
    (sum of the border cells' accumulation == number of cells -- this fails for any cycle, dead end or
    double count), accumulation >= 1 everywhere, and the sharded accumulation equals the single-block one."""
    import torch

    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=seed)
    rd.fill_depressions_dev(Z)
    dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    rd.d8_flow_directions_dev(Z, -9999.0, dirs, flats=True)
    assert bool((dirs[1:-1, 1:-1] != 0).all()) and bool((dirs != 255).all())
    area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    rd.d8_flow_accum_dev(dirs, area)
    torch.cuda.synchronize()
    assert bool((area >= 1).all())
    border = area[0].sum() + area[-1].sum() + area[1:-1, 0].sum() + area[1:-1, -1].sum()
    assert float(border.item()) == float(n * n)
    a32 = torch.empty((n, n), dtype=torch.int32, device="cuda")
    rd.d8_flow_accum_dev(dirs, a32)
    assert bool((a32.to(torch.float64) == area).all())
    # FA_D8 on the same filled DEM: flats do not flow there, so only an upper bound holds
    acc = torch.ones((n, n), dtype=torch.float64, device="cuda")
    rd.fa_d8_dev(Z, -9999.0, acc)
    assert bool((acc >= 1).all()) and float(acc.max().item()) <= float(area.max().item()) * 4 + n
    # row-block shards on one GPU: flat resolution over 8 blocks, accumulation over 4 -- identical
    from richdem_amd.sharded import GpuAccumShard, flat_resolution_blocks, row_split

    sharded_dirs, exchanges = flat_resolution_blocks(Z, -9999.0, 8)
    assert bool((sharded_dirs == dirs).all()), exchanges
    del sharded_dirs

    world = 4
    blocks = [dirs[a:b] for a, b in row_split(n, world)]
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
        assert rounds < 100000
    parts = []
    for sh, blk in zip(shards, blocks):
        a = torch.empty(blk.shape, dtype=torch.float64, device="cuda")
        sh.finish(a)
        parts.append(a)
    assert bool((torch.cat(parts, 0) == area).all())


def _epsilon_cases(orc):
    rng = np.random.default_rng(12)
    for seed in range(3):
        z = fractal_dem(230, 170, 400 + seed)
        yield "filled_f32", orc.port.fill(z), np.float32(-9999)
        yield "steps_f32", np.floor((z - z.min()) * 0.05).astype(np.float32), np.float32(-9999)
        yield "filled_f64", orc.port.fill(np.floor((z - z.min()) * 0.1).astype(np.float64)), np.float64(-9999)
        zi = np.floor((z - z.min()) * 0.05).astype(np.int32) - 40          # negative and positive levels
        yield "filled_i32", orc.port.fill(zi), np.int32(-9999)
        yield "raw_i16", zi.astype(np.int16), np.int16(-9999)
        yield "u8_nodata_high", np.clip(zi + 40, 0, 254).astype(np.uint8), np.uint8(255)
        yield "u16", np.clip(zi + 40, 0, 60000).astype(np.uint16), np.uint16(0)
        d = orc.port.fill(z).copy(); d[30:35, 40:50] = -9999; d[60, 70] = -9999
        yield "holes", d, np.float32(-9999)
        d2 = orc.port.fill(np.floor((z - z.min()) * 0.05).astype(np.float32)); d2[rng.random(d2.shape) < 0.03] = 1e9
        yield "nodata_above_the_data", d2, np.float32(1e9)
    yield "noise", rng.integers(0, 3, (90, 110)).astype(np.float32), np.float32(-1)
    yield "tiny", np.zeros((3, 3), np.float32), np.float32(-1)
    yield "row", np.zeros((1, 9), np.float64), np.float64(-1)
    tiny = np.full((40, 40), -1e-44, np.float32); tiny[0, 0] = -1                # negative denormals: through -0.0 to +denormals
    yield "denormals", tiny, np.float32(-9999)
    # 64-bit integers beyond 2^53: the reference's step goes through double and moves by the double spacing
    big = np.full((30, 40), (1 << 60) + 12345, np.int64); big[10:20, 10:30] = (1 << 53) + 3; big[0, 0] = -(1 << 62)
    yield "i64_beyond_2^53", big, np.int64(-9999)
    yield "i64_negative_beyond_2^53", -big, np.int64(-9999)
    ubig = np.full((30, 40), (1 << 63) + (1 << 40) + 77, np.uint64); ubig[5:25, 5:35] = (1 << 53) + 5
    yield "u64_beyond_2^63", ubig, np.uint64(0)


def test_resolve_flats_epsilon(rd, orc):
    """rd.ResolveFlats = ResolveFlatsEpsilon (flats/flats.hpp:21-28): bit-exact for every element type, including
    the reference's towards-zero behaviour on integer DEMs and NoData values above the data."""
    for name, dem, nd in _epsilon_cases(orc):
        exp = orc.port.resolve_flats_epsilon(dem, nd)
        got = rd.ResolveFlats(dem, nodata=nd)
        assert got.dtype == dem.dtype and got.tobytes() == exp.tobytes(), (name, int((got != exp).sum()))
    dem = orc.port.fill(fractal_dem(120, 100, 9))
    r = rd.rdarray(dem.copy(), no_data=-9999)
    out = rd.ResolveFlats(r)
    assert type(out) is rd.rdarray and np.array_equal(r, dem) and "ResolveFlats" in out.metadata["PROCESSING_HISTORY"]
    assert np.asarray(out).tobytes() == orc.port.resolve_flats_epsilon(dem, np.float32(-9999)).tobytes()
    assert rd.ResolveFlats(r, in_place=True) is None and np.array_equal(r, out)
    # every flat cell of a filled DEM drains afterwards
    dirs = rd.d8_flow_directions(np.asarray(out), np.float32(-9999))
    assert (dirs[1:-1, 1:-1] != 0).all()


def test_10k_chain_equals_reference_on_every_cell(rd, orc):
    """The whole path at 10000 x 10000 (G(seed=2), filled): flat-resolved directions, D8 accumulation and FA_D8 must
    equal the reference on every cell -- big lakes, long flow paths, ~25 s of CPU for the reference side."""
    import torch

    n = 10000
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=2)
    rd.fill_depressions_dev(Z)
    filled = Z.cpu().numpy()
    R = orc.ref if orc.ref.available else orc.port
    nd = np.float32(-9999)
    dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    rd.d8_flow_directions_dev(Z, -9999.0, dirs, flats=True)
    exp_dirs = R.flat_resolution(filled, nd)
    assert np.array_equal(dirs.cpu().numpy(), exp_dirs)
    area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    rd.d8_flow_accum_dev(dirs, area)
    assert np.array_equal(area.cpu().numpy(), R.d8_flow_accum(exp_dirs, 255, np.float64))
    acc = torch.ones((n, n), dtype=torch.float64, device="cuda")
    rd.fa_d8_dev(Z, -9999.0, acc)
    assert np.array_equal(acc.cpu().numpy(), R.fa_d8(filled, nd))
