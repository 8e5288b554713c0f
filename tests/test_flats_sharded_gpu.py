"""GPU parity tests: flat resolution over row-block shards (rdgpu_flat_shard_*, SURVEY.md section 8e /
config 5) == the single-block result == the oracle.  The reference's own acceptance idea for its distributed
programs is tiling invariance (programs/parallel_priority_flood/test.py:44-118); same here."""
import numpy as np
import pytest

from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu


def run_blocks(dem, nd, world):
    import torch

    from richdem_amd.sharded import flat_resolution_blocks

    t = torch.from_numpy(np.ascontiguousarray(dem)).cuda()
    dirs, ex = flat_resolution_blocks(t, nd, world)
    return dirs.cpu().numpy(), ex


def check(rd, orc, dem, nd, worlds=(1, 2, 3, 5)):
    exp = orc.port.flat_resolution(dem, nd)
    assert np.array_equal(rd.barnes_flat_resolution_d8(dem, nd), exp)
    for world in worlds:
        if world > 1 and dem.shape[0] // world < 2:
            continue
        got, ex = run_blocks(dem, nd, world)
        if not np.array_equal(got, exp):
            bad = np.argwhere(got != exp)
            raise AssertionError(f"world {world}: {len(bad)} dirs differ; first {bad[:6].tolist()} "
                                 f"got {got[tuple(bad[0])]} exp {exp[tuple(bad[0])]} exchanges {ex}")


@pytest.mark.parametrize("shape", [(4, 9), (16, 64), (37, 65), (100, 130), (300, 421)])
@pytest.mark.parametrize("scale", [1.0, 0.1, 0.02])
def test_filled_integer_dems(rd, orc, shape, scale):
    h, w = shape
    dem = orc.port.fill(fractal_dem_int(w, h, seed=5 * h + w, scale=scale))
    check(rd, orc, dem, np.int32(-9999))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.uint8])
def test_dtypes(rd, orc, dtype):
    z = fractal_dem(260, 190, seed=43)
    dem = np.floor((z - z.min()) * 0.08).astype(dtype)
    dem = orc.port.fill(dem) if dtype != np.float64 else orc.port.fill(dem.astype(np.float32)).astype(np.float64)
    nd = dtype(250) if dtype == np.uint8 else dtype(-9999)
    check(rd, orc, dem, nd)


def test_flats_spanning_many_cuts(rd, orc):
    """One flat across every cut, a comb whose teeth make the levels cross the cuts again and again,
    undrainable flats, mesas and NoData next to the cuts."""
    check(rd, orc, np.zeros((40, 50), np.float32), np.float32(-1), worlds=(2, 4, 8, 20))
    comb = np.full((64, 96), 9, np.int32)
    comb[1:-1, 1:-1] = 5
    for x in range(4, 92, 4):            # walls with gaps alternately at the top and at the bottom: a snake
        comb[2:-2, x] = 9
        if (x // 4) % 2:
            comb[1:4, x] = 5
        else:
            comb[-4:-1, x] = 5
    comb[30, 0] = 1                      # the only outlet
    check(rd, orc, comb, np.int32(-1), worlds=(2, 4, 7, 16))
    rng = np.random.default_rng(7)
    check(rd, orc, rng.integers(0, 3, (150, 170)).astype(np.int32), np.int32(-1), worlds=(2, 3, 6))
    bowl = np.full((60, 60), 5, np.int32); bowl[20:40, 20:40] = 0
    check(rd, orc, bowl, np.int32(-1), worlds=(2, 3))
    mesa = np.zeros((60, 60), np.int32); mesa[20:40, 20:40] = 5
    check(rd, orc, mesa, np.int32(-1), worlds=(2, 3))
    holes = orc.port.fill(fractal_dem_int(200, 160, 77, 0.03))
    holes[rng.random(holes.shape) < 0.02] = -9999
    holes[78:82, :] = np.where(rng.random((4, 200)) < 0.5, -9999, holes[78:82, :])
    check(rd, orc, holes, np.int32(-9999), worlds=(2, 4))


def test_unfilled_float(rd, orc):
    check(rd, orc, fractal_dem(200, 150, 52), np.float32(-9999), worlds=(2, 3))
    check(rd, orc, fractal_dem_int(300, 200, 51, 0.05), np.int32(-9999), worlds=(2, 5))


def test_sharded_equals_single_block_2k(rd):
    """2000 x 3000 filled integer DEM (big lakes across the cuts): 8 shards == one block, on the device."""
    import torch

    from richdem_amd.sharded import flat_resolution_blocks

    dem = torch.from_numpy(fractal_dem_int(3000, 2000, 9, 0.05)).cuda()
    rd.fill_depressions_dev(dem)
    exp = torch.empty(dem.shape, dtype=torch.uint8, device="cuda")
    rd.d8_flow_directions_dev(dem, -9999, exp, flats=True)
    got, ex = flat_resolution_blocks(dem, -9999, 8)
    assert bool((got == exp).all()), ex
    assert bool((got[1:-1, 1:-1] != 0).all())


def test_errors(rd):
    import torch

    from richdem_amd.sharded import GpuFlatShard, flat_resolution_blocks

    with pytest.raises(rd.RdgpuError):
        flat_resolution_blocks(torch.zeros((5, 8), dtype=torch.float32, device="cuda"), -1.0, 4)
    sh = GpuFlatShard()
    with pytest.raises(rd.RdgpuError):
        sh.begin(torch.zeros((4, 8), dtype=torch.float32, device="cuda"), -1.0, 1, 0)
    with pytest.raises(rd.RdgpuError):
        sh.begin(torch.zeros((4, 8), dtype=torch.float32, device="cuda"), -1.0, 2, 2)


def test_multi_device_flat_resolution_entry_on_one_gpu(rd, orc, monkeypatch):
    """rdgpu_flat_resolution_d8_multi_<T> (one process, a list of devices): with device 0 listed several times the row
    blocks go through exactly the multi-device code -- a worker thread per device, two ghost rows per cut, the cut rows,
    heights and solved levels copied from device to device (r05; RDGPU_MULTI_HOST_STAGED=1: through the host, r03-r04), the
    flat-height solve on devices[0] -- and the directions equal the single-device call (and the oracle) on every cell;
    RDGPU_DEVICES routes the plain host entry the same way."""
    import ctypes

    from richdem_amd._lib import check, lib
    from richdem_amd.synth import fractal_dem_int

    z = orc.port.fill(fractal_dem_int(420, 380, 55, 0.05).astype(np.float32))      # big flats that cross every cut
    exp = orc.port.flat_resolution(z, np.float32(-9999))
    h, w = z.shape
    for devs in ([0], [0, 0], [0] * 5, [0] * 16):
        out = np.empty((h, w), np.uint8)
        arr = (ctypes.c_int * len(devs))(*devs)
        check(lib().rdgpu_flat_resolution_d8_multi_f32(z.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(-9999), w, h,
                                                       out.ctypes.data_as(ctypes.c_void_p), arr, len(devs)), "multi")
        assert np.array_equal(out, exp), len(devs)
    monkeypatch.setenv("RDGPU_MULTI_HOST_STAGED", "1")
    for devs in ([0] * 5,):
        out = np.empty((h, w), np.uint8)
        arr = (ctypes.c_int * len(devs))(*devs)
        check(lib().rdgpu_flat_resolution_d8_multi_f32(z.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(-9999), w, h,
                                                       out.ctypes.data_as(ctypes.c_void_p), arr, len(devs)), "multi staged")
        assert np.array_equal(out, exp), ("staged", len(devs))
    monkeypatch.delenv("RDGPU_MULTI_HOST_STAGED")
    q = z.astype(np.int32)
    out = np.empty((h, w), np.uint8)
    arr = (ctypes.c_int * 3)(0, 0, 0)
    check(lib().rdgpu_flat_resolution_d8_multi_i32(q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(-9999), w, h,
                                                   out.ctypes.data_as(ctypes.c_void_p), arr, 3), "multi")
    assert np.array_equal(out, orc.port.flat_resolution(q, np.int32(-9999)))
    monkeypatch.setenv("RDGPU_DEVICES", "0,0,0,0")
    assert np.array_equal(rd.barnes_flat_resolution_d8(z, np.float32(-9999)), exp)
    monkeypatch.delenv("RDGPU_DEVICES")
    arr = (ctypes.c_int * 2)(0, 99)
    assert lib().rdgpu_flat_resolution_d8_multi_f32(z.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(-9999), w, h,
                                                    out.ctypes.data_as(ctypes.c_void_p), arr, 2) != 0
