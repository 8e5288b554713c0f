"""The command-line apps (apps/*.cpp: the reference's rd_depressions_flood / rd_flow_accumulation / rd_d8_flowdirs /
rd_depressions_mask on the GPU engine, native raster files instead of GDAL ones) end to end: file -> Array2D -> rdgpu:: call -> file."""
import os
import subprocess

import numpy as np
import pytest

from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(app, *args):
    exe = os.path.join(ROOT, "apps", app)
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "apps")], check=True, capture_output=True)
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True)


def test_apps_on_native_files(rd, orc, tmp_path):
    z = fractal_dem(333, 222, 91)
    z[40:44, 50:60] = -9999
    gt = (500.0, 10.0, 0.0, 800.0, 0.0, -10.0)
    src = rd.rdarray(z, no_data=-9999, geotransform=gt)
    dem, out = str(tmp_path / "dem.rd"), str(tmp_path / "out.rd")
    rd.SaveNative(dem, src)

    r = run("rd_depressions_flood", dem, out)
    assert r.returncode == 0, r.stderr
    filled = rd.LoadNative(out, np.float32)
    assert np.array_equal(filled, orc.port.fill(z)) and filled.no_data == -9999 and tuple(filled.geotransform) == gt

    r = run("rd_d8_flowdirs", dem, out)
    assert r.returncode == 0, r.stderr
    dirs = rd.LoadNative(out, np.uint8)
    assert np.array_equal(dirs, orc.port.flat_resolution(orc.port.fill(z), np.float32(-9999))) and dirs.no_data == 255

    r = run("rd_depressions_mask", dem, out)
    assert r.returncode == 0, r.stderr
    mask = rd.LoadNative(out, np.uint8)
    assert np.array_equal(mask, orc.port.pit_mask(z, np.float32(-9999))) and mask.no_data == 3 and tuple(mask.geotransform) == gt
    assert run("rd_depressions_mask", dem, out, "f64").returncode == 1        # pit_mask: 32-bit element types

    filled_path = str(tmp_path / "filled.rd")
    rd.SaveNative(filled_path, filled)
    for alg, exp in ((1, orc.port.fa_d8(np.asarray(filled), np.float32(-9999))),
                     (6, orc.port.fa_tarboton(np.asarray(filled), np.float32(-9999))),
                     (3, orc.port.fa_mfd(np.asarray(filled), np.float32(-9999), "Quinn"))):
        r = run("rd_flow_accumulation", filled_path, out, alg)
        assert r.returncode == 0, r.stderr
        acc = rd.LoadNative(out, np.float64)
        exp = np.where(exp == -1, -1, exp * 100.0)          # accum.scale(cell area = 10 x 10), NoData untouched
        assert acc.no_data == -1 and np.allclose(acc, exp, rtol=2e-6, atol=0), alg
    r = run("rd_flow_accumulation", filled_path, out, 4, 2.0)
    assert r.returncode == 0, r.stderr
    acc = rd.LoadNative(out, np.float64)
    exp = orc.port.fa_mfd(np.asarray(filled), np.float32(-9999), "Holmgren", 2.0)
    assert np.allclose(acc, np.where(exp == -1, -1, exp * 100.0), rtol=2e-6, atol=0)

    zi = fractal_dem_int(120, 90, 92, 0.2)
    rd.SaveNative(dem, rd.rdarray(zi, no_data=-9999))
    r = run("rd_depressions_flood", dem, out, "i32")
    assert r.returncode == 0 and np.array_equal(rd.LoadNative(out, np.int32), orc.port.fill(zi))

    assert run("rd_depressions_flood").returncode != 0                      # usage
    assert run("rd_depressions_flood", str(tmp_path / "missing.rd"), out).returncode == 1
    assert run("rd_flow_accumulation", filled_path, out, 2).returncode != 0   # Rho8: not provided
    assert run("rd_flow_accumulation", filled_path, out, 4).returncode != 0   # Holmgren needs its parameter
