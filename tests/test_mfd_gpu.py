"""GPU parity tests: D-infinity directions / proportions (FM_Tarboton), FA_Tarboton and the generic
FlowAccumulation.  Floating point with atan2 / pow from the device libm: compared within north_star's tolerance -- <= 1 ULP (f32) for
angles and proportions, and END TO END (proportions + accumulation) <= 1 ULP after an f32 cast and 1e-12 relative in f64
(the f64 sums differ from the reference's FIFO order in the last bits only) -- not bit for bit."""
import numpy as np
import pytest

from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu


def ulp_diff_f32(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


CASES = [("frac_f32", lambda: fractal_dem(300, 220, 301)),
         ("int_i32", lambda: fractal_dem_int(200, 150, 302, 0.2)),
         ("filled", lambda: None),
         ("f64", lambda: fractal_dem(120, 90, 303).astype(np.float64) * 1.000001),
         ("u8", lambda: np.floor((fractal_dem(100, 80, 304) - 400) * 0.1).clip(0, 255).astype(np.uint8))]


def dems(orc):
    for name, mk in CASES:
        d = mk()
        if d is None:
            d = orc.port.fill(fractal_dem(260, 200, 305))
        yield name, d


def test_dinf_flow_directions(rd, orc):
    for name, dem in dems(orc):
        nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
        if dem.shape[0] > 50:
            dem = dem.copy(); dem[30:34, 40:50] = nd
        got, exp = rd.dinf_flow_directions(dem, nd), orc.port.dinf_flowdirs(dem, nd)
        assert (ulp_diff_f32(got, exp) <= 1).all(), name
        assert np.array_equal(got == -1, exp == -1) and np.array_equal(got == 0, exp == 0), name


def test_fm_tarboton_proportions(rd, orc):
    for name, dem in dems(orc):
        nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
        got, exp = rd.FlowProportions(dem, "Dinf", nodata=nd), orc.port.fm_tarboton(dem, nd)
        # same receivers (sign pattern) everywhere, proportions within 1 ULP
        assert np.array_equal(np.sign(got), np.sign(exp)), name
        assert (ulp_diff_f32(got, exp) <= 1).all(), name


@pytest.mark.parametrize("stack_below", [None, "0", "4000000000"])
def test_fa_tarboton_and_generic_accumulation(rd, orc, monkeypatch, stack_below):
    """stack_below: the list length under which the launches work off their own by-products from a stack in LDS
    (csrc/mfd.hip, r04f; default 2^22) -- never (one launch per generation), always, and the default"""
    if stack_below is not None:
        monkeypatch.setenv("RDGPU_MFD_STACK_BELOW", stack_below)
    for name, dem in dems(orc):
        nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
        exp = orc.port.fa_tarboton(dem, nd)
        got = rd.FlowAccumulation(dem, "Dinf", nodata=nd)
        # END TO END within north_star's bound: <= 1 ULP after an f32 cast (measured r03: 0 ULP on every cell of these
        # cases, 9e-16 relative in f64 -- the proportions come out bit-identical, only the summation order differs;
        # tests/tools/mfd_ulps.py, profiles/r03_mfd_ulps.json)
        assert (ulp_diff_f32(got, exp) <= 1).all(), (name, int(ulp_diff_f32(got, exp).max()))
        assert np.allclose(got, exp, rtol=1e-12, atol=0), name
        assert np.array_equal(got == -1, exp == -1), name
        # the generic engine on EXACTLY the reference's proportions: only the summation order differs
        props = orc.port.fm_tarboton(dem, nd)
        g2, e2 = rd.FlowAccumFromProps(props), orc.port.flow_accumulation(props)
        assert np.allclose(g2, e2, rtol=1e-12, atol=0), name
        assert (ulp_diff_f32(g2, e2) <= 1).all(), name
        # D8 proportions through the generic entry: integer flows, exact
        p8 = orc.port.fm_d8(dem, nd)
        assert np.array_equal(rd.FlowAccumFromProps(p8), orc.port.flow_accumulation(p8)), name
        w = np.random.default_rng(7).integers(0, 4, dem.shape).astype(np.float64)
        assert np.array_equal(rd.FlowAccumFromProps(p8, w), orc.port.flow_accumulation(p8, w)), name


def test_fm_d8_proportions_exact(rd, orc):
    for name, dem in dems(orc):
        nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
        assert np.array_equal(rd.FlowProportions(dem, "D8", nodata=nd), orc.port.fm_d8(dem, nd)), name


def test_generic_accumulation_errors(rd):
    with pytest.raises(rd.RdgpuError, match="same dimensions"):
        rd.FlowAccumFromProps(np.zeros((4, 5, 9), np.float32), np.ones((3, 3)))
    with pytest.raises(rd.RdgpuError):
        rd.FlowAccumFromProps(np.zeros((4, 5), np.float32))


MFD_METHODS = [("Holmgren", 2.0), ("Holmgren", 0.5), ("Holmgren", 8.0), ("Freeman", 1.1), ("Freeman", 4.0), ("Quinn", None),
               ("D4", None)]


def test_fm_holmgren_freeman_quinn_d4_proportions(rd, orc):
    """FM_Holmgren / FM_Freeman / FM_Quinn / FM_D4 (flowmet/*.hpp): pow() comes from the device libm, so
    proportions are held to <= 1 ULP (f32); which slots flow (and the -2/-1/0 markers) must be identical.
    Quinn (pow(x, 1)) and D4 are exact."""
    for name, dem in dems(orc):
        nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
        if dem.shape[0] > 50:
            dem = dem.copy(); dem[30:34, 40:50] = nd
        for method, x in MFD_METHODS:
            got = rd.FlowProportions(dem, method, nodata=nd, exponent=x)
            exp = orc.port.fm_mfd(dem, nd, method, 1.0 if x is None else x)
            assert np.array_equal(got[..., 0], exp[..., 0]), (name, method)
            assert np.array_equal(got > 0, exp > 0) and np.array_equal(got == -1, exp == -1), (name, method)
            if method in ("Quinn", "D4"):
                assert np.array_equal(got, exp), (name, method)
            else:
                assert (ulp_diff_f32(got, exp) <= 1).all(), (name, method, x, int(ulp_diff_f32(got, exp).max()))


def test_fa_holmgren_freeman_quinn_d4(rd, orc):
    for name, dem in dems(orc):
        nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
        for method, x in MFD_METHODS:
            got = rd.FlowAccumulation(dem, method, nodata=nd, exponent=x)
            exp = orc.port.fa_mfd(dem, nd, method, 1.0 if x is None else x)
            assert np.array_equal(got == -1, exp == -1), (name, method)
            if method == "D4":
                assert np.array_equal(got, exp), name
            else:
                u = ulp_diff_f32(got, exp)      # north_star: <= 1 ULP on float32 (measured: 0, see above)
                assert (u <= 1).all(), (name, method, x, int(u.max()))
                assert np.allclose(got, exp, rtol=1e-12, atol=0), (name, method, x, float(np.abs(got / exp - 1).max()))
        w = np.random.default_rng(3).random(dem.shape)
        got, exp = rd.FlowAccumulation(dem, "Quinn", nodata=nd, weights=w), orc.port.fa_mfd(dem, nd, "Quinn", 1.0, w)
        assert np.allclose(got, exp, rtol=1e-9, atol=0), name


def test_method_errors(rd):
    z = np.zeros((5, 5), np.float32)
    for bad in ("Rho8", "FairfieldLeymarieD4", "nope", None):
        with pytest.raises(Exception):
            rd.FlowProportions(rd.rdarray(z, no_data=-1), method=bad)
    with pytest.raises(Exception, match="requires an exponent"):
        rd.FlowAccumulation(rd.rdarray(z, no_data=-1), method="Freeman")
    with pytest.raises(rd.RdgpuError):
        rd.FlowAccumulation(z, "Holmgren")
