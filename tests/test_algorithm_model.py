"""The algorithm behind the HIP fill, as a numpy model (tests/tools/proto_fill.py), against the oracle -- on the CPU.
Pins the claim of DESIGN.md section 3 independently of any kernel: the descent-forest / Boruvka formulation gives the
reference's surface, and contracting the component-pair list recorded by ONE raster pass (earlier cell of every
adjacent pair, lowest pass per pair) gives the same surface as re-reading the raster every round."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

from richdem_amd.synth import fractal_dem, fractal_dem_int


@pytest.mark.parametrize("topo", [8, 4])
def test_model_equals_oracle_with_and_without_the_pair_list(orc, topo):
    from proto_fill import fill_proto

    rng = np.random.default_rng(5)
    dems = [fractal_dem(190, 130, 11), fractal_dem_int(160, 120, 12, 0.05), rng.random((60, 70)).astype(np.float32),
            rng.integers(0, 3, (50, 40)).astype(np.int32)]
    for z in dems:
        exp = orc.port.fill(z, topo)
        raster, info_r = fill_proto(z, topo, pair_list=False)
        listed, info_l = fill_proto(z, topo, pair_list=True)
        assert np.array_equal(raster, exp) and np.array_equal(listed, exp)
        assert info_l["rounds"] == info_r["rounds"] and info_l["basins"] == info_r["basins"]   # same contraction, round by round
        if info_l["rounds"] > 0:
            assert info_l["pair_records"] > 0


def _descent_sequential(k, topo):
    """k_descent's scan (csrc/fill.hip, r01-r04c): first strictly lowest of the neighbours in the order 0,1,2,3,5,6,7,8 of
    the 3 x 3 window (D4: 1,3,5,7); the cell drains there if that key is lower, or equal with the neighbour BEFORE the
    cell in raster order.  Returns (drains, n)."""
    order = (0, 1, 2, 3, 5, 6, 7, 8) if topo == 8 else (1, 3, 5, 7)
    bk, n = k[order[0]], order[0]
    for m in order[1:]:
        if k[m] < bk:
            bk, n = k[m], m
    kc = k[4]
    return (bk < kc) or (bk == kc and n < 4), n


def _descent_row_triples(k, topo):
    """k_descent16's formulation (r04d): per row triple the minimum and the FIRST of (a, b, c) equal to it, shared by the
    cells above and below; the middle row without its centre; the winner = first of (top, middle, bottom) equal to the
    minimum of the three."""
    a0, b0, c0, a1, kc, c1, a2, b2, c2 = k
    if topo == 8:
        mt, pt = min(a0, b0, c0), (0 if a0 == min(a0, b0, c0) else 1 if b0 == min(a0, b0, c0) else 2)
        mb, pb = min(a2, b2, c2), (0 if a2 == min(a2, b2, c2) else 1 if b2 == min(a2, b2, c2) else 2)
    else:
        mt, pt, mb, pb = b0, 1, b2, 1
    midfirst = a1 <= c1
    mm, pm = (a1, 0) if midfirst else (c1, 2)
    bk = min(mt, mm, mb)
    is_t, is_m = mt == bk, mm == bk
    n = pt if is_t else 3 + pm if is_m else 6 + pb
    drains = (bk < kc) or (bk == kc and (is_t or (is_m and midfirst)))
    return drains, n


@pytest.mark.parametrize("topo", [8, 4])
def test_row_triple_minimum_equals_the_sequential_scan(topo):
    """every 3 x 3 neighbourhood over three key values (all tie patterns), and random ones"""
    import itertools

    for k in itertools.product((0, 1, 2), repeat=9):
        assert _descent_row_triples(k, topo) == _descent_sequential(k, topo), k
    rng = np.random.default_rng(3)
    for k in rng.integers(0, 6, (20000, 9)):
        k = tuple(int(v) for v in k)
        assert _descent_row_triples(k, topo) == _descent_sequential(k, topo), k


def test_d8_tie_rule_on_a_carried_flag_equals_the_reference_scan():
    """d8_FlowDir (flowmet/d8_flowdirs.hpp:63-71): `flowdir % 2 == 0 && n % 2 == 1` on the running choice, against
    k_flowdirs' form of it (r04d: "the choice so far is a diagonal" carried as a flag) and against its closed form -- the
    first cardinal among the lowest neighbours, else the first diagonal -- over every ternary neighbourhood."""
    import itertools

    for nb in itertools.product((0, 1, 2), repeat=8):
        for e in (0, 1, 2):
            m, d = e, 0
            for n in range(1, 9):
                v = nb[n - 1]
                if v < m or (v == m and d > 0 and d % 2 == 0 and n % 2 == 1):
                    m, d = v, n
            m2, d2, diag = e, 0, False
            for n in range(1, 9):
                v = nb[n - 1]
                take = (v < m2 or (v == m2 and diag)) if n % 2 == 1 else v < m2
                if take:
                    m2, d2 = v, n
                diag = (diag and not take) if n % 2 == 1 else (diag or take)
            lo = min(nb)
            closed = 0
            if lo < e:
                card = [n for n in (1, 3, 5, 7) if nb[n - 1] == lo]
                closed = card[0] if card else [n for n in (2, 4, 6, 8) if nb[n - 1] == lo][0]
            assert d == d2 == closed, (nb, e)
