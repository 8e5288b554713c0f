"""CPU: the fixed-point scheme behind PriorityFloodFlowdirs' equal-elevation handling (csrc/pfdirs.hip, k_tie_*), in plain
Python around the oracle's TIE-FREE flood: ranks(z, tau) -> exact flood -> pop ranks = preorder of the record tree -> tau,
until the ranks reproduce themselves, gives the directions of the reference's stable queue on tie-heavy rasters
(tests/tools/proto_tie_order.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
from proto_tie_order import flowdirs_tree_iteration, flowdirs_with_ties  # noqa: E402


def test_fixed_point_of_the_discovery_order_is_the_stable_queues_order(orc):
    rng = np.random.default_rng(5)
    cases = [rng.integers(0, 4, (14, 17)).astype(np.int32), np.zeros((9, 11), np.int32),
             np.where(rng.random((12, 13)) < 0.15, -9999, rng.integers(0, 9, (12, 13))).astype(np.int32)]
    for z in cases:
        ref = orc.port.pf_flowdirs(z, np.int32(-9999))
        assert np.unique(z).size < z.size                      # ties
        got, floods = flowdirs_with_ties(z, lambda rk: orc.port.pf_flowdirs(rk, np.int32(-7777)))
        got = got.copy()
        interior = np.zeros(z.shape, bool)
        interior[1:-1, 1:-1] = True
        got[(z == -9999) & interior] = 0                       # NoData cells carry no direction (:545-548)
        assert np.array_equal(got, ref), (floods, int((got != ref).sum()))
        assert floods >= 2
        # r05, the tree iteration: ONE flood, then (tree of directions, ranks) iterated -- pop order of the queue's walk over the
        # tree, every cell re-pointed at its first-popped neighbour, discovery times, ranks -- to the same directions
        got2, its = flowdirs_tree_iteration(z, lambda rk: orc.port.pf_flowdirs(rk, np.int32(-7777)))
        got2 = got2.copy()
        got2[(z == -9999) & interior] = 0
        assert np.array_equal(got2, ref), (its, int((got2 != ref).sum()))
    # the ranks alone, equal cells in raster order, are NOT the reference's order (what the iteration is for)
    z = cases[0]
    order = np.argsort(z.ravel(), kind="stable")
    rk = np.empty(z.size, np.int32)
    rk[order] = np.arange(z.size, dtype=np.int32)
    assert not np.array_equal(orc.port.pf_flowdirs(rk.reshape(z.shape), np.int32(-7777)), orc.port.pf_flowdirs(z, np.int32(-9999)))
