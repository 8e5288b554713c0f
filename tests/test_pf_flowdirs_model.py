"""The queue-free formulation of PriorityFloodFlowdirs_Barnes2014 that csrc/pfdirs.hip runs on the GPU, as a numpy model
(tests/tools/proto_pf_flowdirs.py), against the oracle's restatement of the reference's stable-queue sweep -- on the CPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


@pytest.mark.parametrize("case", range(6))
def test_nested_fill_levels_give_the_pop_order(orc, case):
    from proto_pf_flowdirs import pf_flowdirs_model
    from richdem_amd.synth import fractal_dem

    rng = np.random.default_rng(50 + case)
    h, w = (int(v) for v in rng.integers(6, 60, 2))
    if case % 2:
        dem = rng.permutation(h * w).reshape(h, w).astype(np.float32)
    else:
        dem = fractal_dem(w, h, seed=70 + case).astype(np.float64) + rng.random((h, w)) * 1e-6
    assert np.unique(dem).size == dem.size
    got, levels = pf_flowdirs_model(dem, -9999.0)
    assert np.array_equal(got, orc.port.pf_flowdirs(dem, -9999.0)) and levels >= 1


@pytest.mark.parametrize("case", range(4))
def test_the_levels_name_the_record_tree(orc, case):
    """r05: after an exact flood csrc/pfdirs.hip builds the record tree (nearest ancestor of greater elevation in the tree of
    directions) WITHOUT a search: parent'(c) = the cell whose elevation is the level of c's innermost pocket.  The numpy model of
    the levels against the definition, on the oracle's directions."""
    from proto_pf_flowdirs import record_parents_by_search, record_parents_from_levels
    from richdem_amd.synth import fractal_dem

    rng = np.random.default_rng(150 + case)
    h, w = (int(v) for v in rng.integers(6, 48, 2))
    if case % 2:
        dem = rng.permutation(h * w).reshape(h, w).astype(np.float32)
    else:
        dem = fractal_dem(w, h, seed=170 + case).astype(np.float64) + rng.random((h, w)) * 1e-6
    assert np.unique(dem).size == dem.size
    dirs = orc.port.pf_flowdirs(dem, -9999.0)
    assert np.array_equal(record_parents_from_levels(dem), record_parents_by_search(dem, dirs))
