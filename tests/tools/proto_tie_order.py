"""The stable queue's tie order as a fixed point -- the scheme of csrc/pfdirs.hip (k_tie_*) on the CPU, around any exact
TIE-FREE PriorityFloodFlowdirs (here: the oracle's restatement / the compiled reference run on rank rasters).

PriorityFloodFlowdirs_Barnes2014 (depressions/Barnes2014.hpp:483-555) pops equal elevations in order of insertion
(GridCellZk_low_pq, common/grid_cell.hpp:101-122): a cell's tie key is its DISCOVERY time
    tau(c) = (pop rank of the cell that closed c, position of c among that cell's pushes in d8_order),
border cells first, in the order of the set-up loops (:508-519).  For a tie-free raster the pop rank is the PREORDER of the
record tree T' (parent'(c) = nearest ancestor of greater elevation in the tree of directions), children sorted by elevation.
Iterating  ranks(z, tau) -> exact flood -> tau  until the ranks reproduce themselves gives the reference's directions."""
import numpy as np

DX = [0, -1, -1, 0, 1, 1, 1, 0, -1]
DY = [0, 0, -1, -1, -1, 0, 1, 1, 1]
D8_ORDER = [1, 3, 5, 7, 2, 4, 6, 8]
INV = [0, 5, 6, 7, 8, 1, 2, 3, 4]


def border_push_index(w, h):
    """position of every border cell in the reference's initial pushes (:508-519); returns (index raster, count)"""
    idx = np.full((h, w), -1, np.int64)
    k = 0
    for x in range(w):
        for y in (0, h - 1):
            if idx[y, x] < 0:
                idx[y, x] = k
            k += 1
    for y in range(1, h - 1):
        for x in (0, w - 1):
            if idx[y, x] < 0:
                idx[y, x] = k
            k += 1
    return idx, k


def pop_ranks(rk, dirs):
    """(parent cell, pop rank) of a TIE-FREE flood from its directions: preorder of the record tree"""
    h, w = rk.shape
    n = h * w
    r = rk.ravel()
    par = np.full(n, -1, np.int64)
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            d = int(dirs[y, x])
            par[y * w + x] = (y + DY[d]) * w + (x + DX[d])
    g = np.full(n, -1, np.int64)                      # nearest ancestor of greater rank
    for c in range(n):
        a = par[c]
        while a >= 0 and r[a] < r[c]:
            a = par[a]
        g[c] = a
    kids = [[] for _ in range(n + 1)]
    for c in range(n):
        kids[g[c] if g[c] >= 0 else n].append(c)
    for k in kids:
        k.sort(key=lambda c: r[c])
    R = np.zeros(n, np.int64)
    t = 0
    stack = [iter(kids[n])]
    while stack:
        try:
            c = next(stack[-1])
            R[c] = t
            t += 1
            stack.append(iter(kids[c]))
        except StopIteration:
            stack.pop()
    return par, R


def flowdirs_with_ties(z, tie_free_flood, maxit=1000):
    """directions of the stable-queue flood of z (any ties), given `tie_free_flood(rank raster int32) -> directions`;
    returns (directions, floods run)"""
    h, w = z.shape
    n = h * w
    bidx, nb = border_push_index(w, h)
    tau = np.arange(n, dtype=np.int64)
    zr = z.ravel()
    prev = None
    dirs = None
    for it in range(maxit):
        order = np.lexsort((tau, zr))
        rk = np.empty(n, np.int32)
        rk[order] = np.arange(n, dtype=np.int32)
        if prev is not None and np.array_equal(rk, prev):
            return dirs, it
        prev = rk
        dirs = tie_free_flood(rk.reshape(h, w))
        par, R = pop_ranks(rk.reshape(h, w), dirs)
        tau = np.empty(n, np.int64)
        fd = dirs.ravel()
        for c in range(n):
            if par[c] < 0:
                tau[c] = bidx.ravel()[c]
            else:
                tau[c] = nb + R[par[c]] * 8 + D8_ORDER.index(INV[int(fd[c])])
    raise AssertionError("the tie order did not settle")


def flowdirs_tree_iteration(z, tie_free_flood, maxit=1000):
    """The same directions with ONE flood (r05, csrc/pfdirs.hip "the tree iteration"): the state (tree of directions D,
    ranks r) is iterated -- R = the order in which a priority queue walks D under r (pop_ranks), D'(c) = the neighbour of
    least R (the flood pushes a cell when the first of its neighbours pops), discovery times from (D', R), r' from them --
    until neither a direction nor a rank changes.  Returns (directions, iterations)."""
    h, w = z.shape
    n = h * w
    bidx, nb = border_push_index(w, h)
    zr = z.ravel()
    order = np.lexsort((np.arange(n, dtype=np.int64), zr))
    rk = np.empty(n, np.int32)
    rk[order] = np.arange(n, dtype=np.int32)
    dirs = tie_free_flood(rk.reshape(h, w)).copy()
    for it in range(1, maxit + 1):
        par, R = pop_ranks(rk.reshape(h, w), dirs)
        R2 = R.reshape(h, w)
        new = dirs.copy()
        tau = np.empty(n, np.int64)
        for y in range(h):
            for x in range(w):
                c = y * w + x
                if x == 0 or y == 0 or x == w - 1 or y == h - 1:
                    tau[c] = bidx[y, x]
                    continue
                best = min(range(1, 9), key=lambda d: R2[y + DY[d], x + DX[d]])
                new[y, x] = best
                tau[c] = nb + R2[y + DY[best], x + DX[best]] * 8 + D8_ORDER.index(INV[best])
        order = np.lexsort((tau, zr))
        rk2 = np.empty(n, np.int32)
        rk2[order] = np.arange(n, dtype=np.int32)
        if np.array_equal(new, dirs) and np.array_equal(rk2, rk):
            return dirs, it
        dirs, rk = new, rk2
    raise AssertionError("the tree iteration did not settle")


if __name__ == "__main__":
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import oracle
    from richdem_amd.synth import fractal_dem, fractal_dem_int

    oracle.build()
    B = oracle.ref if oracle.ref.available else oracle.port
    rng = np.random.default_rng(9)
    cases = {"6 levels 40x50": rng.integers(0, 6, (40, 50)).astype(np.int32), "flat 20x25": np.zeros((20, 25), np.int32),
             "G_int 80x70 x0.05": fractal_dem_int(80, 70, 32, 0.05), "float 200x150": fractal_dem(200, 150, 3)}
    for name, z in cases.items():
        ref = B.pf_flowdirs(z, z.dtype.type(-9999))
        got, it = flowdirs_with_ties(z, lambda rk: B.pf_flowdirs(rk, np.int32(-9999)))
        print(name, "floods", it, "cells differing from the reference", int((got != ref).sum()), "of", z.size, flush=True)
        got2, it2 = flowdirs_tree_iteration(z, lambda rk: B.pf_flowdirs(rk, np.int32(-9999)))
        print(name, "tree iteration: 1 flood +", it2, "iterations, cells differing", int((got2 != ref).sum()), flush=True)
