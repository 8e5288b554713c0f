#!/usr/bin/env python3
"""Model of PriorityFloodFlowdirs_Barnes2014 (depressions/Barnes2014.hpp:483-555) WITHOUT a priority queue, for DEMs
without equal elevations -- what csrc/pfdirs.hip computes on the GPU.

The reference floods without raising on a stable (z, insertion) heap; a cell's direction points at its first-popped
neighbour.  With distinct elevations the pop order R is the lexicographic order of the sequences
    key(c) = (F_0(c), F_1(c), ..., F_k(c) = z(c))
where F_0 is the plain fill (minimax level from the raster border) and, for a cell still below its level
(F_{j-1}(c) > z(c): "wet"), F_j(c) is the minimax level from the cells through which the flood ENTERS its pocket -- the
wet cells next to the one cell of elevation F_{j-1}(c) -- inside the pocket; a sequence that is a prefix of another
comes first.  So: one fill per nesting level, each on the cells still wet, with walls everywhere else and outlets at
the entry cells; and per cell a shrinking set of candidate neighbours (those with the lowest F_0, among them those with
the lowest F_1, ... -- a candidate whose sequence has ended is the phase cell itself and wins).
Checked against the oracle's restatement (which is pinned to the compiled reference)."""
import heapq

import numpy as np

D8 = [(0, 0), (-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]   # dx, dy of neighbour n = 1..8
INV = [0, 5, 6, 7, 8, 1, 2, 3, 4]


def restricted_fill(z, domain, outlet):
    """minimax level from the outlet cells, through domain cells only (8-connected)"""
    h, w = z.shape
    F = np.full((h, w), np.inf)
    heap = []
    for y, x in zip(*np.nonzero(outlet)):
        F[y, x] = z[y, x]
        heap.append((z[y, x], y, x))
    heapq.heapify(heap)
    done = np.zeros((h, w), bool)
    while heap:
        f, y, x = heapq.heappop(heap)
        if done[y, x]:
            continue
        done[y, x] = True
        for n in range(1, 9):
            nx, ny = x + D8[n][0], y + D8[n][1]
            if nx < 0 or ny < 0 or nx >= w or ny >= h or not domain[ny, nx] or done[ny, nx]:
                continue
            v = max(f, z[ny, nx])
            if v < F[ny, nx]:
                F[ny, nx] = v
                heapq.heappush(heap, (v, ny, nx))
    return F


def pf_flowdirs_model(dem, nodata, want_levels=False):
    z = dem.astype(np.float64)
    h, w = z.shape
    assert np.unique(z).size == z.size, "the model is for DEMs without equal elevations"
    border = np.zeros((h, w), bool)
    border[0, :] = border[-1, :] = border[:, 0] = border[:, -1] = True
    levels = [restricted_fill(z, np.ones((h, w), bool), border)]          # F_0: the plain fill
    while True:
        Fp = levels[-1]
        wet = np.isfinite(Fp) & (Fp > z)
        if not wet.any():
            break
        # entry cells: wet cells next to THE cell whose elevation is their level
        outlet = np.zeros((h, w), bool)
        for n in range(1, 9):
            dx, dy = D8[n]
            ys, xs = np.nonzero(wet)
            ny, nx = ys + dy, xs + dx
            ok = (ny >= 0) & (ny < h) & (nx >= 0) & (nx < w)
            hit = np.zeros(ys.size, bool)
            hit[ok] = z[ny[ok], nx[ok]] == Fp[ys[ok], xs[ok]]
            outlet[ys[hit], xs[hit]] = True
        levels.append(restricted_fill(z, wet, outlet))
    if want_levels:
        return levels
    dirs = np.zeros((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            if border[y, x]:
                continue
            cand = [n for n in range(1, 9) if 0 <= x + D8[n][0] < w and 0 <= y + D8[n][1] < h]
            for k, F in enumerate(levels):
                vals = {n: F[y + D8[n][1], x + D8[n][0]] for n in cand}
                lo = min(vals.values())
                cand = [n for n in cand if vals[n] == lo]
                ended = [n for n in cand if z[y + D8[n][1], x + D8[n][0]] == lo]      # its sequence ends here: the phase cell
                if ended:
                    cand = ended
                    break
                if len(cand) == 1:
                    break
            assert len(cand) == 1, (y, x, cand)
            dirs[y, x] = cand[0]
    # the border cells' fixed directions (:508-528) and NoData cells (:545-548)
    dirs[0, :] = 3; dirs[-1, :] = 7; dirs[:, 0] = 1; dirs[:, -1] = 5
    dirs[0, 0] = 2; dirs[0, -1] = 4; dirs[-1, 0] = 8; dirs[-1, -1] = 6
    inner = np.zeros((h, w), bool); inner[1:-1, 1:-1] = True
    dirs[inner & (dem == nodata)] = 0
    return dirs, len(levels)


def record_parents_from_levels(dem):
    """parent' of every cell in the record tree (the nearest ancestor of greater elevation in the flood's tree of directions),
    read off the nested fill levels alone: key(c) = (F_0(c), ..., z(c)) IS the path root -> c of that tree, so parent'(c) is
    THE cell whose elevation is the level of c's innermost pocket (the last level at which c was still wet); -1: the root.
    What csrc/pfdirs.hip does after an exact flood (k_next_level keeps that level, k_tie_from_levels looks the cell up)."""
    z = dem.astype(np.float64)
    h, w = z.shape
    levels = pf_flowdirs_model(dem, None, want_levels=True)
    last = np.full((h, w), np.nan)
    for F in levels:
        wet = np.isfinite(F) & (F > z)
        last[wet] = F[wet]
    cell_of = {float(v): i for i, v in enumerate(z.ravel())}
    return np.array([-1 if np.isnan(v) else cell_of[float(v)] for v in last.ravel()], np.int64)


def record_parents_by_search(dem, dirs):
    """the same by definition: walk up the tree of directions to the first ancestor of greater elevation"""
    z = dem.astype(np.float64).ravel()
    h, w = dem.shape
    par = np.full(h * w, -1, np.int64)
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            d = int(dirs[y, x])
            par[y * w + x] = (y + D8[d][1]) * w + (x + D8[d][0])
    g = np.full(h * w, -1, np.int64)
    for c in range(h * w):
        a = par[c]
        while a >= 0 and z[a] < z[c]:
            a = par[a]
        g[c] = a
    return g


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import oracle
    from richdem_amd.synth import fractal_dem
    rng = np.random.default_rng(1)
    for t in range(12):
        h, w = rng.integers(8, 90, 2)
        if t % 2:
            dem = rng.permutation(h * w).reshape(h, w).astype(np.float32)
        else:
            dem = fractal_dem(int(w), int(h), seed=100 + t).astype(np.float64)
            dem = dem + rng.random((h, w)) * 1e-6            # break the ties of the value noise
            dem = dem.astype(np.float64)
        if np.unique(dem).size != dem.size:
            continue
        got, nlev = pf_flowdirs_model(dem, -9999.0)
        exp = oracle.port.pf_flowdirs(dem, -9999.0)
        print(t, dem.shape, dem.dtype, "levels", nlev, "mismatches", int((got != exp).sum()))
