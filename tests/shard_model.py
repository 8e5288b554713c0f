"""numpy/pure-Python MODEL of the shard-local engine -- test infrastructure only.

Implements the local phase literally the way the reference does (a labelled Priority-Flood from the tile
perimeter, programs/parallel_priority_flood/Zhou2016pf.hpp:142-227, with the spillover graph of
WatershedsMeet :37-62), with the same interface as richdem_amd.sharded.GpuShardEngine, so the exchange
and the host graph solve (product code) can run on CPU ranks under gloo."""
import heapq

import numpy as np

OUT = 0xFFFFFFFF
D8 = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]
D4 = [(0, -1), (-1, 0), (0, 1), (1, 0)]


def to_key(a: np.ndarray) -> np.ndarray:
    """Order-preserving uint32 keys, identical to rdgpu::Key32 (richdem_amd/csrc/common.hpp)."""
    if a.dtype == np.float32:
        b = a.view(np.uint32).copy()
        b[b == 0x80000000] = 0
        neg = (b & 0x80000000) != 0
        return np.where(neg, ~b, b | np.uint32(0x80000000)).astype(np.uint32)
    if a.dtype == np.int32:
        return (a.view(np.uint32) ^ np.uint32(0x80000000)).astype(np.uint32)
    if a.dtype == np.int16:
        return (a.astype(np.int64) + 32768).astype(np.uint32)
    if a.dtype in (np.uint8, np.uint16, np.uint32):
        return a.astype(np.uint32)
    raise TypeError(a.dtype)


def from_key(k: np.ndarray, dtype) -> np.ndarray:
    k = k.astype(np.uint32)
    if dtype == np.float32:
        pos = (k & 0x80000000) != 0
        return np.where(pos, k & np.uint32(0x7FFFFFFF), ~k).astype(np.uint32).view(np.float32)
    if dtype == np.int32:
        return (k ^ np.uint32(0x80000000)).view(np.int32)
    if dtype == np.int16:
        return (k.astype(np.int64) - 32768).astype(np.int16)
    return k.astype(dtype)


class NumpyShardEngine:
    def begin(self, block: np.ndarray, open_top: bool, open_bottom: bool, topology: int):
        h, w = block.shape
        nb = D8 if topology == 8 else D4
        k = to_key(block).astype(np.int64)
        W = k.copy()
        label = np.full((h, w), -1, np.int64)
        heap = []
        for y in range(h):
            for x in range(w):
                border = x == 0 or x == w - 1 or (y == 0 and not open_top) or (y == h - 1 and not open_bottom)
                if border:
                    label[y, x] = OUT
                elif y == 0:
                    label[y, x] = x
                elif y == h - 1:
                    label[y, x] = w + x
                else:
                    continue
                heapq.heappush(heap, (int(k[y, x]), y, x))
        while heap:
            lvl, y, x = heapq.heappop(heap)
            for dy, dx in nb:
                ny, nx = y + dy, x + dx
                if 0 <= ny < h and 0 <= nx < w and label[ny, nx] < 0:
                    label[ny, nx] = label[y, x]
                    W[ny, nx] = max(int(k[ny, nx]), lvl)
                    heapq.heappush(heap, (int(W[ny, nx]), ny, nx))
        edges = {}
        for y in range(h):
            for x in range(w):
                for dy, dx in nb:
                    ny, nx = y + dy, x + dx
                    if 0 <= ny < h and 0 <= nx < w and label[ny, nx] != label[y, x]:
                        a, b = int(label[y, x]), int(label[ny, nx])
                        key = (min(a, b), max(a, b))
                        p = max(int(W[y, x]), int(W[ny, nx]))
                        if p < edges.get(key, 1 << 40):
                            edges[key] = p
        self.block, self.W, self.label = block, W, label
        keys = np.stack([to_key(block[0]), to_key(block[-1])]).astype(np.uint32)
        e = np.array([[a, b, p] for (a, b), p in sorted(edges.items())], np.uint32).reshape(-1, 3)
        return keys, e

    def finish(self, levels: np.ndarray) -> None:
        lv = levels.reshape(-1).astype(np.int64)
        W, label = self.W, self.label
        inner = label != OUT
        W[inner] = np.maximum(W[inner], lv[label[inner]])
        self.block[...] = from_key(W.astype(np.uint32), self.block.dtype)

    def abort(self):
        pass


class NumpyAccumShard:
    """Pure-Python model of rdgpu_accum_shard_* (same packed outbox format), for the gloo tests."""

    CNT1 = 1 << 56
    LOW = (1 << 56) - 1
    DX = [0, -1, -1, 0, 1, 1, 1, 0, -1]
    DY = [0, 0, -1, -1, -1, 0, 1, 1, 1]

    def begin_local(self, dirs, nodata, above, below):
        """rdgpu_accum_shard_begin_local: the donors across the cuts are not counted"""
        self.begin(dirs, nodata, above, below, local=True)

    def links(self):
        """rdgpu_accum_shard_links: ([2, w] int32 links, [1] int64 incomplete cells)"""
        links = np.full((2, self.w), -1, np.int32)
        for k, y0 in ((0, 0), (1, self.h - 1)):
            for x0 in range(self.w):
                x, y = x0, y0
                if int(self.d[y, x]) == self.nd:
                    continue
                for _ in range(self.w * self.h + 1):
                    d = int(self.d[y, x])
                    if d < 1 or d > 8:
                        break
                    nx, ny = x + self.DX[d], y + self.DY[d]
                    if nx < 0 or nx >= self.w:
                        break
                    if ny < 0:
                        if self.above is not None and int(self.above[nx]) != self.nd:
                            links[k, x0] = nx
                        break
                    if ny >= self.h:
                        if self.below is not None and int(self.below[nx]) != self.nd:
                            links[k, x0] = np.int32(np.uint32(0x80000000 | nx).astype(np.int32))
                        break
                    if int(self.d[ny, nx]) == self.nd:
                        break
                    x, y = nx, ny
        pending = np.array([int(((self.d != self.nd) & ~self.done).sum())], np.int64)
        return links, pending

    def add_paths(self, in_top, in_bottom):
        """rdgpu_accum_shard_add_paths"""
        for y0, box in ((0, in_top), (self.h - 1, in_bottom)):
            if box is None:
                continue
            for x0 in range(self.w):
                v = int(box[x0]) & self.LOW
                if v == 0 or int(self.d[y0, x0]) == self.nd:
                    continue
                x, y = x0, y0
                for _ in range(self.w * self.h + 1):
                    self.total[y, x] += v
                    d = int(self.d[y, x])
                    if d < 1 or d > 8:
                        break
                    nx, ny = x + self.DX[d], y + self.DY[d]
                    if nx < 0 or nx >= self.w or ny < 0 or ny >= self.h or int(self.d[ny, nx]) == self.nd:
                        break
                    x, y = nx, ny

    def begin(self, dirs, nodata, above, below, local=False):
        self.d, self.nd, self.above, self.below = dirs, int(nodata), above, below
        h, w = dirs.shape
        self.h, self.w = h, w
        self.total = np.ones((h, w), np.int64)
        self.pending = np.zeros((h, w), np.int64)
        self.done = np.zeros((h, w), bool)
        self.out = np.zeros((2, w), np.int64)

        def dir_at(x, y):
            if x < 0 or x >= w:
                return self.nd
            if y < 0:
                return int(above[x]) if above is not None and not local else self.nd
            if y >= h:
                return int(below[x]) if below is not None and not local else self.nd
            return int(dirs[y, x])

        for y in range(h):
            for x in range(w):
                if dirs[y, x] == self.nd:
                    continue
                for m in range(1, 9):
                    d = dir_at(x + self.DX[m], y + self.DY[m])
                    if d != self.nd and d == (m + 4 if m <= 4 else m - 4):
                        self.pending[y, x] += 1
        sources = [(x, y) for y in range(h) for x in range(w) if dirs[y, x] != self.nd and self.pending[y, x] == 0]
        for x, y in sources:   # fixed before any walk: walks complete other cells (pending -> 0) on the way
            self._walk(x, y)

    def _walk(self, x, y):
        while True:
            self.done[y, x] = True
            v = int(self.total[y, x])
            d = int(self.d[y, x])
            if d < 1 or d > 8:
                return
            nx, ny = x + self.DX[d], y + self.DY[d]
            if nx < 0 or nx >= self.w:
                return
            if ny < 0:
                if self.above is not None and int(self.above[nx]) != self.nd:
                    self.out[0, nx] += v + self.CNT1
                return
            if ny >= self.h:
                if self.below is not None and int(self.below[nx]) != self.nd:
                    self.out[1, nx] += v + self.CNT1
                return
            if int(self.d[ny, nx]) == self.nd:
                return
            self.total[ny, nx] += v
            self.pending[ny, nx] -= 1
            if self.pending[ny, nx] != 0:
                return
            x, y = nx, ny

    def outbox(self):
        o = self.out.copy()
        self.out[:] = 0
        return o

    def inject(self, from_above, from_below):
        for row, box in ((0, from_above), (self.h - 1, from_below)):
            if box is None:
                continue
            for x in range(self.w):
                pk = int(box[x])
                if pk == 0:
                    continue
                k, s = pk >> 56, pk & self.LOW
                self.total[row, x] += s
                self.pending[row, x] -= k
                if self.pending[row, x] == 0:
                    self._walk(x, row)

    def finish(self, area):
        a = np.where(self.done, self.total, self.total - 1)
        a = np.where(self.d == self.nd, -1, a)
        area[...] = a.astype(area.dtype)


class NumpyFlatShard:
    """Model of rdgpu_flat_shard_* (include/rdgpu.h) on torch CPU tensors: same interface as
    richdem_amd.sharded.GpuFlatShard plus ``solve`` (a Python union-find standing in for
    rdgpu_flat_graph_solve_dev), so that flat_resolution_sharded's exchange loop runs on gloo ranks."""

    INF = 0x7F7F7F7F

    def begin(self, ext, nodata, gtop, gbot):
        import oracle
        from scipy import ndimage

        z = ext.numpy()
        self.z, (self.rows, self.w) = z, z.shape
        self.gtop, self.gbot = gtop, gbot
        rows, w = z.shape
        dirs = oracle.port.d8_flowdirs(z, nodata)
        own = np.zeros(rows, bool)
        own[gtop:rows - gbot] = True
        low, high = [], []
        for y in range(rows):
            if not own[y]:
                continue
            for x in range(w):
                d = dirs[y, x]
                if d == 255:
                    continue
                for dy in (-1, 0, 1):
                    hit = False
                    for dx in (-1, 0, 1):
                        ny, nx = y + dy, x + dx
                        if (dx == 0 and dy == 0) or not (0 <= ny < rows and 0 <= nx < w) or dirs[ny, nx] == 255:
                            continue
                        if d != 0 and dirs[ny, nx] == 0 and z[ny, nx] == z[y, x]:
                            low.append((y, x)); hit = True; break
                        if d == 0 and z[y, x] < z[ny, nx]:
                            high.append((y, x)); hit = True; break
                    if hit:
                        break
        self.src = [low, high]
        dirs[~own] = 1
        self.dirs = dirs
        labels = np.zeros((rows, w), np.int32)
        nxt = 0
        for v in np.unique(z):
            lab, k = ndimage.label(z == v, structure=np.ones((3, 3)))
            labels[lab > 0] = lab[lab > 0] + nxt
            nxt += k
        self.labels = labels
        self.fh = np.zeros(nxt + 1, np.int64)
        self.D = [np.full((rows, w), self.INF, np.int64), np.full((rows, w), self.INF, np.int64)]
        self.seeded = [False, False]
        self.elig = (dirs == 0) & own[:, None]

    def relax(self, phase):
        D = self.D[phase]
        if not self.seeded[phase]:
            self.seeded[phase] = True
            for (y, x) in self.src[phase]:
                if phase == 1 and self.D[0][y, x] >= self.INF:
                    continue
                D[y, x] = 1
        rows, w = self.rows, self.w
        lp = np.pad(self.labels, 1, constant_values=-1)
        while True:
            dp = np.pad(D, 1, constant_values=self.INF)
            best = D.copy()
            for dy in (0, 1, 2):
                for dx in (0, 1, 2):
                    if dx == 1 and dy == 1:
                        continue
                    same = lp[dy:dy + rows, dx:dx + w] == self.labels
                    cand = np.where(same, dp[dy:dy + rows, dx:dx + w] + 1, self.INF)
                    best = np.minimum(best, cand)
            new = np.where(self.elig, best, D)
            if np.array_equal(new, D):
                break
            D[...] = new

    def _cut_rows(self):
        return [self.gtop - 1 if self.gtop else -1, self.gtop, self.rows - self.gbot - 1,
                self.rows - self.gbot if self.gbot else -1]

    def boundary(self, phase):
        import torch

        cr = self._cut_rows()
        return torch.from_numpy(np.stack([self.D[phase][cr[1]], self.D[phase][cr[2]]]).astype(np.int32))

    def inject(self, phase, above, below):
        cr = self._cut_rows()
        if above is not None and cr[0] >= 0:
            self.D[phase][cr[0]] = np.minimum(self.D[phase][cr[0]], above.numpy().astype(np.int64))
        if below is not None and cr[3] >= 0:
            self.D[phase][cr[3]] = np.minimum(self.D[phase][cr[3]], below.numpy().astype(np.int64))

    def heights(self):
        import torch

        A = self.D[1]
        reached = A < self.INF
        np.maximum.at(self.fh, self.labels[reached], A[reached])
        w = self.w
        out = np.zeros(8 * w, np.int32)
        first = {}
        cr = self._cut_rows()
        for p in range(4 * w):
            r = cr[p // w]
            if r >= 0:
                first.setdefault(int(self.labels[r, p % w]), p)
        for p in range(4 * w):
            r = cr[p // w]
            if r < 0:
                out[p] = p
            else:
                lab = int(self.labels[r, p % w])
                out[p], out[4 * w + p] = first[lab], self.fh[lab]
        return torch.from_numpy(out)

    def solve(self, gathered):
        import torch

        g = gathered.numpy()
        world, w = g.shape[0], g.shape[1] // 8
        per = 4 * w
        parent = list(range(world * per))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        def unite(a, b):
            a, b = find(a), find(b)
            if a != b:
                parent[max(a, b)] = min(a, b)

        for r in range(world):
            for p in range(per):
                unite(r * per + p, r * per + int(g[r, p]))
        for r in range(world - 1):
            for j in range(2 * w):
                unite(r * per + 2 * w + j, (r + 1) * per + j)
        val = {}
        for r in range(world):
            for p in range(per):
                root = find(r * per + p)
                val[root] = max(val.get(root, 0), int(g[r, per + p]))
        out = np.zeros((world, per), np.int32)
        for r in range(world):
            for p in range(per):
                out[r, p] = val[find(r * per + p)]
        return torch.from_numpy(out)

    def finish(self, heights):
        import ctypes

        import oracle
        import torch

        cr = self._cut_rows()
        hts = heights.numpy()
        for p in range(4 * self.w):
            r = cr[p // self.w]
            if r >= 0:
                lab = self.labels[r, p % self.w]
                self.fh[lab] = max(self.fh[lab], int(hts[p]))
        T, A = self.D[0], self.D[1]
        M = np.where(T < self.INF, np.where(A < self.INF, self.fh[self.labels] - A, 0) + 2 * T, 0).astype(np.int32)
        dirs = np.ascontiguousarray(self.dirs)
        labels = np.ascontiguousarray(self.labels)
        fn = oracle.port.lib.orc_d8_flow_flats_apply
        fn.restype = None
        fn(M.ctypes.data_as(ctypes.c_void_p), labels.ctypes.data_as(ctypes.c_void_p), self.w, self.rows,
           dirs.ctypes.data_as(ctypes.c_void_p))
        return torch.from_numpy(dirs[self.gtop:self.rows - self.gbot].copy())

    def abort(self):
        pass
