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
