"""Screen-tile sharding helpers (host side of SURVEY.md 8e): which 32x32 tiles a rank renders and how the gathered
compact tile buffers map back to the frame.  The ownership rule itself lives in libart (art_shard_layout); nothing
here touches a device, so the multi-rank plumbing can be exercised with the gloo backend on CPUs."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

TILE = 32


def shard_layout(width, height, shard_count, shard_rank, root_relief=0):
    """-> (tile ids owned by shard_rank in compact-buffer order, padded tile count of the gather)"""
    L = _lib.load()
    owned, padded = C.c_uint32(), C.c_uint32()
    _lib.check(L.art_shard_layout(width, height, shard_count, shard_rank, root_relief, None, 0, C.byref(owned), C.byref(padded)))
    tiles = np.zeros(max(1, owned.value), np.uint32)
    _lib.check(L.art_shard_layout(width, height, shard_count, shard_rank, root_relief, tiles.ctypes.data_as(C.c_void_p), tiles.size, C.byref(owned), C.byref(padded)))
    return tiles[:owned.value], padded.value


def untile_host(gathered, width, height, shard_count, root_relief=0):
    """numpy mirror of the k_untile kernel: gathered [shard_count, padded, 32, 32, C] -> frame [height, width, C]
    (or [shard_count, padded, 32, 32] -> [height, width] for packed B10G11R11 words)"""
    gathered = np.asarray(gathered)
    frame = np.zeros((height, width) + gathered.shape[4:], gathered.dtype)
    tiles_x = (width + TILE - 1) // TILE
    for s in range(shard_count):
        tiles, _ = shard_layout(width, height, shard_count, s, root_relief)
        for j, t in enumerate(tiles):
            tx, ty = int(t) % tiles_x, int(t) // tiles_x
            x0, y0 = tx * TILE, ty * TILE
            w, h = min(TILE, width - x0), min(TILE, height - y0)
            frame[y0:y0 + h, x0:x0 + w] = gathered[s, j, :h, :w]
    return frame


def tile_host(frame, shard_count, shard_rank, root_relief=0):
    """what a rank's compact colour-tile buffer holds for a given full frame: [padded, 32, 32, C] (zero padded)"""
    frame = np.asarray(frame)
    height, width = frame.shape[:2]
    tiles, padded = shard_layout(width, height, shard_count, shard_rank, root_relief)
    out = np.zeros((padded, TILE, TILE) + frame.shape[2:], frame.dtype)
    tiles_x = (width + TILE - 1) // TILE
    for j, t in enumerate(tiles):
        tx, ty = int(t) % tiles_x, int(t) // tiles_x
        x0, y0 = tx * TILE, ty * TILE
        w, h = min(TILE, width - x0), min(TILE, height - y0)
        out[j, :h, :w] = frame[y0:y0 + h, x0:x0 + w]
    return out
