"""Synthetic rough-terrain mesh for the height-scanner (the reference's terrain *generator* is out of scope;
only the height-field -> triangle-mesh topology is on the path).

``height_field_to_mesh`` restates ``convert_height_field_to_mesh``
(reference ``isaaclab/terrains/height_field/utils.py:79-173``): vertices on an (x-major) grid, optional slope-threshold
vertex snapping, two triangles per cell ``(i0, i3, i1)`` and ``(i0, i2, i3)``.  It is pinned against the reference
function by ``tests/golden/hf_mesh.npz``.
"""

from __future__ import annotations

import numpy as np


def height_field_to_mesh(height_field: np.ndarray, horizontal_scale: float, vertical_scale: float,
                         slope_threshold: float | None = None) -> tuple[np.ndarray, np.ndarray]:
    num_rows, num_cols = height_field.shape
    y = np.linspace(0, (num_cols - 1) * horizontal_scale, num_cols)
    x = np.linspace(0, (num_rows - 1) * horizontal_scale, num_rows)
    yy, xx = np.meshgrid(y, x)
    hf = height_field.copy()
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        move_x = np.zeros((num_rows, num_cols))
        move_y = np.zeros((num_rows, num_cols))
        move_c = np.zeros((num_rows, num_cols))
        move_x[:-1, :] += hf[1:, :] - hf[:-1, :] > thr
        move_x[1:, :] -= hf[:-1, :] - hf[1:, :] > thr
        move_y[:, :-1] += hf[:, 1:] - hf[:, :-1] > thr
        move_y[:, 1:] -= hf[:, :-1] - hf[:, 1:] > thr
        move_c[:-1, :-1] += hf[1:, 1:] - hf[:-1, :-1] > thr
        move_c[1:, 1:] -= hf[:-1, :-1] - hf[1:, 1:] > thr
        xx = xx + (move_x + move_c * (move_x == 0)) * horizontal_scale
        yy = yy + (move_y + move_c * (move_y == 0)) * horizontal_scale
    vertices = np.zeros((num_rows * num_cols, 3), dtype=np.float32)
    vertices[:, 0] = xx.reshape(-1)
    vertices[:, 1] = yy.reshape(-1)
    vertices[:, 2] = hf.reshape(-1) * vertical_scale
    # two triangles per cell, row-major over (i, j)
    i0 = (np.arange(num_rows - 1)[:, None] * num_cols + np.arange(num_cols - 1)[None, :]).reshape(-1)
    i1, i2 = i0 + 1, i0 + num_cols
    i3 = i2 + 1
    triangles = np.empty((2 * i0.size, 3), dtype=np.uint32)
    triangles[0::2] = np.stack([i0, i3, i1], axis=1)
    triangles[1::2] = np.stack([i0, i2, i3], axis=1)
    return vertices, triangles


def _tile_random_uniform(rng, n, lo, hi, step, vs):
    # random_uniform_terrain-like: heights sampled on a coarse grid, quantised to `step`, nearest-upsampled
    coarse = max(2, n // 4)
    levels = np.arange(lo, hi + step, step)
    h = rng.choice(levels, size=(coarse, coarse))
    idx = np.minimum((np.arange(n) * coarse) // n, coarse - 1)
    return np.rint(h[np.ix_(idx, idx)] / vs)


def _tile_pyramid_slope(n, slope, hs, vs, inverted):
    c = (n - 1) / 2.0
    d = np.maximum(np.abs(np.arange(n) - c)[:, None], np.abs(np.arange(n) - c)[None, :])
    platform = 1.0 / hs  # 1 m half-width flat top
    h = np.clip(c - d, 0, c - platform) * hs * slope
    return np.rint((-h if inverted else h) / vs)


def _tile_stairs(n, step_h, step_w, hs, vs, inverted):
    c = (n - 1) / 2.0
    d = np.maximum(np.abs(np.arange(n) - c)[:, None], np.abs(np.arange(n) - c)[None, :])
    k = np.floor((c - d) / (step_w / hs)).clip(0, None)
    k = np.minimum(k, np.floor((c - 1.0 / hs) / (step_w / hs)))
    h = k * step_h
    return np.rint((-h if inverted else h) / vs)


def _tile_boxes(rng, n, hs, vs, height):
    h = np.zeros((n, n))
    for _ in range(20):
        w = rng.integers(int(0.3 / hs), int(1.0 / hs) + 1, size=2)
        p = rng.integers(0, n - w.max(), size=2)
        h[p[0]:p[0] + w[0], p[1]:p[1] + w[1]] = rng.choice([-height, height, 0.5 * height])
    return np.rint(h / vs)


def make_rough_terrain(num_rows: int = 10, num_cols: int = 20, tile: float = 8.0, horizontal_scale: float = 0.1,
                       vertical_scale: float = 0.005, border: float = 20.0, seed: int = 0,
                       slope_threshold: float | None = 0.75):
    """Seeded rough terrain: ``num_rows x num_cols`` tiles (random-uniform, pyramid slopes, stairs, boxes) on one
    global height field, centred at the origin, plus a flat border ring made of 8 large triangles.

    Returns ``(vertices f32[V,3], triangles u32[F,3], half_extent_xy)``; proportions follow the reference's
    ``ROUGH_TERRAINS_CFG`` (``isaaclab/terrains/config/rough.py:12-51``) loosely -- this is synthetic input.
    """
    rng = np.random.default_rng(seed)
    n = int(round(tile / horizontal_scale))
    R, C = num_rows * n + 1, num_cols * n + 1
    hf = np.zeros((R, C))
    kinds = ("uniform", "slope", "slope_inv", "stairs", "stairs_inv", "boxes")
    for r in range(num_rows):
        for c in range(num_cols):
            kind = kinds[(r * num_cols + c) % len(kinds)]
            difficulty = (r + 0.5) / num_rows
            if kind == "uniform":
                t = _tile_random_uniform(rng, n + 1, 0.02, 0.10, 0.02, vertical_scale)
            elif kind in ("slope", "slope_inv"):
                t = _tile_pyramid_slope(n + 1, 0.4 * difficulty, horizontal_scale, vertical_scale, kind == "slope_inv")
            elif kind in ("stairs", "stairs_inv"):
                t = _tile_stairs(n + 1, 0.05 + 0.18 * difficulty, 0.3, horizontal_scale, vertical_scale,
                                 kind == "stairs_inv")
            else:
                t = _tile_boxes(rng, n + 1, horizontal_scale, vertical_scale, 0.05 + 0.15 * difficulty)
            hf[r * n:(r + 1) * n + 1, c * n:(c + 1) * n + 1] = t
    verts, tris = height_field_to_mesh(hf, horizontal_scale, vertical_scale, slope_threshold)
    hx, hy = 0.5 * num_rows * tile, 0.5 * num_cols * tile
    verts[:, 0] -= hx
    verts[:, 1] -= hy
    if border > 0:
        bx, by = hx + border, hy + border
        V0 = len(verts)
        bv = np.array([[-bx, -by, 0], [bx, -by, 0], [bx, by, 0], [-bx, by, 0],
                       [-hx, -hy, 0], [hx, -hy, 0], [hx, hy, 0], [-hx, hy, 0]], dtype=np.float32)
        quads = [(0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
        bt = []
        for a, b, c_, d in quads:
            bt += [(a, b, c_), (a, c_, d)]
        verts = np.concatenate([verts, bv], axis=0)
        tris = np.concatenate([tris, (np.array(bt, dtype=np.uint32) + V0)], axis=0)
    return verts.astype(np.float32), tris.astype(np.uint32), (hx, hy)
