"""Synthetic rough-terrain mesh for the height-scanner (the reference's terrain *generator* is out of scope;
only the height-field -> triangle-mesh topology is on the path).

``height_field_to_mesh`` restates ``convert_height_field_to_mesh``
(reference ``isaaclab/terrains/height_field/utils.py:79-173``): vertices on an (x-major) grid, optional slope-threshold
vertex snapping, two triangles per cell ``(i0, i3, i1)`` and ``(i0, i2, i3)``.  It is pinned against the reference
function by ``tests/golden/hf_mesh.npz``.
"""

from __future__ import annotations

import numpy as np


def _steeper(hf: np.ndarray, di: int, dj: int, thr: float) -> np.ndarray:
    """mask[i, j] = 1 where the sample one step along (di, dj) lies more than ``thr`` ABOVE sample (i, j) (0 at the far border)."""
    out = np.zeros(hf.shape)
    ni, nj = hf.shape
    src = hf[max(di, 0):ni + min(di, 0), max(dj, 0):nj + min(dj, 0)]   # the neighbour
    dst = hf[max(-di, 0):ni + min(-di, 0), max(-dj, 0):nj + min(-dj, 0)]  # the sample itself
    out[max(-di, 0):ni + min(-di, 0), max(-dj, 0):nj + min(-dj, 0)] = (src - dst) > thr
    return out


def height_field_to_mesh(height_field: np.ndarray, horizontal_scale: float, vertical_scale: float,
                         slope_threshold: float | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Height samples on an x-major grid -> vertices + two triangles per cell, ``(i0, i3, i1)`` and ``(i0, i2, i3)`` -- the topology
    of the reference's ``convert_height_field_to_mesh`` (isaaclab/terrains/height_field/utils.py:79-173), which the lattice cells of
    the ray-caster rely on; pinned against that function by ``tests/golden/hf_mesh.npz``.  With a slope threshold a vertex at the foot of
    a step steeper than the threshold slides one grid step towards the higher neighbour (along x, along y, or -- where neither
    moved -- along the diagonal), which turns the steep quad into a vertical wall."""
    num_rows, num_cols = height_field.shape
    x = np.linspace(0, (num_rows - 1) * horizontal_scale, num_rows)
    y = np.linspace(0, (num_cols - 1) * horizontal_scale, num_cols)
    yy, xx = np.meshgrid(y, x)
    hf = height_field.copy()
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        # +1: the next sample is higher (slide forward); -1: the previous one is (slide back)
        shift_x = _steeper(hf, 1, 0, thr) - _steeper(hf, -1, 0, thr)
        shift_y = _steeper(hf, 0, 1, thr) - _steeper(hf, 0, -1, thr)
        shift_d = _steeper(hf, 1, 1, thr) - _steeper(hf, -1, -1, thr)
        xx = xx + np.where(shift_x != 0, shift_x, shift_d) * horizontal_scale
        yy = yy + np.where(shift_y != 0, shift_y, shift_d) * horizontal_scale
    vertices = np.empty((num_rows * num_cols, 3), dtype=np.float32)
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


_BOX_TRIS = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [0, 4, 5], [0, 5, 1], [1, 5, 6], [1, 6, 2],
                      [2, 6, 7], [2, 7, 3], [3, 7, 4], [3, 4, 0]], dtype=np.uint32)


def _box(x0, y0, x1, y1, z0, z1):
    """Axis-aligned box as 8 vertices / 12 triangles (top = vertices 4..7), like trimesh.creation.box."""
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0],
                  [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], dtype=np.float32)
    return v, _BOX_TRIS


def _tile_mesh_stairs(size, step_h, step_w, platform, inverted):
    """MeshPyramidStairsTerrain-like (reference isaaclab/terrains/trimesh/mesh_terrains.py, pyramid_stairs_terrain):
    every step is a square RING of four boxes, the centre a platform box -- each xy location is covered by one box."""
    vs, ts, nv = [], [], 0
    half = 0.5 * size
    n_steps = max(int((half - 0.5 * platform) / step_w), 1)
    for k in range(n_steps + 1):
        r = half - k * step_w
        r_in = half - (k + 1) * step_w if k < n_steps else 0.0
        z_top = -(k * step_h) if inverted else k * step_h
        z_bot = z_top - max(step_h, 0.05) - (n_steps * step_h if inverted else 0.0) * 0.0 - 0.5
        if k == n_steps:  # central platform
            v, t = _box(-r, -r, r, r, z_bot, z_top)
            vs.append(v); ts.append(t + nv); nv += 8
            break
        for (a0, b0, a1, b1) in ((-r, -r, r, -r_in), (-r, r_in, r, r), (-r, -r_in, -r_in, r_in), (r_in, -r_in, r, r_in)):
            v, t = _box(a0, b0, a1, b1, z_bot, z_top)
            vs.append(v); ts.append(t + nv); nv += 8
    return np.concatenate(vs), np.concatenate(ts)


def _tile_mesh_boxes(rng, size, grid_w, h_lo, h_hi, platform):
    """MeshRandomGridTerrain-like: a grid of boxes with random top heights around a flat platform."""
    vs, ts, nv = [], [], 0
    n = int(size / grid_w)
    off = 0.5 * (size - n * grid_w)
    half = 0.5 * size
    for i in range(n):
        for j in range(n):
            x0, y0 = -half + off + i * grid_w, -half + off + j * grid_w
            cx, cy = x0 + 0.5 * grid_w, y0 + 0.5 * grid_w
            h = 0.0 if max(abs(cx), abs(cy)) < 0.5 * platform else float(rng.uniform(-h_hi, h_hi))
            if abs(h) < h_lo and h != 0.0:
                h = np.sign(h) * h_lo
            v, t = _box(x0, y0, x0 + grid_w, y0 + grid_w, -1.0, h)
            vs.append(v); ts.append(t + nv); nv += 8
    # margin ring so the tile is closed
    if off > 1e-6:
        for (a0, b0, a1, b1) in ((-half, -half, half, -half + off), (-half, half - off, half, half),
                                  (-half, -half + off, -half + off, half - off), (half - off, -half + off, half, half - off)):
            v, t = _box(a0, b0, a1, b1, -1.0, 0.0)
            vs.append(v); ts.append(t + nv); nv += 8
    return np.concatenate(vs), np.concatenate(ts)


def make_rough_terrain(num_rows: int = 10, num_cols: int = 20, tile: float = 8.0, horizontal_scale: float = 0.1,
                       vertical_scale: float = 0.005, border: float = 20.0, seed: int = 0,
                       slope_threshold: float | None = 0.75):
    """Seeded rough terrain with the composition of the reference's ``ROUGH_TERRAINS_CFG``
    (isaaclab/terrains/config/rough.py:12-51): per tile, in proportion 2:2:2:2:1:1, mesh pyramid stairs, inverted mesh
    stairs, mesh random-grid boxes (box primitives, 12 triangles each), height-field random-uniform noise, height-field
    pyramid slope and inverted slope (``height_field_to_mesh`` with slope-threshold snapping) -- 40 % height-field tiles
    as in SURVEY.md section 8(d) -- centred at the origin, plus a flat border ring of 8 large triangles.  Difficulty
    grows with the row.  Synthetic input, not a restatement of the reference's generator.

    Returns ``(vertices f32[V,3], triangles u32[F,3], half_extent_xy)``.
    """
    rng = np.random.default_rng(seed)
    n = int(round(tile / horizontal_scale))
    kinds = ("stairs", "stairs_inv", "boxes", "uniform", "stairs", "stairs_inv", "boxes", "uniform", "slope", "slope_inv")
    hx, hy = 0.5 * num_rows * tile, 0.5 * num_cols * tile
    vs, ts, nv = [], [], 0
    for r in range(num_rows):
        for c in range(num_cols):
            kind = kinds[(r * num_cols + c) % len(kinds)]
            difficulty = (r + 0.5) / num_rows
            ox, oy = -hx + (r + 0.5) * tile, -hy + (c + 0.5) * tile  # tile centre
            if kind in ("uniform", "slope", "slope_inv"):
                if kind == "uniform":
                    hf = _tile_random_uniform(rng, n + 1, 0.02, 0.10, 0.02, vertical_scale)
                else:
                    hf = _tile_pyramid_slope(n + 1, 0.4 * difficulty, horizontal_scale, vertical_scale, kind == "slope_inv")
                v, t = height_field_to_mesh(hf, horizontal_scale, vertical_scale, slope_threshold)
                v = v.copy()
                v[:, 0] += ox - 0.5 * tile
                v[:, 1] += oy - 0.5 * tile
            else:
                if kind in ("stairs", "stairs_inv"):
                    v, t = _tile_mesh_stairs(tile, 0.05 + 0.18 * difficulty, 0.3, 2.0, kind == "stairs_inv")
                else:
                    v, t = _tile_mesh_boxes(rng, tile, 0.45, 0.05, 0.05 + 0.15 * difficulty, 2.0)
                v = v.copy()
                v[:, 0] += ox
                v[:, 1] += oy
            vs.append(v.astype(np.float32)); ts.append(t.astype(np.uint32) + np.uint32(nv)); nv += len(v)
    if border > 0:
        bx, by = hx + border, hy + border
        bv = np.array([[-bx, -by, 0], [bx, -by, 0], [bx, by, 0], [-bx, by, 0],
                       [-hx, -hy, 0], [hx, -hy, 0], [hx, hy, 0], [-hx, hy, 0]], dtype=np.float32)
        quads = [(0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
        bt = []
        for a, b, c_, d in quads:
            bt += [(a, b, c_), (a, c_, d)]
        vs.append(bv); ts.append(np.array(bt, dtype=np.uint32) + np.uint32(nv)); nv += 8
    return np.concatenate(vs).astype(np.float32), np.concatenate(ts).astype(np.uint32), (hx, hy)
