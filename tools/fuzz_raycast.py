"""Random sweep of the ray-cast (TerrainMesh.raycast: the height scanner's grid walk) against the brute-force oracles over ALL triangles
(oracle/raycast_oracle.c: fp32 Woop -- same arithmetic, bit-exact distances expected -- and fp64 Moeller-Trumbore, 1e-5): random terrains
(tile mix by seed, 1..4 x 1..4 tiles, border), cell sizes, vertical rays (a share of them snapped onto lattice lines), upward rays and
slanted rays through the DDA path.  Test infrastructure, run on the GPU box:  python tools/fuzz_raycast.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from isaaclab_amd.env import TerrainMesh
from isaaclab_amd.terrain import make_rough_terrain
from oracle.raycast import raycast_f64, raycast_woop_f32


def one_case(seed: int) -> str:
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    tile, border = float(rng.choice([4.0, 8.0])), float(rng.choice([0.0, 2.0, 5.0]))
    v, t, ext = make_rough_terrain(rows, cols, tile=tile, border=border, seed=int(rng.integers(0, 100000)))
    cell = float(rng.choice([0.1, 0.1, 0.0, 0.37, 0.05, 0.25]))
    mesh = TerrainMesh(v, t, cell)
    hx, hy = 0.5 * rows * tile + border, 0.5 * cols * tile + border
    R = 12000
    # ---- vertical rays down (the height scanner), some snapped onto lattice / cell lines
    starts = np.stack([rng.uniform(-hx - 0.2, hx + 0.2, R), rng.uniform(-hy - 0.2, hy + 0.2, R), rng.uniform(15, 25, R)], 1).astype(np.float32)
    snap = float(rng.choice([0.1, 0.05, 0.25]))
    starts[:1500, 0] = np.round(starts[:1500, 0] / snap) * snap
    starts[800:2500, 1] = np.round(starts[800:2500, 1] / snap) * snap
    dirs = np.tile(np.array([0, 0, -1], np.float32), (R, 1))
    hits, dist, _, face = mesh.raycast(torch.from_numpy(starts).cuda(), torch.from_numpy(dirs).cuda(), 1e6, True, True)
    hits, dist = hits.cpu().numpy(), dist.cpu().numpy()
    h32, t32, _ = raycast_woop_f32(v, t, starts, dirs)
    h64, t64, _ = raycast_f64(v, t, starts, dirs)
    miss = ~np.isfinite(t32)
    assert np.array_equal(~np.isfinite(dist), miss), f"vertical: {int((np.isfinite(dist) != np.isfinite(t32)).sum())} hit / miss disagreements with the fp32 brute force"
    assert np.array_equal(dist[~miss], t32[~miss]), f"vertical: {int((dist[~miss] != t32[~miss]).sum())} distances differ from the fp32 brute force"
    both = ~miss & np.isfinite(t64)
    assert np.abs(hits[both] - h64[both]).max() <= 1e-5, f"vertical vs fp64: {np.abs(hits[both] - h64[both]).max():.2e}"
    # ---- upward rays from below
    up_s = starts[:3000].copy()
    up_s[:, 2] = -8.0
    up_d = np.tile(np.array([0, 0, 1], np.float32), (3000, 1))
    _, du, _, _ = mesh.raycast(torch.from_numpy(up_s).cuda(), torch.from_numpy(up_d).cuda(), 1e6, True, True)
    _, tu, _ = raycast_woop_f32(v, t, up_s, up_d)
    du = du.cpu().numpy()
    assert np.array_equal(np.isfinite(du), np.isfinite(tu)) and np.array_equal(du[np.isfinite(tu)], tu[np.isfinite(tu)]), "upward rays"
    # ---- slanted rays (DDA): exactly the triangle the exhaustive fp32 search finds; fp64 within 1e-5 / cos(incidence)
    R2 = 3000
    s2 = np.stack([rng.uniform(-hx * 0.8, hx * 0.8, R2), rng.uniform(-hy * 0.8, hy * 0.8, R2), rng.uniform(0.5, 3, R2)], 1).astype(np.float32)
    d2 = rng.normal(size=(R2, 3)).astype(np.float32)
    d2[:, 2] = -np.abs(d2[:, 2]) * float(rng.choice([0.3, 1.0, 3.0]))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    md = float(rng.choice([50.0, 5.0]))
    _, dd, _, fd = mesh.raycast(torch.from_numpy(s2).cuda(), torch.from_numpy(d2).cuda(), md, True, True)
    dd = dd.cpu().numpy()
    _, t32s, _ = raycast_woop_f32(v, t, s2, d2, md)
    dis = np.flatnonzero((np.isfinite(dd) != np.isfinite(t32s)) | (np.isfinite(t32s) & (dd != t32s)))
    assert dis.size == 0, f"slanted: {dis.size} rays disagree with the fp32 brute force, e.g. {[(int(i), float(dd[i]), float(t32s[i])) for i in dis[:3]]}"
    _, t64s, f64s = raycast_f64(v, t, s2, d2, md)
    bs = np.isfinite(dd) & np.isfinite(t64s)
    if bs.any():
        tri = v[t[f64s[bs]]]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
        cosi = np.abs((n * d2[bs]).sum(1)) / np.maximum(np.linalg.norm(n, axis=1), 1e-30)
        err = np.abs(dd[bs] - t64s[bs])
        assert (err * np.maximum(cosi, 1e-3)).max() <= 1e-5 * max(1.0, float(t64s[bs].max()) / 10), f"slanted vs fp64: {(err * cosi).max():.2e}"
    flips = int((np.isfinite(dd) != np.isfinite(t64s)).sum())
    return f"{rows}x{cols} tiles of {tile} m border {border} cell {cell} ({len(t)} triangles): vertical hits {int((~miss).sum())}/{R}, slanted hits {int(np.isfinite(dd).sum())}/{R2}, fp64 hit/miss flips {flips}"


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for c in range(first, first + cases):
        try:
            print(f"case {c}: ok   {one_case(c)}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"case {c}: FAIL {str(e)[:400]}", flush=True)
    print(f"{cases - bad} / {cases} cases agree")
    sys.exit(1 if bad else 0)
