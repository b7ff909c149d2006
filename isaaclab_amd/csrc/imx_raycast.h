// Device-side closest-hit ray query against the 2-D uniform grid over the terrain mesh.
//
// Replaces `wp.mesh_query_ray` as used by raycast_mesh_kernel (reference isaaclab/utils/warp/kernels.py:66).
// Warp walks a BVH; a terrain is a 2.5-D surface and the height-scanner rays are vertical, so a uniform grid
// in xy is the better structure here: a vertical ray needs exactly one cell lookup and ~2 triangle tests, all
// neighbouring rays touch neighbouring cells (L2-friendly), and the build is a counting sort.
//
// Per-triangle test: Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection", JCGT 2(1) 2013 -- fp32 edge
// functions with the fp64 fallback on exact zeros, two-sided, t >= 0.  Same arithmetic as
// oracle/raycast_oracle.c:woop_f32 (compile with -ffp-contract=off on both sides).
//
// Grid membership (host builder in mesh.cpp): with g = (coord - origin) * inv_cell, a triangle spanning
// [g_lo, g_hi] is listed in cells floor(g_lo + tau) .. ceil(g_hi - tau) - 1 (at least one cell), tau = IMX_GRID_TAU.
// A ray at g_r looks at floor(g_r), plus the lower neighbour when frac(g_r) < tau and the upper neighbour when
// frac(g_r) > 1 - tau.  Every triangle whose xy-extent contains the ray is therefore seen, and a height-field
// mesh aligned with the grid costs 2 triangle references per cell instead of 8.
#pragma once

#include "imx_internal.h"

struct WoopRay {
    int kx, ky, kz;
    float Sx, Sy, Sz;
    float ox, oy, oz;
};

IMX_DEV float pick3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

IMX_DEV WoopRay woop_setup(float ox, float oy, float oz, float dx, float dy, float dz) {
    WoopRay r;
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    r.kx = r.kz + 1;
    if (r.kx == 3) r.kx = 0;
    r.ky = r.kx + 1;
    if (r.ky == 3) r.ky = 0;
    const float dkz = pick3(dx, dy, dz, r.kz);
    if (dkz < 0.0f) {
        const int t = r.kx;
        r.kx = r.ky;
        r.ky = t;
    }
    r.Sx = pick3(dx, dy, dz, r.kx) / dkz;
    r.Sy = pick3(dx, dy, dz, r.ky) / dkz;
    r.Sz = 1.0f / dkz;
    r.ox = ox;
    r.oy = oy;
    r.oz = oz;
    return r;
}

// corners a, b, c of one triangle
IMX_DEV bool woop_hit(const WoopRay& r, float ax, float ay, float az, float bx_, float by_, float bz_, float cx_,
                      float cy_, float cz_, float& t) {
    const float a0 = ax - r.ox, a1 = ay - r.oy, a2 = az - r.oz;
    const float b0 = bx_ - r.ox, b1 = by_ - r.oy, b2 = bz_ - r.oz;
    const float c0 = cx_ - r.ox, c1 = cy_ - r.oy, c2 = cz_ - r.oz;
    const float Akz = pick3(a0, a1, a2, r.kz), Bkz = pick3(b0, b1, b2, r.kz), Ckz = pick3(c0, c1, c2, r.kz);
    const float Ax = pick3(a0, a1, a2, r.kx) - r.Sx * Akz, Ay = pick3(a0, a1, a2, r.ky) - r.Sy * Akz;
    const float Bx = pick3(b0, b1, b2, r.kx) - r.Sx * Bkz, By = pick3(b0, b1, b2, r.ky) - r.Sy * Bkz;
    const float Cx = pick3(c0, c1, c2, r.kx) - r.Sx * Ckz, Cy = pick3(c0, c1, c2, r.ky) - r.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) return false;
    t = T * (1.0f / det);
    return true;
}

IMX_DEV void test_record(const WoopRay& r, const float4 q0, const float4 q1, const float4 q2, float& best, int32_t& face) {
    float t;
    if (woop_hit(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, t)) {
        if (t >= 0.0f && (t < best || (face < 0 && t <= best))) {
            best = t;
            face = __float_as_int(q2.y);
        }
    }
}

IMX_DEV void test_tri(const WoopRay& r, float4 a, float4 b, float4 c, int32_t f, float& best, int32_t& face) {
    float t;
    if (woop_hit(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, t)) {
        if (t >= 0.0f && (t < best || (face < 0 && t <= best))) {
            best = t;
            face = f;
        }
    }
}

IMX_DEV void test_cell(const MeshView& m, const WoopRay& r, int ix, int iy, float& best, int32_t& face) {
    if (ix < 0 || iy < 0 || ix >= m.nx || iy >= m.ny) return;
    const int c = imx_cell_index(ix, iy, m.ntx);
    const int32_t d = m.cell_desc[c].x;
    const int kind = d & 3;
    if (kind == IMX_CELL_LATTICE) {
        const float4* p = m.tile_pool + (size_t)(c >> 6) * 81 + ((iy & 7) * 9 + (ix & 7));
        const float4 va = p[0], vd = p[1], vc = p[9], vb = p[10];
        const int32_t f0 = (int32_t)((uint32_t)d >> 2);
        test_tri(r, va, vb, vc, f0, best, face);      // (a, b, c)
        test_tri(r, va, vd, vb, f0 + 1, best, face);  // (a, d, b)
    } else if (kind != IMX_CELL_EMPTY) {  // (the FLAT mark of a descriptor lives in its id0 word, which this path does not read)
        int2 g = make_int2((int)((uint32_t)d >> 8), (int)(((uint32_t)d >> 2) & 63u));
        if (kind == IMX_CELL_GENERAL_IND) g = m.gtab[(uint32_t)d >> 2];
        for (int k = g.x; k < g.x + g.y; k += 2) {
            const int4 rr = m.refs[k >> 1];
            const float4* q0 = m.tri_rec + (size_t)rr.x * 3;
            const float4* q1 = m.tri_rec + (size_t)rr.z * 3;
            test_record(r, q0[0], q0[1], q0[2], best, face);
            test_record(r, q1[0], q1[1], q1[2], best, face);
        }
    }
}

// cell column/row of a coordinate plus the optional snapped neighbour (-1/+1, 0 = none)
IMX_DEV int cell_of(float g, int& nb) {
    const float fl = floorf(g);
    const float fr = g - fl;
    nb = (fr < IMX_GRID_TAU) ? -1 : ((fr > 1.0f - IMX_GRID_TAU) ? 1 : 0);
    return (int)fl;
}

// Register-lean closest hit for a VERTICAL ray (dx = dy = 0): the Woop shear terms vanish (Sx = Sy = -+0, so
// A[kx] - Sx*A[kz] == A[kx] bit for bit) and one 48-byte record is live at a time.  Same results as cast_ray().
// one triangle, vertical ray: kx, ky = (y, x) when dz < 0 (flip), (x, y) otherwise
IMX_DEV void vertical_tri(float ax_, float ay_, float az_, float bx_, float by_, float bz_, float cx_, float cy_, float cz_,
                          int32_t f, float ox, float oy, float oz, bool flip, float Sz, float& best, int32_t& face) {
    const float ax = ax_ - ox, ay = ay_ - oy, bx = bx_ - ox, by = by_ - oy, cx = cx_ - ox, cy = cy_ - oy;
    const float Ax = flip ? ay : ax, Ay = flip ? ax : ay;
    const float Bx = flip ? by : bx, By = flip ? bx : by;
    const float Cx = flip ? cy : cx, Cy = flip ? cx : cy;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return;
    const float det = U + V + W;
    if (det == 0.0f) return;
    float t;
    if (az_ == bz_ && bz_ == cz_) {
        // a horizontal face (box tops, stair treads, platforms): the hit parameter is the height difference itself; the barycentric
        // sum T / det of three equal terms would only add rounding (and a division) to it
        t = Sz * (az_ - oz);
    } else {
        const float Az = Sz * (az_ - oz), Bz = Sz * (bz_ - oz), Cz = Sz * (cz_ - oz);
        const float T = U * Az + V * Bz + W * Cz;
        if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) return;
        t = T * (1.0f / det);
    }
    if (t >= 0.0f && (t < best || (face < 0 && t <= best))) {
        best = t;
        face = f;
    }
}

// the references of a GENERAL / GENERAL_IND cell after its first pair (which the descriptor itself carries)
IMX_DEV void vertical_cell_tail(const MeshView& m, int32_t d, float ox, float oy, float oz, bool flip, float Sz, float dz,
                                float& best, int32_t& face) {
    int2 g = make_int2((int)((uint32_t)d >> 8), (int)(((uint32_t)d >> 2) & 63u));
    if ((d & 3) == IMX_CELL_GENERAL_IND) g = m.gtab[(uint32_t)d >> 2];
    const int end = g.x + g.y;
    for (int k = g.x + 2; k < end; k += 2) {  // the rest of the list: one 16-byte load gives the next two ids
        const int4 rr = m.refs[k >> 1];
        const float4* p = m.tri_rec + (size_t)rr.x * 3;
        const float4* p2 = m.tri_rec + (size_t)rr.z * 3;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
        const float4 r0 = p2[0], r1 = p2[1], r2 = p2[2];
        vertical_tri(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, __float_as_int(q2.y), ox, oy, oz, flip, Sz,
                     best, face);
        vertical_tri(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, __float_as_int(r2.y), ox, oy, oz, flip, Sz,
                     best, face);
        if (flip && face >= 0 && __int_as_float(rr.w) < oz + best * dz) break;
    }
}

IMX_DEV void vertical_cell(const MeshView& m, int ix, int iy, float ox, float oy, float oz, bool flip, float Sz, float dz,
                           bool interior, float& best, int32_t& face) {
    if (ix < 0 || iy < 0 || ix >= m.nx || iy >= m.ny) return;
    const int c = imx_cell_index(ix, iy, m.ntx);
    int4 d4 = m.cell_desc[c];
    const int32_t d = d4.x;
    const int kind = d & 3;
    if (d4.y < 0) {  // FLAT cell (see imx_internal.h): the descriptor answers an interior downward ray
        d4.y &= 0x7FFFFFFF;
        if (interior && flip) {
            const float t = Sz * (__int_as_float(d4.w) - oz);
            if (t >= 0.0f && (t < best || (face < 0 && t <= best))) { best = t; face = d4.y; }
            return;
        }
        d4.w = __float_as_int(__builtin_huge_valf());  // w held the height, not zrest: no early exit on the first pair
    }
    if (kind == IMX_CELL_LATTICE) {  // height-field quad: descriptor -> 4 shared corners of the tile's vertex pool
        const float4* p = m.tile_pool + (size_t)(c >> 6) * 81 + ((iy & 7) * 9 + (ix & 7));
        const float4 va = p[0], vd = p[1], vc = p[9], vb = p[10];
        const int32_t f0 = (int32_t)((uint32_t)d >> 2);
        vertical_tri(va.x, va.y, va.z, vb.x, vb.y, vb.z, vc.x, vc.y, vc.z, f0, ox, oy, oz, flip, Sz, best, face);
        vertical_tri(va.x, va.y, va.z, vd.x, vd.y, vd.z, vb.x, vb.y, vb.z, f0 + 1, ox, oy, oz, flip, Sz, best, face);
    } else if (kind != IMX_CELL_EMPTY) {
        // first pair straight from the descriptor: descriptor -> two shared triangle records (six 16-byte loads)
        {
            const float4* p = m.tri_rec + (size_t)d4.y * 3;
            const float4* p2 = m.tri_rec + (size_t)d4.z * 3;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2];
            const float4 r0 = p2[0], r1 = p2[1], r2 = p2[2];
            vertical_tri(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, __float_as_int(q2.y), ox, oy, oz, flip, Sz,
                         best, face);
            vertical_tri(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, __float_as_int(r2.y), ox, oy, oz, flip, Sz,
                         best, face);
            // references are sorted by descending top; d4.w = highest top among those after this pair.  A downward
            // ray whose hit already lies above all of them is done (box bottoms, lower steps, ... are never loaded).
            if (flip && face >= 0 && __int_as_float(d4.w) < oz + best * dz) return;
        }
        vertical_cell_tail(m, d, ox, oy, oz, flip, Sz, dz, best, face);
    }
}

IMX_DEV bool cast_ray_vertical(const MeshView& m, float ox, float oy, float oz, float dz, float max_dist, float& t_hit,
                               int32_t& face) {
    float best = max_dist;
    face = -1;
    const bool flip = dz < 0.0f;
    const float Sz = 1.0f / dz;
    int nbx, nby;
    const int ix = cell_of((ox - m.x0) * m.inv_cell, nbx);
    const int iy = cell_of((oy - m.y0) * m.inv_cell, nby);
    vertical_cell(m, ix, iy, ox, oy, oz, flip, Sz, dz, (nbx | nby) == 0, best, face);
    if (nbx | nby) {  // within tau of a cell boundary (rare): the neighbouring cells as well
        if (nbx) vertical_cell(m, ix + nbx, iy, ox, oy, oz, flip, Sz, dz, false, best, face);
        if (nby) vertical_cell(m, ix, iy + nby, ox, oy, oz, flip, Sz, dz, false, best, face);
        if (nbx && nby) vertical_cell(m, ix + nbx, iy + nby, ox, oy, oz, flip, Sz, dz, false, best, face);
    }
    if (face < 0) return false;
    t_hit = best;
    return true;
}

// Closest hit with t in [0, max_dist]; returns false on a miss.
IMX_DEV bool cast_ray(const MeshView& m, float ox, float oy, float oz, float dx, float dy, float dz, float max_dist,
                      float& t_hit, int32_t& face) {
    const WoopRay r = woop_setup(ox, oy, oz, dx, dy, dz);
    float best = max_dist;
    face = -1;
    if (dx == 0.0f && dy == 0.0f) {
        // vertical ray: one cell (+ snapped neighbours)
        int nbx, nby;
        const int ix = cell_of((ox - m.x0) * m.inv_cell, nbx);
        const int iy = cell_of((oy - m.y0) * m.inv_cell, nby);
        test_cell(m, r, ix, iy, best, face);
        if (nbx) test_cell(m, r, ix + nbx, iy, best, face);
        if (nby) test_cell(m, r, ix, iy + nby, best, face);
        if (nbx && nby) test_cell(m, r, ix + nbx, iy + nby, best, face);
    } else {
        // general ray: 2-D DDA over the grid; every visited cell tests its 3x3 block (covers the tau-shrunk lists)
        const float gx_lo = m.x0, gy_lo = m.y0, gx_hi = m.x0 + m.nx * m.cell, gy_hi = m.y0 + m.ny * m.cell;
        float t0 = 0.0f, t1 = max_dist;
        if (dx != 0.0f) {
            const float ta = (gx_lo - ox) / dx, tb = (gx_hi - ox) / dx;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        } else if (ox < gx_lo || ox > gx_hi) {
            return false;
        }
        if (dy != 0.0f) {
            const float ta = (gy_lo - oy) / dy, tb = (gy_hi - oy) / dy;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        } else if (oy < gy_lo || oy > gy_hi) {
            return false;
        }
        if (t0 > t1) return false;
        const float px = ox + t0 * dx, py = oy + t0 * dy;
        int ix = min(max((int)floorf((px - m.x0) * m.inv_cell), 0), m.nx - 1);
        int iy = min(max((int)floorf((py - m.y0) * m.inv_cell), 0), m.ny - 1);
        const int sx = dx > 0.0f ? 1 : -1, sy = dy > 0.0f ? 1 : -1;
        const float inf = __builtin_huge_valf();
        float tmx = inf, tmy = inf, tdx = inf, tdy = inf;
        if (dx != 0.0f) {
            const float bx = m.x0 + (ix + (sx > 0 ? 1 : 0)) * m.cell;
            tmx = (bx - ox) / dx;
            tdx = m.cell / fabsf(dx);
        }
        if (dy != 0.0f) {
            const float by = m.y0 + (iy + (sy > 0 ? 1 : 0)) * m.cell;
            tmy = (by - oy) / dy;
            tdy = m.cell / fabsf(dy);
        }
        float t_enter = t0;
        const int max_steps = m.nx + m.ny + 2;
        for (int step = 0; step < max_steps; ++step) {
            if (ix < 0 || iy < 0 || ix >= m.nx || iy >= m.ny) break;
            if (t_enter > t1 || (face >= 0 && best < t_enter)) break;
            for (int jy = -1; jy <= 1; ++jy)
                for (int jx = -1; jx <= 1; ++jx) test_cell(m, r, ix + jx, iy + jy, best, face);
            if (tmx < tmy) {
                t_enter = tmx;
                tmx += tdx;
                ix += sx;
            } else {
                t_enter = tmy;
                tmy += tdy;
                iy += sy;
            }
        }
    }
    if (face < 0) return false;
    t_hit = best;
    return true;
}
