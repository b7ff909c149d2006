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
// A vertical ray on a height-field quad (LATTICE cell) or over horizontal faces (QH cell) is answered from the cell's own 32 bytes:
// see vertical_cell below and imx_internal.h.  The Woop test remains for everything else (GENERAL cells, rays on cell boundaries,
// slanted rays).
//
// Grid membership (host builder in mesh.hip): with g = (coord - origin) * inv_cell, a triangle spanning
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
    const int4 b4 = m.cells[2 * (size_t)c + 1];
    const int kind = b4.x & 0xFF;
    if (kind == IMX_CELL_LATTICE) {  // the quad's corners: grid lines x heights
        const int4 a4 = m.cells[2 * (size_t)c];
        const float x0 = m.gx[ix], x1 = m.gx[ix + 1], y0 = m.gy[iy], y1 = m.gy[iy + 1];
        const float4 va = make_float4(x0, y0, __int_as_float(a4.x), 0.0f), vd = make_float4(x1, y0, __int_as_float(a4.y), 0.0f);
        const float4 vc = make_float4(x0, y1, __int_as_float(a4.z), 0.0f), vb = make_float4(x1, y1, __int_as_float(a4.w), 0.0f);
        test_tri(r, va, vb, vc, b4.y, best, face);      // (a, b, c)
        test_tri(r, va, vd, vb, b4.y + 1, best, face);  // (a, d, b)
    } else if (kind != IMX_CELL_EMPTY) {  // QH / GENERAL: the full reference list
        const int2 g = m.cell_list[c];
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

// Two triangle records, requested TOGETHER (six 16-byte loads in flight) and tested one after the other.  Under 1 % of the rays come
// here, but one such lane keeps its whole wave: with one record per round trip the GENERAL cells of the bench terrain (0.5 % of the
// cells, in 15 % of the waves) cost 4 of the kernel's 17 us.  24 VGPRs live here; the kernel stays under 64.
IMX_DEV void vertical_pair(const MeshView& m, int id0, int id1, float ox, float oy, float oz, bool flip, float Sz, float& best, int32_t& face) {
    const float4* p0 = m.tri_rec + (size_t)id0 * 3;
    const float4* p1 = m.tri_rec + (size_t)id1 * 3;
    const float4 q0 = p0[0], q1 = p0[1], q2 = p0[2];
    const float4 r0 = p1[0], r1 = p1[1], r2 = p1[2];
    vertical_tri(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, __float_as_int(q2.y), ox, oy, oz, flip, Sz, best, face);
    vertical_tri(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, __float_as_int(r2.y), ox, oy, oz, flip, Sz, best, face);
}

// references [k0, end) of a cell's list, two per trip (one 16-byte load gives the next two ids); a downward ray stops once its hit
// lies above everything that is left
IMX_DEV void vertical_list(const MeshView& m, int k0, int end, float ox, float oy, float oz, bool flip, float Sz, float dz,
                           float& best, int32_t& face) {
    for (int k = k0; k < end; k += 2) {
        const int4 rr = m.refs[k >> 1];
        vertical_pair(m, rr.x, rr.z, ox, oy, oz, flip, Sz, best, face);
        if (flip && face >= 0 && __int_as_float(rr.w) < oz + best * dz) break;
    }
}

IMX_DEV void take_hit(float t, int32_t f, float& best, int32_t& face) {
    if (t >= 0.0f && (t < best || (face < 0 && t <= best))) {
        best = t;
        face = f;
    }
}

// bit of the neighbour (dx, dy) in a cell's continuity mask (bits 8.. of the kind word, mesh.hip pass 2b)
IMX_DEV int imx_nb_bit(int dx, int dy) { return 8 + (dy + 1) * 3 + (dx + 1); }

// One cell, vertical ray.  `interior`: the ray is not within tau of a cell boundary (then a QH cell answers from its heights).
// `need`: for a ray within tau of a boundary of ITS cell, the continuity bits of the neighbours it would have to visit; when the cell has
// them all, the surface is proven continuous across those boundaries (mesh.hip pass 2b) and the cell's own record answers the ray like an
// interior one.  Returns true when it did -- the caller then skips the neighbours.
// `defer`: when given, a GENERAL cell hit by a downward ray is not evaluated here; its inline references go to *defer for the wave to
// work them off together (cast_ray_vertical_wave).
struct GeneralCell {
    int id0, id1, id2, id3, first, count;  // count == 0: nothing deferred
};
IMX_DEV bool vertical_cell(const MeshView& m, int ix, int iy, float ox, float oy, float oz, bool flip, float Sz, float dz,
                           bool interior, int need, float& best, int32_t& face, GeneralCell* defer = nullptr) {
    if ((unsigned)ix >= (unsigned)m.nx || (unsigned)iy >= (unsigned)m.ny) return false;
    const int c = imx_cell_index(ix, iy, m.ntx);
    // both halves of the cell: one load level.  32-bit byte offset from the (scalar) base -- the builder refuses grids beyond 2^27 cells
    const int4* cp = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(m.cells) + ((uint32_t)c << 5));
    const int4 a4 = cp[0];
    const int4 b4 = cp[1];
    const int kind = b4.x & 0xFF;
    const bool complete = need != 0 && (b4.x & need) == need;
    if (kind == IMX_CELL_LATTICE) {
        // Height-field quad a (x0,y0), d (x1,y0), c (x0,y1), b (x1,y1), split along a-b.  The ray is inside the quad or not by exact
        // comparisons with its coordinate lines (neighbouring quads share them: no gaps, a ray ON a line is in both and gets the
        // same height from either); the hit height is the plane of the triangle on its side of the diagonal, evaluated in the
        // quad's own coordinates u, w in [0, 1] -- no 20 m lever arm between ray origin and surface as in the edge-function form.
        const float x0 = m.gx[ix], x1 = m.gx[ix + 1], y0 = m.gy[iy], y1 = m.gy[iy + 1];
        if (ox >= x0 && ox <= x1 && oy >= y0 && oy <= y1) {
            const float za = __int_as_float(a4.x), zd = __int_as_float(a4.y), zc = __int_as_float(a4.z), zb = __int_as_float(a4.w);
            // (quad coordinates through v_rcp_f32, 1 ulp, instead of two IEEE divisions of ~10 VALU instructions each: u, w in [0, 1] move
            //  by <= 1e-7, the hit height by <= 1e-7 x the quad's height span -- against the 1e-5 bound of the fp64 brute-force tests)
            const float u = (ox - x0) * __builtin_amdgcn_rcpf(x1 - x0), w = (oy - y0) * __builtin_amdgcn_rcpf(y1 - y0);
            const bool upper = w >= u;  // the (a, b, c) side of the diagonal
            const float z = upper ? za + ((zb - zc) * u + (zc - za) * w) : za + ((zd - za) * u + (zb - zd) * w);
            take_hit(Sz * (z - oz), b4.y + (upper ? 0 : 1), best, face);
            return complete;  // inside this quad, and the quads across the near boundaries continue it with the same corner heights
        }
    } else if (kind == IMX_CELL_QH) {
        if ((interior || complete) && flip) {
            // the highest surface is horizontal on either side of the lines x = cx, y = cy; a ray exactly on a line touches both sides
            // and the closest hit is the higher one
            const float cx = __int_as_float(b4.y), cy = __int_as_float(b4.z);
            const float h00 = __int_as_float(a4.x), h10 = __int_as_float(a4.y), h01 = __int_as_float(a4.z), h11 = __int_as_float(a4.w);
            const float lo = oy < cy ? (ox < cx ? h00 : (ox > cx ? h10 : fmaxf(h00, h10))) : -__builtin_huge_valf();
            const float hi = oy > cy ? (ox < cx ? h01 : (ox > cx ? h11 : fmaxf(h01, h11))) : -__builtin_huge_valf();
            const float on = oy == cy ? fmaxf(ox <= cx ? fmaxf(h00, h01) : -__builtin_huge_valf(), ox >= cx ? fmaxf(h10, h11) : -__builtin_huge_valf())
                                      : -__builtin_huge_valf();
            take_hit(Sz * (fmaxf(fmaxf(lo, hi), on) - oz), b4.w, best, face);
            return complete;
        } else {  // a ray on a cell boundary, or an upward one: the full list
            const int2 g = m.cell_list[c];
            vertical_list(m, g.x, g.x + g.y, ox, oy, oz, flip, Sz, dz, best, face);
        }
    } else if (kind == IMX_CELL_GENERAL) {
#ifdef IMX_EXP_NOGENERAL
        return false;
#endif
        // the first two pairs of references ride in the cell itself: {id2, id0, id1, zrest after the first pair} {kind, first, count, id3}
        if (defer && flip) {
            defer->id0 = a4.y; defer->id1 = a4.z; defer->id2 = a4.x; defer->id3 = b4.w; defer->first = b4.y; defer->count = b4.z;
            return false;
        }
        vertical_pair(m, a4.y, a4.z, ox, oy, oz, flip, Sz, best, face);
        // references are sorted by descending top; a4.w = highest top among those after this pair.  A downward
        // ray whose hit already lies above all of them is done (box bottoms, lower steps, ... are never loaded).
        if (b4.z <= 2 || (flip && face >= 0 && __int_as_float(a4.w) < oz + best * dz)) return false;
        vertical_pair(m, a4.x, b4.w, ox, oy, oz, flip, Sz, best, face);
        if (b4.z > 4) vertical_list(m, b4.y + 4, b4.y + b4.z, ox, oy, oz, flip, Sz, dz, best, face);
    }
    return false;
}

// Sz = 1 / dz (the caller's: the observation kernel takes it from the plan, divided once on the host)
IMX_DEV bool cast_ray_vertical(const MeshView& m, float ox, float oy, float oz, float dz, float Sz, float max_dist, float& t_hit,
                               int32_t& face) {
    float best = max_dist;
    face = -1;
    const bool flip = dz < 0.0f;
    int nbx, nby;
    const int ix = cell_of((ox - m.x0) * m.inv_cell, nbx);
    const int iy = cell_of((oy - m.y0) * m.inv_cell, nby);
    int need = 0;
    if (nbx) need |= 1 << imx_nb_bit(nbx, 0);
    if (nby) need |= 1 << imx_nb_bit(0, nby);
    if (nbx && nby) need |= 1 << imx_nb_bit(nbx, nby);
#ifdef IMX_EXP_NONEIGHBOUR
    nbx = nby = 0;
#endif
    const bool settled = vertical_cell(m, ix, iy, ox, oy, oz, flip, Sz, dz, (nbx | nby) == 0, need, best, face);
    if ((nbx | nby) && !settled) {  // within tau of a cell boundary the surface is not known to continue across: the neighbouring cells as well
        if (nbx) vertical_cell(m, ix + nbx, iy, ox, oy, oz, flip, Sz, dz, false, 0, best, face);
        if (nby) vertical_cell(m, ix, iy + nby, ox, oy, oz, flip, Sz, dz, false, 0, best, face);
        if (nbx && nby) vertical_cell(m, ix + nbx, iy + nby, ox, oy, oz, flip, Sz, dz, false, 0, best, face);
    }
    if (face < 0) return false;
    t_hit = best;
    return true;
}

// cast_ray_vertical for a whole wave at once -- ALL 64 lanes must call it together (`active`: this lane has a ray).  Same results.
// The difference is the GENERAL cells: one such lane (0.5 % of the bench terrain's cells, but three of an env's 187 rays on a snapped
// height field) used to keep its wave for two to four extra round trips while 60 lanes idled.  Here the lanes that met a GENERAL cell
// hand their up to four inline triangle references to the wave: lane 4 s + k tests reference k of the s-th such ray (16 rays per
// round), ONE round trip and ONE triangle test for everybody, a quad minimum, and the owner collects its hit.  Longer lists (> 4
// references, 18 % of the GENERAL cells) continue on their own lane as before.
IMX_DEV bool cast_ray_vertical_wave(const MeshView& m, bool active, float ox, float oy, float oz, float dz, float Sz, float max_dist,
                                    float& t_hit) {
    float best = max_dist;
    int32_t face = -1;
    const bool flip = dz < 0.0f;
    int nbx = 0, nby = 0, ix = 0, iy = 0;
    bool settled = true;
    GeneralCell gc;
    gc.id0 = gc.id1 = gc.id2 = gc.id3 = gc.first = gc.count = 0;
    if (active) {
        ix = cell_of((ox - m.x0) * m.inv_cell, nbx);
        iy = cell_of((oy - m.y0) * m.inv_cell, nby);
        int need = 0;
        if (nbx) need |= 1 << imx_nb_bit(nbx, 0);
        if (nby) need |= 1 << imx_nb_bit(0, nby);
        if (nbx && nby) need |= 1 << imx_nb_bit(nbx, nby);
        settled = vertical_cell(m, ix, iy, ox, oy, oz, flip, Sz, dz, (nbx | nby) == 0, need, best, face, &gc);
    }
    const uint64_t mask = __ballot(gc.count > 0);
    if (mask != 0ull) {  // wave-uniform
        const int lane = (int)__lane_id();
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));  // owners below this lane
        const bool owner = gc.count > 0;
        const int total = __popcll(mask);
        const float inf = __builtin_huge_valf();
        float mine = inf;
        for (int base = 0; base < total; base += 16) {
            // the owners of this round post their lane number to lane (rank - base); everybody else posts to lanes >= 16 (ignored)
            const bool in_round = owner && rank >= base && rank < base + 16;
            const int dst = in_round ? rank - base : 16 + (lane % 48);
            const int posted = __builtin_amdgcn_ds_permute(dst << 2, lane);
            const int slot = lane >> 2, k = lane & 3;
            const bool valid = slot < min(16, total - base);
            const int own = valid ? __builtin_amdgcn_ds_bpermute(slot << 2, posted) : lane;
            const int a0 = __builtin_amdgcn_ds_bpermute(own << 2, gc.id0), a1 = __builtin_amdgcn_ds_bpermute(own << 2, gc.id1);
            const int a2 = __builtin_amdgcn_ds_bpermute(own << 2, gc.id2), a3 = __builtin_amdgcn_ds_bpermute(own << 2, gc.id3);
            const float rx = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(ox)));
            const float ry = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(oy)));
            const float rz = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(oz)));
            float th = inf;
            if (valid) {
                const int id = k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : a3));
                const float4* p = m.tri_rec + (size_t)id * 3;
                const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                float b2 = max_dist;
                int32_t f2 = -1;
                vertical_tri(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, __float_as_int(q2.y), rx, ry, rz, flip, Sz, b2, f2);
                if (f2 >= 0) th = b2;
            }
            // minimum over the four lanes of a slot, then the owner fetches it from the slot's first lane
            th = fminf(th, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, __float_as_int(th))));
            th = fminf(th, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 2) << 2, __float_as_int(th))));
            const float got = __int_as_float(__builtin_amdgcn_ds_bpermute(in_round ? (rank - base) << 4 : lane << 2, __float_as_int(th)));
            if (in_round) mine = got;
        }
        if (owner) {
            take_hit(mine, 0, best, face);  // (+inf = no hit: refused by t < best / t <= best against max_dist)
            if (gc.count > 4) vertical_list(m, gc.first + 4, gc.first + gc.count, ox, oy, oz, flip, Sz, dz, best, face);
        }
    }
    if (active && (nbx | nby) && !settled) {  // within tau of a cell boundary the surface is not known to continue across: the neighbouring cells as well
        if (nbx) vertical_cell(m, ix + nbx, iy, ox, oy, oz, flip, Sz, dz, false, 0, best, face);
        if (nby) vertical_cell(m, ix, iy + nby, ox, oy, oz, flip, Sz, dz, false, 0, best, face);
        if (nbx && nby) vertical_cell(m, ix + nbx, iy + nby, ox, oy, oz, flip, Sz, dz, false, 0, best, face);
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
