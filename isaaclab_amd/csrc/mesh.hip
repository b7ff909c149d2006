// Terrain mesh upload + 2-D uniform-grid build (host side, counting sort).  Replaces convert_to_warp_mesh /
// wp.Mesh's BVH build (reference isaaclab/utils/warp/ops.py:130-145, called once at RayCaster init,
// sensors/ray_caster/ray_caster.py:182-189).  Membership rule and cell encodings: see imx_raycast.h.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

#include "imx_internal.h"

// area of triangle (grid units) clipped to the axis-aligned square [x0,x1] x [y0,y1] (Sutherland-Hodgman + shoelace)
static double clipped_area(const double tri[3][2], double x0, double y0, double x1, double y1) {
    double px[16], py[16], qx[16], qy[16];
    int n = 3;
    for (int i = 0; i < 3; ++i) { px[i] = tri[i][0]; py[i] = tri[i][1]; }
    for (int side = 0; side < 4 && n > 0; ++side) {
        int m = 0;
        auto inside = [&](double x, double y) { return side == 0 ? x >= x0 : side == 1 ? x <= x1 : side == 2 ? y >= y0 : y <= y1; };
        for (int i = 0; i < n; ++i) {
            const double ax = px[i], ay = py[i], bx = px[(i + 1) % n], by = py[(i + 1) % n];
            const bool ia = inside(ax, ay), ib = inside(bx, by);
            if (ia) { qx[m] = ax; qy[m] = ay; ++m; }
            if (ia != ib) {
                double t;
                if (side < 2) { const double xb = side == 0 ? x0 : x1; t = (xb - ax) / (bx - ax); qx[m] = xb; qy[m] = ay + t * (by - ay); }
                else { const double yb = side == 2 ? y0 : y1; t = (yb - ay) / (by - ay); qy[m] = yb; qx[m] = ax + t * (bx - ax); }
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
    }
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += px[i] * py[(i + 1) % n] - px[(i + 1) % n] * py[i];
    return 0.5 * std::fabs(a);
}

static bool point_in_tri(const double tri[3][2], double x, double y, double eps) {
    double s[3];
    for (int e = 0; e < 3; ++e) {
        const double* a = tri[e];
        const double* b = tri[(e + 1) % 3];
        s[e] = (b[0] - a[0]) * (y - a[1]) - (b[1] - a[1]) * (x - a[0]);
    }
    return (s[0] >= -eps && s[1] >= -eps && s[2] >= -eps) || (s[0] <= eps && s[1] <= eps && s[2] <= eps);
}

static inline void cell_range(float lo, float hi, float origin, float inv_cell, int n, int& a, int& b) {
    // same fp32 expression as the device (imx_raycast.h: cell_of)
    const float g_lo = (lo - origin) * inv_cell, g_hi = (hi - origin) * inv_cell;
    const float tau = 0.5f * IMX_GRID_TAU;  // builder shrinks by tau/2, rays snap within tau: strict margin
    a = (int)std::floor(g_lo + tau);
    b = (int)std::ceil(g_hi - tau) - 1;
    if (b < a) {  // thinner than 2*tau (or a vertical wall): keep the cell of the centre, it is reached by snapping
        const int c = (int)std::floor(0.5f * (g_lo + g_hi));
        a = b = c;
    }
    a = std::max(a, 0);
    b = std::min(b, n - 1);
}

extern "C" int imx_mesh_create(const float* verts, int64_t V, const uint32_t* tris, int64_t F, float cell_size,
                               imx_mesh_t** out) {
    IMX_REQUIRE(verts && tris && out, "imx_mesh_create: null argument");
    IMX_REQUIRE(V > 0 && F > 0 && F < (1ll << 29), "imx_mesh_create: empty or oversized mesh (V=%lld F=%lld)",
                (long long)V, (long long)F);
    float xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
    for (int64_t i = 0; i < V; ++i) {
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        IMX_REQUIRE(std::isfinite(x) && std::isfinite(y) && std::isfinite(z), "imx_mesh_create: vertex %lld is not finite",
                    (long long)i);
        xmin = std::min(xmin, x); xmax = std::max(xmax, x);
        ymin = std::min(ymin, y); ymax = std::max(ymax, y);
    }
    for (int64_t f = 0; f < 3 * F; ++f)
        IMX_REQUIRE((int64_t)tris[f] < V, "imx_mesh_create: triangle index %u out of range", tris[f]);
    if (!(cell_size > 0.0f)) {
        // auto: two triangles per cell on average, like a height-field mesh
        const double area = std::max((double)(xmax - xmin) * (double)(ymax - ymin), 1e-12);
        cell_size = (float)std::sqrt(2.0 * area / (double)F);
    }
    const double ex = (double)xmax - xmin, ey = (double)ymax - ymin;
    int64_t nx = (int64_t)std::floor(ex / cell_size) + 1, ny = (int64_t)std::floor(ey / cell_size) + 1;
    while (((nx + 7) / 8) * ((ny + 7) / 8) * 64 > (1ll << 27)) {  // cap the table at 128 M cells
        cell_size *= 2.0f;
        nx = (int64_t)std::floor(ex / cell_size) + 1;
        ny = (int64_t)std::floor(ey / cell_size) + 1;
    }
    const float inv_cell = 1.0f / cell_size;
    const int ntx = (int)((nx + 7) / 8), nty = (int)((ny + 7) / 8);
    const int64_t ntile = (int64_t)ntx * nty;
    const int64_t ncell = ntile * 64;  // 8x8-tiled layout (imx_cell_index)
    IMX_REQUIRE(ncell <= (1ll << 27), "imx_mesh_create: %lld grid cells (the ray path addresses cells by a 32-bit byte offset: at most 2^27); use a larger cell size", (long long)ncell);

    // ---- pass 1: per-triangle cell ranges, per-cell reference counts (CSR), ascending triangle ids per cell
    std::vector<int32_t> start((size_t)ncell + 1, 0);
    std::vector<int32_t> ra((size_t)F * 4);
    // exact pruning inside the bounding-box range: a triangle stays in a cell only if it overlaps the cell grown by tau
    // (separating-axis test on the three edge normals, grid units, double).  A ray in cell k lies inside the grown cell
    // of every cell it looks at (its own and the tau-snapped neighbours), so no hit is lost; the diagonal half of a box
    // top that only touches a cell's bounding box is dropped.  Strips one cell wide (walls, slivers) are kept whole.
    auto overlaps = [&](int64_t f, int ix, int iy) -> bool {
        const int32_t* q = &ra[(size_t)f * 4];
        if (q[0] == q[1] || q[2] == q[3]) return true;
        double g[3][2];
        for (int c = 0; c < 3; ++c) {
            const float* p = verts + 3 * (size_t)tris[3 * f + c];
            g[c][0] = ((double)p[0] - (double)xmin) * (double)inv_cell;
            g[c][1] = ((double)p[1] - (double)ymin) * (double)inv_cell;
        }
        const double h = 0.5 + 1.1 * (double)IMX_GRID_TAU, cx = ix + 0.5, cy = iy + 0.5;
        for (int e = 0; e < 3; ++e) {
            const double* a = g[e];
            const double* b = g[(e + 1) % 3];
            const double* o = g[(e + 2) % 3];
            const double nx_ = -(b[1] - a[1]), ny_ = b[0] - a[0];
            const double da = nx_ * a[0] + ny_ * a[1], dop = nx_ * o[0] + ny_ * o[1];
            const double tmin = std::min(da, dop), tmax = std::max(da, dop);
            const double bc = nx_ * cx + ny_ * cy, br = h * (std::fabs(nx_) + std::fabs(ny_));
            const double slack = 1e-9 * (std::fabs(bc) + br + std::fabs(tmin) + std::fabs(tmax));
            if (tmin > bc + br + slack || tmax < bc - br - slack) return false;
        }
        return true;
    };
    for (int64_t f = 0; f < F; ++f) {
        float lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
        for (int c = 0; c < 3; ++c) {
            const float* p = verts + 3 * (size_t)tris[3 * f + c];
            lox = std::min(lox, p[0]); hix = std::max(hix, p[0]);
            loy = std::min(loy, p[1]); hiy = std::max(hiy, p[1]);
        }
        int ax, bx, ay, by;
        cell_range(lox, hix, xmin, inv_cell, (int)nx, ax, bx);
        cell_range(loy, hiy, ymin, inv_cell, (int)ny, ay, by);
        ra[f * 4 + 0] = ax; ra[f * 4 + 1] = bx; ra[f * 4 + 2] = ay; ra[f * 4 + 3] = by;
        for (int iy = ay; iy <= by; ++iy)
            for (int ix = ax; ix <= bx; ++ix)
                if (overlaps(f, ix, iy)) start[(size_t)imx_cell_index(ix, iy, ntx) + 1]++;
    }
    int32_t max_refs = 0;
    for (int64_t c = 0; c < ncell; ++c) {
        max_refs = std::max(max_refs, start[c + 1]);
        IMX_REQUIRE((int64_t)start[c] + start[c + 1] < (1ll << 31), "imx_mesh_create: too many cell references");
        start[c + 1] += start[c];
    }
    const int64_t nrefs = start[ncell];
    std::vector<int32_t> refs((size_t)std::max<int64_t>(nrefs, 1));
    {
        std::vector<int32_t> cursor(start.begin(), start.end() - 1);
        for (int64_t f = 0; f < F; ++f)
            for (int iy = ra[f * 4 + 2]; iy <= ra[f * 4 + 3]; ++iy)
                for (int ix = ra[f * 4 + 0]; ix <= ra[f * 4 + 1]; ++ix)
                    if (overlaps(f, ix, iy)) refs[cursor[(size_t)imx_cell_index(ix, iy, ntx)]++] = (int32_t)f;
    }

    // ---- pass 2: encode every cell: 32 bytes {a: four floats, b: kind + three words} (imx_internal.h), so that ONE load level answers
    // a vertical ray on almost every cell:
    //   LATTICE  the two triangles (a,b,c) + (a,d,b) of one height-field quad whose corners sit on the grid's coordinate lines
    //            gx[ix], gx[ix+1], gy[iy], gy[iy+1] (exact float compare; the lines are taken from the mesh itself, they are the
    //            vertex coordinates of the height field): a = the four corner heights;
    //   QH       ("quad heights") the highest surface over the cell's interior is horizontal in each of the <= 4 rectangles an
    //            axis-aligned line x = cx and / or y = cy cuts the cell into -- box tops, stair treads, platforms, their edges and
    //            corners, the plateaus of a snapped height field: a = the four heights, b = {kind, cx, cy, a face id};
    //   GENERAL  everything else: a = {id2, id0, id1, zrest} b = {kind, first reference, count, id3} (the first two reference pairs inline).
    // QH and GENERAL cells also keep their full reference list (cell_list -> refs -> 48-byte triangle records, one record per
    // TRIANGLE shared by all cells): slanted rays, upward rays and rays within tau of a cell boundary walk it.
    std::vector<int32_t> cellv((size_t)ncell * 8, 0);
    std::vector<int32_t> clist((size_t)ncell * 2, 0);
    const float qnan = std::nanf("");
    std::vector<float> gxl((size_t)nx + 1, qnan), gyl((size_t)ny + 1, qnan);
    std::vector<int32_t> refl;  // (triangle id, zrest bits) pairs of the non-lattice cells
    std::map<std::vector<int32_t>, uint64_t> dedup;  // reference list -> first index (identical lists are shared)
    // one 48-byte record per TRIANGLE (shared by every cell that references it): ax ay az bx | by bz cx cy | cz face ztop 0
    std::vector<float> recs((size_t)F * 12, 0.0f);
    std::vector<float> ztop((size_t)F);
    for (int64_t f = 0; f < F; ++f) {
        float* rec = &recs[(size_t)f * 12];
        for (int cc = 0; cc < 3; ++cc) memcpy(&rec[cc * 3], verts + 3 * (size_t)tris[3 * (size_t)f + cc], 12);
        const int32_t fid = (int32_t)f;
        memcpy(&rec[9], &fid, 4);
        ztop[f] = std::max(rec[2], std::max(rec[5], rec[8]));
        rec[10] = ztop[f];
        // Axis-aligned vertical walls (three equal x or three equal y) sort LAST: a vertical ray can never hit one --
        // with X the common ray-relative coordinate, U, V, W are fl(b-c), fl(c-a), fl(a-b) of three rounded products
        // a, b, c (fp32 and the fp64 retry alike), which share a sign only when all three vanish (det == 0) -- so the
        // downward early exit must not wait for their tops.  The generic DDA path tests the whole list in any order.
        if ((rec[0] == rec[3] && rec[3] == rec[6]) || (rec[1] == rec[4] && rec[4] == rec[7])) ztop[f] = -INFINITY;
    }
    int64_t n_lattice = 0, n_general = 0, n_flat = 0;
    int64_t why_general[4] = {0, 0, 0, 0};  // statistics (IMX_MESH_STATS): a sloped triangle / two split lines on one axis / a rectangle not covered / other
    std::vector<int64_t> gen_refs(17, 0);
    auto single_cell = [&](int32_t f) {
        return ra[(size_t)f * 4] == ra[(size_t)f * 4 + 1] && ra[(size_t)f * 4 + 2] == ra[(size_t)f * 4 + 3];
    };
    auto grid_x = [&](float x) { return ((double)x - (double)xmin) * (double)inv_cell; };
    auto grid_y = [&](float y) { return ((double)y - (double)ymin) * (double)inv_cell; };
    // a coordinate line of the lattice: the value the mesh uses for grid line i (first come, must lie on the line within 1 % of a cell)
    auto claim = [&](std::vector<float>& line, int i, float v, double g) -> bool {
        if (std::fabs(g - (double)i) > 0.01) return false;
        if (std::isnan(line[i])) { line[i] = v; return true; }
        return line[i] == v;
    };
    for (int iy = 0; iy < (int)ny; ++iy)
        for (int ix = 0; ix < (int)nx; ++ix) {
            const int c = imx_cell_index(ix, iy, ntx);
            const int n = start[c + 1] - start[c];
            if (n == 0) continue;
            const int32_t* r = &refs[start[c]];
            int32_t* ca = &cellv[(size_t)c * 8];
            int32_t* cb = ca + 4;
            if (n == 2 && r[1] == r[0] + 1 && single_cell(r[0]) && single_cell(r[1])) {
                const uint32_t* t0 = tris + 3 * (size_t)r[0];
                const uint32_t* t1 = tris + 3 * (size_t)r[1];
                if (t0[0] == t1[0] && t0[1] == t1[2]) {  // (a,b,c), (a,d,b)
                    const float* va = verts + 3 * (size_t)t0[0];
                    const float* vb = verts + 3 * (size_t)t0[1];
                    const float* vc = verts + 3 * (size_t)t0[2];
                    const float* vd = verts + 3 * (size_t)t1[1];
                    // a -> (ix, iy), d -> (ix+1, iy), c -> (ix, iy+1), b -> (ix+1, iy+1): an axis-aligned rectangle on the grid lines
                    const bool rect = va[0] == vc[0] && vd[0] == vb[0] && va[1] == vd[1] && vc[1] == vb[1] && va[0] < vd[0] && va[1] < vc[1];
                    // (claims are only made when all four would hold, so a rejected quad leaves no line behind)
                    auto would = [&](const std::vector<float>& line, int i, float v, double g) {
                        return std::fabs(g - (double)i) <= 0.01 && (std::isnan(line[i]) || line[i] == v);
                    };
                    if (rect && would(gxl, ix, va[0], grid_x(va[0])) && would(gxl, ix + 1, vd[0], grid_x(vd[0])) &&
                        would(gyl, iy, va[1], grid_y(va[1])) && would(gyl, iy + 1, vc[1], grid_y(vc[1]))) {
                        claim(gxl, ix, va[0], grid_x(va[0])); claim(gxl, ix + 1, vd[0], grid_x(vd[0]));
                        claim(gyl, iy, va[1], grid_y(va[1])); claim(gyl, iy + 1, vc[1], grid_y(vc[1]));
                        memcpy(&ca[0], &va[2], 4); memcpy(&ca[1], &vd[2], 4); memcpy(&ca[2], &vc[2], 4); memcpy(&ca[3], &vb[2], 4);
                        cb[0] = IMX_CELL_LATTICE;
                        cb[1] = r[0];
                        ++n_lattice;
                        continue;
                    }
                }
            }
            // ---- the cell's reference list: sorted by descending top (max z), ties by triangle id; each reference carries zrest = the
            // highest top among the references AFTER it, so a downward ray stops as soon as its hit lies above what is left
            const int npad = (n + 1) & ~1;
            std::vector<std::pair<float, int32_t>> order((size_t)n);
            for (int k = 0; k < n; ++k) order[k] = {ztop[(size_t)r[k]], r[k]};
            std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            std::vector<int32_t> key((size_t)npad);
            for (int k = 0; k < npad; ++k) key[k] = order[std::min(k, n - 1)].second;  // odd counts repeat the last one
            // identical lists are stored once: every interior cell of a box top / stair tread shares its neighbours' list
            uint64_t first;
            auto it = dedup.find(key);
            if (it != dedup.end()) {
                first = it->second;
            } else {
                first = refl.size() / 2;  // in refs (int2 units); always even (16-byte aligned pairs)
                dedup.emplace(key, first);
                const size_t base = refl.size();
                for (int k = 0; k < npad; ++k) {
                    refl.push_back(key[k]);
                    refl.push_back(0);
                }
                float suffix = -INFINITY;
                for (int k = npad - 1; k >= 0; --k) {
                    memcpy(&refl[base + (size_t)k * 2 + 1], &suffix, 4);
                    if (k < n) suffix = std::max(suffix, order[k].first);
                }
            }
            IMX_REQUIRE(first < (1ull << 31), "imx_mesh_create: too many cell references");
            clist[(size_t)c * 2] = (int32_t)first;
            clist[(size_t)c * 2 + 1] = npad;
            ++n_general;
            // ---- QH?  Only horizontal faces and axis-aligned walls in the list; at most one wall / face-edge line x = cx and one
            // line y = cy through the interior; in each rectangle they cut, the faces at the highest height present cover it all.
            bool qh = std::isfinite(order[0].first);
            std::vector<std::array<double, 7>> hf;  // horizontal faces: 3 x (gx, gy), height
            float cxw = INFINITY, cyw = INFINITY;   // split lines (world coordinates, the mesh's own floats)
            double cxg = 0.0, cyg = 0.0;
            const double mg = 0.5 * (double)IMX_GRID_TAU;
            const double sx0 = ix + mg, sx1 = ix + 1 - mg, sy0 = iy + mg, sy1 = iy + 1 - mg;
            for (int k = 0; k < n && qh; ++k) {
                const float* rec = &recs[(size_t)r[k] * 12];
                const bool horizontal = rec[2] == rec[5] && rec[5] == rec[8];
                const bool wall = ztop[(size_t)r[k]] == -INFINITY;
                if (!horizontal && !wall) { qh = false; ++why_general[0]; break; }  // a sloped triangle: the Woop path
                double g[3][2];
                for (int cc = 0; cc < 3; ++cc) { g[cc][0] = grid_x(rec[cc * 3]); g[cc][1] = grid_y(rec[cc * 3 + 1]); }
                if (horizontal) hf.push_back({g[0][0], g[0][1], g[1][0], g[1][1], g[2][0], g[2][1], (double)rec[2]});
                for (int e = 0; e < 3 && qh; ++e) {  // axis-aligned edges that cross the interior are the only allowed height boundaries
                    const int e2 = (e + 1) % 3;
                    const float xa = rec[e * 3], ya = rec[e * 3 + 1], xb = rec[e2 * 3], yb = rec[e2 * 3 + 1];
                    if (xa == xb && g[e][0] > sx0 && g[e][0] < sx1 && std::max(g[e][1], g[e2][1]) > sy0 && std::min(g[e][1], g[e2][1]) < sy1) {
                        if (cxw == INFINITY) { cxw = xa; cxg = g[e][0]; } else if (cxw != xa) { if (qh) ++why_general[1]; qh = false; }
                    }
                    if (ya == yb && g[e][1] > sy0 && g[e][1] < sy1 && std::max(g[e][0], g[e2][0]) > sx0 && std::min(g[e][0], g[e2][0]) < sx1) {
                        if (cyw == INFINITY) { cyw = ya; cyg = g[e][1]; } else if (cyw != ya) { if (qh) ++why_general[1]; qh = false; }
                    }
                }
            }
            float hq[4] = {0, 0, 0, 0};
            int32_t face_q = order[0].second;
            if (qh) {
                const double xs[3] = {sx0, cxw == INFINITY ? sx1 : cxg, sx1}, ys[3] = {sy0, cyw == INFINITY ? sy1 : cyg, sy1};
                const int nxq = cxw == INFINITY ? 1 : 2, nyq = cyw == INFINITY ? 1 : 2;
                for (int qy = 0; qy < 2 && qh; ++qy)
                    for (int qx = 0; qx < 2 && qh; ++qx) {
                        const int ux = qx < nxq ? qx : 0, uy = qy < nyq ? qy : 0;  // without a split both halves are the whole cell
                        const double x0 = xs[ux], x1 = xs[ux + 1], y0 = ys[uy], y1 = ys[uy + 1];
                        const double qa = (x1 - x0) * (y1 - y0);
                        if (!(qa > 0.0)) { qh = false; break; }
                        double top = -INFINITY;
                        for (const auto& f : hf) {
                            const double tri[3][2] = {{f[0], f[1]}, {f[2], f[3]}, {f[4], f[5]}};
                            if (f[6] > top && clipped_area(tri, x0, y0, x1, y1) > 1e-9 * qa) top = f[6];
                        }
                        if (!std::isfinite(top)) { qh = false; break; }
                        double cover = 0.0;
                        for (const auto& f : hf)
                            if (f[6] == top) {
                                const double tri[3][2] = {{f[0], f[1]}, {f[2], f[3]}, {f[4], f[5]}};
                                cover += clipped_area(tri, x0, y0, x1, y1);
                            }
                        if (cover < qa * (1.0 - 1e-7)) { qh = false; break; }
                        for (int sy = 0; sy < 3 && qh; ++sy)  // overlapping coplanar faces could fake the area: sample as well
                            for (int sx = 0; sx < 3 && qh; ++sx) {
                                const double px_ = x0 + (x1 - x0) * (0.1 + 0.4 * sx), py_ = y0 + (y1 - y0) * (0.1 + 0.4 * sy);
                                bool in = false;
                                for (const auto& f : hf)
                                    if (f[6] == top) {
                                        const double tri[3][2] = {{f[0], f[1]}, {f[2], f[3]}, {f[4], f[5]}};
                                        if (point_in_tri(tri, px_, py_, 1e-9)) { in = true; break; }
                                    }
                                qh = in;
                            }
                        hq[qy * 2 + qx] = (float)top;
                    }
            }
            if (qh) {
                memcpy(&ca[0], &hq[0], 16);  // h00 (x < cx, y < cy), h10 (x > cx), h01 (y > cy), h11
                cb[0] = IMX_CELL_QH;
                memcpy(&cb[1], &cxw, 4);
                memcpy(&cb[2], &cyw, 4);
                cb[3] = face_q;
                ++n_flat;
            } else {
                ++gen_refs[std::min(n, 16)];
                // the first two pairs of references ride in the cell itself: {id2, id0, id1, zrest after the first pair} {kind, first, count, id3}
                ca[1] = key[0];
                ca[2] = key[1];
                ca[0] = key[npad > 2 ? 2 : 1];
                cb[3] = key[npad > 2 ? 3 : 1];
                float zr = -INFINITY;
                for (int k = 2; k < n; ++k) zr = std::max(zr, order[k].first);
                memcpy(&ca[3], &zr, 4);
                cb[0] = IMX_CELL_GENERAL;
                cb[1] = (int32_t)first;
                cb[2] = npad;
            }
        }
    // ---- pass 2b: continuity masks.  A vertical ray within tau of a boundary of its cell must normally look into the neighbouring
    // cell(s) as well (a triangle reaching < tau into a cell is only listed on the other side) and, on QH cells, walk the reference
    // lists: three or four dependent loads for 0.4 % of the rays -- but a quarter of all 64-ray waves has one.  Most of those boundaries
    // run through the middle of a box top, a stair tread, flat ground or a height field.  Bit 8 + 3 (dy + 1) + (dx + 1) of the kind word
    // says: across the boundary towards neighbour (dx, dy) the highest surface CONTINUES -- then the cell's own record answers the ray:
    //   LATTICE | LATTICE   the quads share their corner heights along the common edge (corner for a diagonal neighbour);
    //   QH | QH             the heights on either side of the common edge agree part by part (same split line across, or no step along
    //                       the edge at all) and no horizontal face listed in either cell lies higher anywhere in the 2 tau band
    //                       around the edge (the band is outside both cells' own proofs; QH lists hold horizontal faces and walls only);
    //   x | EMPTY           nothing is listed over there (same band check on the cell's own list).
    // Everything else (QH next to LATTICE, GENERAL anywhere, unequal heights = a real edge) keeps the bit clear: full path, as before.
    int64_t n_cont_bits = 0, n_cont_cells = 0;
    {
        const double band = 2.0 * (double)IMX_GRID_TAU;
        auto kind_at = [&](int ix, int iy) -> int {
            if (ix < 0 || iy < 0 || ix >= (int)nx || iy >= (int)ny) return IMX_CELL_EMPTY;
            return cellv[(size_t)imx_cell_index(ix, iy, ntx) * 8 + 4] & 0xFF;
        };
        auto fl = [&](int32_t w) { float f; memcpy(&f, &w, 4); return f; };
        // a horizontal face of cell c's list above height h with xy-extent reaching into [x0,x1] x [y0,y1] (grid units)?
        auto higher_face = [&](int c, double x0, double y0, double x1, double y1, float h) -> bool {
            for (int k = start[c]; k < start[c + 1]; ++k) {
                const float* rec = &recs[(size_t)refs[k] * 12];
                if (!(rec[2] == rec[5] && rec[5] == rec[8]) || !(rec[2] > h)) continue;
                const double gx0 = std::min({grid_x(rec[0]), grid_x(rec[3]), grid_x(rec[6])}), gx1 = std::max({grid_x(rec[0]), grid_x(rec[3]), grid_x(rec[6])});
                const double gy0 = std::min({grid_y(rec[1]), grid_y(rec[4]), grid_y(rec[7])}), gy1 = std::max({grid_y(rec[1]), grid_y(rec[4]), grid_y(rec[7])});
                if (gx1 > x0 && gx0 < x1 && gy1 > y0 && gy0 < y1) return true;
            }
            return false;
        };
        for (int iy = 0; iy < (int)ny; ++iy)
            for (int ix = 0; ix < (int)nx; ++ix) {
                const int c = imx_cell_index(ix, iy, ntx);
                int32_t* ca = &cellv[(size_t)c * 8];
                int32_t* cb = ca + 4;
                const int kc = cb[0] & 0xFF;
                if (kc != IMX_CELL_LATTICE && kc != IMX_CELL_QH) continue;
                uint32_t mask = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (dx == 0 && dy == 0) continue;
                        const int kn = kind_at(ix + dx, iy + dy);
                        const bool in_grid = ix + dx >= 0 && iy + dy >= 0 && ix + dx < (int)nx && iy + dy < (int)ny;
                        const int cn = in_grid ? imx_cell_index(ix + dx, iy + dy, ntx) : -1;
                        const int32_t* na = in_grid ? &cellv[(size_t)cn * 8] : nullptr;
                        bool ok = false;
                        // the band around the common edge / corner, in grid units
                        const double bx0 = dx < 0 ? ix - band : (dx > 0 ? ix + 1 - band : ix - band), bx1 = dx < 0 ? ix + band : (dx > 0 ? ix + 1 + band : ix + 1 + band);
                        const double by0 = dy < 0 ? iy - band : (dy > 0 ? iy + 1 - band : iy - band), by1 = dy < 0 ? iy + band : (dy > 0 ? iy + 1 + band : iy + 1 + band);
                        if (kc == IMX_CELL_LATTICE) {
                            // corners: [0] a (ix,iy)  [1] d (ix+1,iy)  [2] c (ix,iy+1)  [3] b (ix+1,iy+1)
                            auto corner = [&](const int32_t* q, int cx_, int cy_) { return q[cy_ * 2 + cx_]; };
                            if (kn == IMX_CELL_EMPTY) ok = true;
                            else if (kn == IMX_CELL_LATTICE) {
                                ok = true;
                                // every corner of this quad that lies on the common edge / corner is the same height in the neighbour's quad
                                for (int qy = 0; qy < 2 && ok; ++qy)
                                    for (int qx = 0; qx < 2 && ok; ++qx) {
                                        const int nxq = qx - dx, nyq = qy - dy;  // the same vertex in the neighbour's corner numbering
                                        if (nxq < 0 || nxq > 1 || nyq < 0 || nyq > 1) continue;
                                        ok = corner(ca, qx, qy) == corner(na, nxq, nyq);
                                    }
                            }
                        } else {  // QH: [0] h00 (x<cx,y<cy) [1] h10 [2] h01 [3] h11; cb[1] = cx, cb[2] = cy (+inf: no split)
                            auto hq = [&](const int32_t* q, int qx, int qy) { return q[qy * 2 + qx]; };
                            const int sx = dx > 0 ? 1 : 0, sy = dy > 0 ? 1 : 0;  // this cell's half towards the neighbour (per axis)
                            if (kn == IMX_CELL_EMPTY || kn == IMX_CELL_QH) {
                                const int32_t* nb_ = kn == IMX_CELL_QH ? na + 4 : nullptr;
                                ok = true;
                                float hmax = -INFINITY;  // the heights this cell claims along the edge
                                if (dx != 0 && dy != 0) {
                                    hmax = fl(hq(ca, sx, sy));
                                    if (kn == IMX_CELL_QH) ok = hq(ca, sx, sy) == hq(na, 1 - sx, 1 - sy);
                                } else if (dx != 0) {
                                    hmax = std::max(fl(hq(ca, sx, 0)), fl(hq(ca, sx, 1)));
                                    if (kn == IMX_CELL_QH)
                                        ok = hq(ca, sx, 0) == hq(na, 1 - sx, 0) && hq(ca, sx, 1) == hq(na, 1 - sx, 1) &&
                                             (cb[2] == nb_[2] || (hq(ca, sx, 0) == hq(ca, sx, 1) && hq(na, 1 - sx, 0) == hq(na, 1 - sx, 1)));
                                } else {
                                    hmax = std::max(fl(hq(ca, 0, sy)), fl(hq(ca, 1, sy)));
                                    if (kn == IMX_CELL_QH)
                                        ok = hq(ca, 0, sy) == hq(na, 0, 1 - sy) && hq(ca, 1, sy) == hq(na, 1, 1 - sy) &&
                                             (cb[1] == nb_[1] || (hq(ca, 0, sy) == hq(ca, 1, sy) && hq(na, 0, 1 - sy) == hq(na, 1, 1 - sy)));
                                }
                                // nothing higher hides in the band.  With a step ALONG the edge the lower part could still have a face between
                                // the two heights: compare against the lower of the claimed heights there (conservative)
                                float hmin = hmax;
                                if (dx != 0 && dy == 0) hmin = std::min(fl(hq(ca, sx, 0)), fl(hq(ca, sx, 1)));
                                if (dy != 0 && dx == 0) hmin = std::min(fl(hq(ca, 0, sy)), fl(hq(ca, 1, sy)));
                                if (ok && hmin != hmax) {
                                    // the higher part's own faces are above hmin by design: check part by part along the edge
                                    const float cut = dx != 0 ? fl(cb[2]) : fl(cb[1]);  // the split coordinate along the edge (world)
                                    const double gcut = dx != 0 ? grid_y(cut) : grid_x(cut);
                                    for (int part = 0; part < 2 && ok; ++part) {
                                        const float hp = dx != 0 ? fl(hq(ca, sx, part)) : fl(hq(ca, part, sy));
                                        double px0 = bx0, px1 = bx1, py0 = by0, py1 = by1;
                                        if (dx != 0) { if (part == 0) py1 = gcut; else py0 = gcut; } else { if (part == 0) px1 = gcut; else px0 = gcut; }
                                        ok = !higher_face(c, px0, py0, px1, py1, hp) && (cn < 0 || !higher_face(cn, px0, py0, px1, py1, hp));
                                    }
                                } else if (ok) {
                                    ok = !higher_face(c, bx0, by0, bx1, by1, hmax) && (cn < 0 || !higher_face(cn, bx0, by0, bx1, by1, hmax));
                                }
                            }
                        }
                        if (ok) mask |= 1u << (3 * (dy + 1) + (dx + 1));
                    }
                cb[0] |= (int32_t)(mask << 8);
                n_cont_bits += __builtin_popcount(mask);
                n_cont_cells += (mask & 0x1EFu) == 0x1EFu;
            }
    }
    for (auto& v : gxl) if (std::isnan(v)) v = 0.0f;  // lines no lattice cell uses
    for (auto& v : gyl) if (std::isnan(v)) v = 0.0f;
    if (getenv("IMX_MESH_STATS")) {  // debugging aid: histogram of references per general cell
        std::vector<int64_t> hist(16, 0), histv(16, 0);
        int64_t degenerate = 0, total = 0;
        for (int64_t c = 0; c < ncell; ++c) {
            const int n = start[c + 1] - start[c];
            if (n == 0 || (cellv[(size_t)c * 8 + 4] & 0xFF) == IMX_CELL_LATTICE) continue;
            int nv = 0;
            for (int k = 0; k < n; ++k) {
                const uint32_t* t = tris + 3 * (size_t)refs[start[c] + k];
                const float* a = verts + 3 * (size_t)t[0]; const float* b = verts + 3 * (size_t)t[1]; const float* cc = verts + 3 * (size_t)t[2];
                const double area = (double)(b[0] - a[0]) * (cc[1] - a[1]) - (double)(b[1] - a[1]) * (cc[0] - a[0]);
                if (area != 0.0) ++nv; else ++degenerate;
                ++total;
            }
            hist[std::min(n, 15)]++; histv[std::min(nv, 15)]++;
        }
        fprintf(stderr, "[imx mesh] general cells by #refs:");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %d:%lld", i, (long long)hist[i]);
        fprintf(stderr, "\n[imx mesh] general cells by #non-degenerate refs:");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %d:%lld", i, (long long)histv[i]);
        fprintf(stderr, "\n[imx mesh] GENERAL cells %lld: sloped triangle %lld, two split lines on an axis %lld; by #refs:", (long long)(n_general - n_flat), (long long)why_general[0], (long long)why_general[1]);
        for (int i = 0; i <= 16; ++i) fprintf(stderr, " %d:%lld", i, (long long)gen_refs[i]);
        fprintf(stderr, "\n[imx mesh] continuity: %lld cells with all 8 neighbours continuous, %lld bits set (of %lld)", (long long)n_cont_cells, (long long)n_cont_bits, (long long)(8 * (n_lattice + n_flat)));
        fprintf(stderr, "\n[imx mesh] degenerate (zero xy-area) refs %lld of %lld\n", (long long)degenerate, (long long)total);
    }
    IMX_REQUIRE(refl.size() / 2 < (1ull << 31), "imx_mesh_create: too many general cell references");
    if (refl.empty()) refl.assign(4, 0);

    IMX_REQUIRE(imx_device_count() > 0, "imx_mesh_create: no GPU visible");
    auto m = std::make_unique<imx_mesh>();
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {
        IMX_HIP(hipMalloc(dst, bytes));
        IMX_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    if (up((void**)&m->d_cell, cellv.data(), cellv.size() * 4) || up((void**)&m->d_cell_list, clist.data(), clist.size() * 4) ||
        up((void**)&m->d_gx, gxl.data(), gxl.size() * 4) || up((void**)&m->d_gy, gyl.data(), gyl.size() * 4) ||
        up((void**)&m->d_tri_rec, recs.data(), recs.size() * 4) || up((void**)&m->d_refs, refl.data(), refl.size() * 4))
        return 1;
    m->v.refs = reinterpret_cast<const int4*>(m->d_refs);
    m->v.cells = reinterpret_cast<const int4*>(m->d_cell);
    m->v.cell_list = reinterpret_cast<const int2*>(m->d_cell_list);
    m->v.gx = m->d_gx;
    m->v.gy = m->d_gy;
    m->v.tri_rec = reinterpret_cast<const float4*>(m->d_tri_rec);
    m->v.nx = (int)nx; m->v.ny = (int)ny; m->v.ntx = ntx; m->v.nty = nty;
    m->v.x0 = xmin; m->v.y0 = ymin; m->v.cell = cell_size; m->v.inv_cell = inv_cell;
    m->v.F = F;
    m->num_refs = (int64_t)(refl.size() / 2);
    m->max_refs = max_refs;
    m->n_lattice = n_lattice;
    m->n_general = n_general;
    m->n_flat = n_flat;
    *out = m.release();
    return 0;
}

extern "C" void imx_mesh_destroy(imx_mesh_t* m) {
    if (!m) return;
    if (m->d_tri_rec) (void)hipFree(m->d_tri_rec);
    if (m->d_cell) (void)hipFree(m->d_cell);
    if (m->d_cell_list) (void)hipFree(m->d_cell_list);
    if (m->d_gx) (void)hipFree(m->d_gx);
    if (m->d_gy) (void)hipFree(m->d_gy);
    if (m->d_refs) (void)hipFree(m->d_refs);
    delete m;
}

extern "C" int imx_mesh_info(const imx_mesh_t* m, int64_t* info8) {
    IMX_REQUIRE(m && info8, "imx_mesh_info: null argument");
    info8[0] = m->v.nx; info8[1] = m->v.ny; info8[2] = m->v.F; info8[3] = m->num_refs; info8[4] = m->max_refs;
    info8[5] = m->n_lattice; info8[6] = m->n_general;  // info8[6] includes the QH cells; their count rides in the upper half of info8[4]
    info8[4] = (int64_t)m->max_refs | (m->n_flat << 32);
    int32_t b;
    memcpy(&b, &m->v.cell, 4); info8[7] = b;
    return 0;
}
