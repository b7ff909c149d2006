// Terrain mesh upload + 2-D uniform-grid build (host side, counting sort).  Replaces convert_to_warp_mesh /
// wp.Mesh's BVH build (reference isaaclab/utils/warp/ops.py:130-145, called once at RayCaster init,
// sensors/ray_caster/ray_caster.py:182-189).  Membership rule: see imx_raycast.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "imx_internal.h"

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
    IMX_REQUIRE(V > 0 && F > 0 && F < (1ll << 31), "imx_mesh_create: empty or oversized mesh (V=%lld F=%lld)",
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
    const int64_t ncell = (int64_t)ntx * nty * 64;  // 8x8-tiled layout (imx_cell_index)

    std::vector<int32_t> start((size_t)ncell + 1, 0);
    std::vector<float> tv((size_t)F * 9);  // host copy of the corners, packed per reference below
    std::vector<int32_t> ra((size_t)F * 4);  // per-triangle cell ranges
    for (int64_t f = 0; f < F; ++f) {
        float lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
        for (int c = 0; c < 3; ++c) {
            const float* p = verts + 3 * (size_t)tris[3 * f + c];
            tv[f * 9 + c * 3 + 0] = p[0]; tv[f * 9 + c * 3 + 1] = p[1]; tv[f * 9 + c * 3 + 2] = p[2];
            lox = std::min(lox, p[0]); hix = std::max(hix, p[0]);
            loy = std::min(loy, p[1]); hiy = std::max(hiy, p[1]);
        }
        int ax, bx, ay, by;
        cell_range(lox, hix, xmin, inv_cell, (int)nx, ax, bx);
        cell_range(loy, hiy, ymin, inv_cell, (int)ny, ay, by);
        ra[f * 4 + 0] = ax; ra[f * 4 + 1] = bx; ra[f * 4 + 2] = ay; ra[f * 4 + 3] = by;
        for (int iy = ay; iy <= by; ++iy)
            for (int ix = ax; ix <= bx; ++ix) start[(size_t)imx_cell_index(ix, iy, ntx) + 1]++;
    }
    int32_t max_refs = 0;
    for (int64_t c = 0; c < ncell; ++c) {
        max_refs = std::max(max_refs, start[c + 1]);
        IMX_REQUIRE((int64_t)start[c] + start[c + 1] < (1ll << 31), "imx_mesh_create: too many cell references");
        start[c + 1] += start[c];
    }
    const int64_t nrefs = start[ncell];
    // per-reference 48-byte records (see MeshView): ascending triangle id inside every cell -> deterministic tie-break
    std::vector<float> recs((size_t)std::max<int64_t>(nrefs, 1) * 12, 0.0f);
    std::vector<int32_t> cursor(start.begin(), start.end() - 1);
    for (int64_t f = 0; f < F; ++f)
        for (int iy = ra[f * 4 + 2]; iy <= ra[f * 4 + 3]; ++iy)
            for (int ix = ra[f * 4 + 0]; ix <= ra[f * 4 + 1]; ++ix) {
                float* r = &recs[(size_t)(cursor[(size_t)imx_cell_index(ix, iy, ntx)]++) * 12];
                memcpy(r, &tv[(size_t)f * 9], 9 * sizeof(float));
                const int32_t fid = (int32_t)f;
                memcpy(r + 9, &fid, 4);
            }

    IMX_REQUIRE(imx_device_count() > 0, "imx_mesh_create: no GPU visible");
    auto m = std::make_unique<imx_mesh>();
    IMX_HIP(hipMalloc((void**)&m->d_tri_rec, recs.size() * sizeof(float)));
    IMX_HIP(hipMalloc((void**)&m->d_cell_start, start.size() * sizeof(int32_t)));
    IMX_HIP(hipMemcpy(m->d_tri_rec, recs.data(), recs.size() * sizeof(float), hipMemcpyHostToDevice));
    IMX_HIP(hipMemcpy(m->d_cell_start, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    m->v.tri_rec = m->d_tri_rec;
    m->v.cell_start = m->d_cell_start;
    m->v.nx = (int)nx; m->v.ny = (int)ny; m->v.ntx = ntx; m->v.nty = nty;
    m->v.x0 = xmin; m->v.y0 = ymin; m->v.cell = cell_size; m->v.inv_cell = inv_cell;
    m->v.F = F;
    m->num_refs = nrefs;
    m->max_refs = max_refs;
    *out = m.release();
    return 0;
}

extern "C" void imx_mesh_destroy(imx_mesh_t* m) {
    if (!m) return;
    if (m->d_tri_rec) (void)hipFree(m->d_tri_rec);
    if (m->d_cell_start) (void)hipFree(m->d_cell_start);
    delete m;
}

extern "C" int imx_mesh_info(const imx_mesh_t* m, int64_t* info8) {
    IMX_REQUIRE(m && info8, "imx_mesh_info: null argument");
    info8[0] = m->v.nx; info8[1] = m->v.ny; info8[2] = m->v.F; info8[3] = m->num_refs; info8[4] = m->max_refs;
    int32_t b;
    memcpy(&b, &m->v.x0, 4); info8[5] = b;
    memcpy(&b, &m->v.y0, 4); info8[6] = b;
    memcpy(&b, &m->v.cell, 4); info8[7] = b;
    return 0;
}
