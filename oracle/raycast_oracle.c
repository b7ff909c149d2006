/* TEST INFRASTRUCTURE -- CPU oracle for the RayCaster height-scanner path.  Never linked into the product.
 *
 * The reference casts rays with NVIDIA Warp: `wp.mesh_query_ray(mesh, start, dir, max_dist, t,u,v,sign,n,f)` from
 * `raycast_mesh_kernel` (reference isaaclab/utils/warp/kernels.py:13-75, call at :66, hit = start + t*dir at :69;
 * misses stay +inf: isaaclab/utils/warp/ops.py:70).  `warp-lang` is a third-party dependency that is absent from
 * /root/reference and from this image (unpinned in source/isaaclab/setup.py:37), so its BVH traversal cannot be run.
 * PARITY UNPINNED by runnable reference tests.  Two restatements of the published
 * algorithms are given instead:
 *
 *  (1) imxo_raycast_f64: brute-force closest hit over ALL triangles with the Moller-Trumbore test in fp64
 *      (two-sided, t in [0, max_dist]); the geometric ground truth.
 *  (2) imxo_raycast_woop_f32: brute-force closest hit with the fp32 watertight test of Woop, Benthin & Wald,
 *      "Watertight Ray/Triangle Intersection" (JCGT 2013), which is the per-triangle test Warp's mesh query uses;
 *      same arithmetic order as the HIP kernel, so the two agree to the last bit on non-degenerate hits.
 *
 * Both write hit = start + t*dir in fp32 (as the reference kernel does) and leave +inf on a miss.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

int imxo_raycast_f64(const float* verts, const uint32_t* tris, int64_t ntris, const float* starts, const float* dirs,
                     int64_t nrays, double max_dist, float* hits, double* t_out, int32_t* face_out) {
    const double eps = 1e-12;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nrays; ++r) {
        const double o[3] = {starts[3 * r], starts[3 * r + 1], starts[3 * r + 2]};
        const double d[3] = {dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]};
        double best = INFINITY;
        int32_t bestf = -1;
        for (int64_t f = 0; f < ntris; ++f) {
            const float* pa = verts + 3 * (size_t)tris[3 * f];
            const float* pb = verts + 3 * (size_t)tris[3 * f + 1];
            const float* pc = verts + 3 * (size_t)tris[3 * f + 2];
            const double e1[3] = {(double)pb[0] - pa[0], (double)pb[1] - pa[1], (double)pb[2] - pa[2]};
            const double e2[3] = {(double)pc[0] - pa[0], (double)pc[1] - pa[1], (double)pc[2] - pa[2]};
            double p[3], q[3];
            cross3(d, e2, p);
            const double det = dot3(e1, p);
            if (fabs(det) < 1e-30) continue;
            const double inv = 1.0 / det;
            const double tv[3] = {o[0] - pa[0], o[1] - pa[1], o[2] - pa[2]};
            const double u = dot3(tv, p) * inv;
            if (u < -eps || u > 1.0 + eps) continue;
            cross3(tv, e1, q);
            const double v = dot3(d, q) * inv;
            if (v < -eps || u + v > 1.0 + eps) continue;
            const double t = dot3(e2, q) * inv;
            if (t >= 0.0 && t <= max_dist && t < best) {
                best = t;
                bestf = (int32_t)f;
            }
        }
        if (bestf >= 0) {
            const float tf = (float)best;
            hits[3 * r + 0] = starts[3 * r + 0] + tf * dirs[3 * r + 0];
            hits[3 * r + 1] = starts[3 * r + 1] + tf * dirs[3 * r + 1];
            hits[3 * r + 2] = starts[3 * r + 2] + tf * dirs[3 * r + 2];
        } else {
            hits[3 * r + 0] = hits[3 * r + 1] = hits[3 * r + 2] = INFINITY;
        }
        if (t_out) t_out[r] = best;
        if (face_out) face_out[r] = bestf;
    }
    return 0;
}

static inline int max_dim3(const float v[3]) {
    const float x = fabsf(v[0]), y = fabsf(v[1]), z = fabsf(v[2]);
    return (x > y) ? ((x > z) ? 0 : 2) : ((y > z) ? 1 : 2);
}

/* Woop et al. 2013, section 3 (fp32; the double-precision edge fallback of the paper is kept). */
static inline int woop_f32(const float org[3], const float dir[3], const float* pa, const float* pb, const float* pc,
                           float* t_hit) {
    int kz = max_dim3(dir);
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (dir[kz] < 0.0f) { int tmp = kx; kx = ky; ky = tmp; }
    const float Sx = dir[kx] / dir[kz];
    const float Sy = dir[ky] / dir[kz];
    const float Sz = 1.0f / dir[kz];
    const float A[3] = {pa[0] - org[0], pa[1] - org[1], pa[2] - org[2]};
    const float B[3] = {pb[0] - org[0], pb[1] - org[1], pb[2] - org[2]};
    const float C[3] = {pc[0] - org[0], pc[1] - org[1], pc[2] - org[2]};
    const float Ax = A[kx] - Sx * A[kz], Ay = A[ky] - Sy * A[kz];
    const float Bx = B[kx] - Sx * B[kz], By = B[ky] - Sy * B[kz];
    const float Cx = C[kx] - Sx * C[kz], Cy = C[ky] - Sy * C[kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        const double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        const double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        const double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    const float det = U + V + W;
    if (det == 0.0f) return 0;
    const float Az = Sz * A[kz], Bz = Sz * B[kz], Cz = Sz * C[kz];
    const float T = U * Az + V * Bz + W * Cz;
    /* sign test of the paper: T and det must agree in sign (t >= 0) */
    if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) return 0;
    const float rcp = 1.0f / det;
    *t_hit = T * rcp;
    return 1;
}

int imxo_raycast_woop_f32(const float* verts, const uint32_t* tris, int64_t ntris, const float* starts,
                          const float* dirs, int64_t nrays, float max_dist, float* hits, float* t_out,
                          int32_t* face_out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nrays; ++r) {
        const float* o = starts + 3 * r;
        const float* d = dirs + 3 * r;
        float best = max_dist;
        int32_t bestf = -1;
        for (int64_t f = 0; f < ntris; ++f) {
            float t;
            if (woop_f32(o, d, verts + 3 * (size_t)tris[3 * f], verts + 3 * (size_t)tris[3 * f + 1],
                         verts + 3 * (size_t)tris[3 * f + 2], &t)) {
                if (t >= 0.0f && (t < best || (bestf < 0 && t <= best))) {
                    best = t;
                    bestf = (int32_t)f;
                }
            }
        }
        if (bestf >= 0) {
            hits[3 * r + 0] = o[0] + best * d[0];
            hits[3 * r + 1] = o[1] + best * d[1];
            hits[3 * r + 2] = o[2] + best * d[2];
        } else {
            hits[3 * r + 0] = hits[3 * r + 1] = hits[3 * r + 2] = INFINITY;
        }
        if (t_out) t_out[r] = bestf >= 0 ? best : INFINITY;
        if (face_out) face_out[r] = bestf;
    }
    return 0;
}
