// Micro-benchmark: where does the height-scanner ray kernel spend its time?  (experiments, not product code)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/bench_raycast.hip -Iisaaclab_amd/csrc -Lisaaclab_amd -limx -o /tmp/bench_raycast
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "imx_internal.h"
#include "imx_raycast.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Env { float px, py, pz, yw, yz; };

// MODE 0 full; 1 = no triangle test (desc + corners loaded, summed); 2 = desc only; 3 = no mesh loads; 4 = full, all envs same cell
template <int MODE>
__global__ void __launch_bounds__(256) k_rays(MeshView M, const Env* __restrict__ envs, const float* __restrict__ local, int R,
                                              float* __restrict__ out) {
    const int e = blockIdx.x;
    const Env en = envs[MODE == 4 ? 0 : e];
    for (int j = threadIdx.x; j < R; j += blockDim.x) {
        float sx, sy, sz;
        quat_apply_yaw_only(en.yw, en.yz, local[3 * j], local[3 * j + 1], local[3 * j + 2], sx, sy, sz);
        sx += en.px; sy += en.py; sz += en.pz;
        float v = 0.0f;
        if (MODE == 0 || MODE == 4) {
            float t; int32_t f;
            v = cast_ray_vertical(M, sx, sy, sz, -1.0f, 1e6f, t, f) ? sz - t : __builtin_huge_valf();
        } else if (MODE == 3) {
            v = sx + sy;
        } else {
            int nbx, nby;
            const int ix = cell_of((sx - M.x0) * M.inv_cell, nbx), iy = cell_of((sy - M.y0) * M.inv_cell, nby);
            if (ix >= 0 && iy >= 0 && ix < M.nx && iy < M.ny) {
                const int c = imx_cell_index(ix, iy, M.ntx);
                const int32_t d = M.cell_desc[c].x;
                v = (float)d;
                if (MODE == 1) {
                    const float4* p = M.tile_pool + (size_t)(c >> 6) * 81 + ((iy & 7) * 9 + (ix & 7));
                    v += p[0].z + p[1].z + p[9].z + p[10].z;
                }
            }
        }
        out[(size_t)e * R + j] = v;
    }
}

template <int MODE>
float run(MeshView M, const Env* envs, const float* local, int N, int R, float* out, int bs) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_rays<MODE>, dim3(N), dim3(bs), 0, 0, M, envs, local, R, out);
    CK(hipEventRecord(a));
    const int L = 100;
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_rays<MODE>, dim3(N), dim3(bs), 0, 0, M, envs, local, R, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / L;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, R = 187;
    const int rows = 801, cols = 1601;
    const float hs = 0.1f;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> v((size_t)rows * cols * 3);
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) {
            float* p = &v[((size_t)i * cols + j) * 3];
            p[0] = i * hs - 40.0f; p[1] = j * hs - 80.0f; p[2] = 0.05f * U(rng);
        }
    std::vector<uint32_t> t((size_t)(rows - 1) * (cols - 1) * 6);
    size_t k = 0;
    for (int i = 0; i < rows - 1; ++i)
        for (int j = 0; j < cols - 1; ++j) {
            const uint32_t i0 = i * cols + j, i1 = i0 + 1, i2 = i0 + cols, i3 = i2 + 1;
            t[k++] = i0; t[k++] = i3; t[k++] = i1; t[k++] = i0; t[k++] = i2; t[k++] = i3;
        }
    imx_mesh_t* mesh = nullptr;
    if (imx_mesh_create(v.data(), (int64_t)rows * cols, t.data(), (int64_t)t.size() / 3, hs, &mesh)) { printf("mesh: %s\n", imx_last_error()); return 1; }
    int64_t info[8]; imx_mesh_info(mesh, info);
    printf("mesh nx %lld ny %lld F %lld general recs %lld lattice %lld general cells %lld\n", (long long)info[0], (long long)info[1],
           (long long)info[2], (long long)info[3], (long long)info[5], (long long)info[6]);
    std::vector<Env> envs(N);
    for (auto& e : envs) {
        e.px = -38.f + 76.f * U(rng); e.py = -78.f + 156.f * U(rng); e.pz = 20.6f;
        const float yaw = 6.2831853f * U(rng);
        e.yw = cosf(0.5f * yaw); e.yz = sinf(0.5f * yaw);
    }
    std::vector<float> local(R * 3);
    for (int j = 0; j < R; ++j) { local[3 * j] = -0.8f + 0.1f * (j % 17); local[3 * j + 1] = -0.5f + 0.1f * (j / 17); local[3 * j + 2] = 0.f; }
    Env* d_env; float *d_local, *d_out;
    CK(hipMalloc(&d_env, N * sizeof(Env))); CK(hipMalloc(&d_local, R * 12)); CK(hipMalloc(&d_out, (size_t)N * R * 4));
    CK(hipMemcpy(d_env, envs.data(), N * sizeof(Env), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_local, local.data(), R * 12, hipMemcpyHostToDevice));
    const MeshView M = mesh->v;
    for (int bs : {64, 192, 256}) {
        printf("block %3d: full %.1f us | no-tri-test %.1f | desc-only %.1f | no-loads %.1f | same-cell %.1f\n", bs,
               run<0>(M, d_env, d_local, N, R, d_out, bs), run<1>(M, d_env, d_local, N, R, d_out, bs),
               run<2>(M, d_env, d_local, N, R, d_out, bs), run<3>(M, d_env, d_local, N, R, d_out, bs),
               run<4>(M, d_env, d_local, N, R, d_out, bs));
    }
    // sorted envs (x-major) -> neighbouring blocks touch neighbouring tiles
    std::sort(envs.begin(), envs.end(), [](const Env& a, const Env& b) { return (int)(a.py / 0.8f) != (int)(b.py / 0.8f) ? a.py < b.py : a.px < b.px; });
    CK(hipMemcpy(d_env, envs.data(), N * sizeof(Env), hipMemcpyHostToDevice));
    printf("sorted envs, block 192: full %.1f us\n", run<0>(M, d_env, d_local, N, R, d_out, 192));
    imx_mesh_destroy(mesh);
    return 0;
}
