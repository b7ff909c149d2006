// How fast does the chip start workgroups?  Empty / one-round-trip kernels at the grid sizes of the observation kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o gpurun_out/dispatch_rate && gpurun_out/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(float* out) {
    if (out == nullptr && threadIdx.x == 9999) out[0] = 1.0f;
}
__global__ void k_trip(const float* __restrict__ in, float* __restrict__ out, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = in[(i * 2654435761u) & mask] + 1.0f;  // one scattered load, one coalesced store
}
__global__ void k_valu(const float* __restrict__ in, float* __restrict__ out, unsigned mask, int iters) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[(i * 2654435761u) & mask];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;  // dependent FMAs: iters VALU instructions
    out[i] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class F>
static float time_us(F launch, int n = 200) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / n;
}

int main() {
    const size_t NIN = 1u << 26;  // 256 MB of floats: scattered loads miss
    float *in, *out;
    CK(hipMalloc(&in, NIN * 4));
    CK(hipMalloc(&out, (size_t)327680 * 256 * 4));
    CK(hipMemset(in, 0, NIN * 4));
    const int grids[] = {4096, 8192, 12288, 20480, 65536, 131072, 327680};
    printf("%8s %6s | %10s %10s %10s %10s\n", "grid", "block", "empty us", "1 trip us", "+200 VALU", "+400 VALU");
    for (int bs : {64, 256})
        for (int g : grids) {
            const float t0 = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(bs), 0, 0, out); });
            const float t1 = time_us([&] { hipLaunchKernelGGL(k_trip, dim3(g), dim3(bs), 0, 0, in, out, (unsigned)(NIN - 1)); });
            const float t2 = time_us([&] { hipLaunchKernelGGL(k_valu, dim3(g), dim3(bs), 0, 0, in, out, (unsigned)(NIN - 1), 200); });
            const float t3 = time_us([&] { hipLaunchKernelGGL(k_valu, dim3(g), dim3(bs), 0, 0, in, out, (unsigned)(NIN - 1), 400); });
            printf("%8d %6d | %10.2f %10.2f %10.2f %10.2f   (%.0f waves/us empty)\n", g, bs, t0, t1, t2, t3, g * (bs / 64) / t0);
        }
    return 0;
}
