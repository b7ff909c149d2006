// Micro-benchmark (experiment, not product): where does an f32-MFMA stream lose time?  hipcc --offload-arch=gfx950 -O3 tools/bench_mfma.hip -o tools/bench_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 160
// MODE 0: MFMAs only (register operands); 1: + LDS ping-pong reads; 2: + one barrier per stage; 4: b128 reads from a [row][k] layout (pitch 36 floats), four k-steps per float4, + barrier
template <int MODE>
__global__ void __launch_bounds__(512) k(int nst, float* out) {
    __shared__ float sY[2][32 * LD];
    __shared__ float sX[2][32 * LD];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * 32 * LD; i += blockDim.x) { (&sY[0][0])[i] = 1.0f + i; (&sX[0][0])[i] = 0.5f; }
    __syncthreads();
    if (t >= 256) {
        if (MODE >= 2) for (int st = 0; st < nst; ++st) __syncthreads();
        return;
    }
    const int lane = t & 63, w = t >> 6, r = lane & 31, half = lane >> 5, wn = w & 1, wk = w >> 1;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
    if (MODE == 4) {
        const float* base = &sY[0][0];
        for (int st = 0; st < nst; ++st) {
            const float* pa = base + (st & 1) * 4608 + (wn * 64 + r) * 36 + 4 * half;       // [128 rows][36]
            const float* pb = base + 9216 + (st & 1) * 4608 + (wk * 64 + r) * 36 + 4 * half;
            float4 A0 = *(const float4*)pa, A1 = *(const float4*)(pa + 32 * 36), B0 = *(const float4*)pb, B1 = *(const float4*)(pb + 32 * 36);
            float4 An0 = A0, An1 = A1, Bn0 = B0, Bn1 = B1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) { An0 = *(const float4*)(pa + 8 * (g + 1)); An1 = *(const float4*)(pa + 32 * 36 + 8 * (g + 1)); Bn0 = *(const float4*)(pb + 8 * (g + 1)); Bn1 = *(const float4*)(pb + 32 * 36 + 8 * (g + 1)); }
                __builtin_amdgcn_sched_barrier(0);
#define STEP(c) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, acc[0][0], 0, 0, 0); acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, acc[0][1], 0, 0, 0); acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, acc[1][0], 0, 0, 0); acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, acc[1][1], 0, 0, 0);
                STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
                __builtin_amdgcn_sched_barrier(0);
                A0 = An0; A1 = An1; B0 = Bn0; B1 = Bn1;
            }
            __syncthreads();
        }
    } else
    for (int st = 0; st < nst; ++st) {
        const float* py = sY[st & 1] + half * LD + wn * 64 + r;
        const float* px = sX[st & 1] + half * LD + wk * 64 + r;
        float p0 = py[0], p1 = py[32], p2 = px[0], p3 = px[32], q0 = p0, q1 = p1, q2 = p2, q3 = p3;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            if (MODE >= 1) { q0 = py[(kk + 1) * 2 * LD]; q1 = py[(kk + 1) * 2 * LD + 32]; q2 = px[(kk + 1) * 2 * LD]; q3 = px[(kk + 1) * 2 * LD + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, p2, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, p3, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, p2, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, p3, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1 && kk + 2 < 16) { p0 = py[(kk + 2) * 2 * LD]; p1 = py[(kk + 2) * 2 * LD + 32]; p2 = px[(kk + 2) * 2 * LD]; p3 = px[(kk + 2) * 2 * LD + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, q2, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, q3, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, q2, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, q3, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + t] = s;
}
template <int MODE>
void run(int grid, int block, int nst, float* out, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, nst, out);
    hipEventRecord(a);
    const int L = 20;
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, nst, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / L, fl = (double)grid * 4 * nst * 64 * 4096.0;
    printf("%-44s grid %4d block %3d nst %3d: %7.1f us  %6.1f TF\n", name, grid, block, nst, us, fl / us / 1e6);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int nst : {24, 96}) {
        run<0>(256, 256, nst, out, "MFMA only");
        run<1>(256, 256, nst, out, "+ LDS ping-pong reads");
        run<2>(256, 256, nst, out, "+ barrier/stage (4 waves)");
        run<2>(256, 512, nst, out, "+ barrier/stage (4 mult + 4 idle waves)");
        run<4>(256, 512, nst, out, "b128 [row][k] pitch-36 reads + barrier (4 mult + 4 idle)");
        run<0>(512, 256, nst, out, "MFMA only, 2 WG/CU");
        run<1>(512, 256, nst, out, "+ LDS reads, 2 WG/CU");
    }
    return 0;
}
