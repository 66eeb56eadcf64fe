// Round 3: do two waves on one SIMD overlap when each alternates a burst of MFMAs with a burst of VALU work (the shape of a wave of the
// pair kernels: a tile's MFMAs, then its epilogue)?  Per iteration a wave issues NM dependent v_mfma_f32_32x32x16_bf16 and then NV VALU
// instructions (v_fma_f32 on four independent registers); 1 or 2 waves per SIMD, with or without an s_barrier per iteration.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/r3/phase_overlap profiles/microbench/phase_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
template <int NM, int NV, bool BAR>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters)
{
    f32x16 acc = {};
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + threadIdx.x; b[i] = 0x3f803f80u; }
    float v[4] = {1.0f + threadIdx.x, 2.0f, 3.0f, 4.0f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) MFMA(acc, a, b);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i & 3]));
        if (BAR) asm volatile("s_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    float s = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NM, int NV, bool BAR>
void run(float *out, long long *cyc)
{
    for (int w = 1; w <= 2; ++w) {
        const int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NM, NV, BAR>), dim3(256), dim3(256 * w), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        long long c;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%2d MFMA (%4d cycles) + %3d VALU per iteration%s, %d wave(s)/SIMD: %7.1f cycles per iteration per wave\n", NM, NM * 32, NV,
               BAR ? " + s_barrier" : "", w, (double)c / iters);
    }
}
int main()
{
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    run<24, 0, false>(out, cyc); run<0, 90, false>(out, cyc); run<24, 90, false>(out, cyc); run<24, 90, true>(out, cyc);
    run<48, 180, false>(out, cyc); run<12, 45, false>(out, cyc); run<12, 45, true>(out, cyc);
    return 0;
}
