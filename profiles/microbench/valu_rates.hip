// Round 3: issue cost of the instructions the float64 step is made of (gfx950), cycles per wave-instruction with W waves per SIMD
// (independent operands, 8 instructions of one kind per loop iteration).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S S S S S S S S
template <int KIND>
__global__ __launch_bounds__(1024) void k(double *out, long long *cyc, int iters)
{
    double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.0f + threadIdx.x * 1e-3f + i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define OPD(I) \
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[I])); \
        else if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[I])); \
        else if (KIND == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[I])); \
        else if (KIND == 3) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[I])); \
        else if (KIND == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[I])); \
        else if (KIND == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[I]) : "v"(f[I])); \
        else if (KIND == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[I]) : "v"(d[I])); \
        else if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[I])); \
        else if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[I])); \
        else if (KIND == 9) asm volatile("v_exp_f32 %0, %0" : "+v"(f[I])); \
        else if (KIND == 10) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[I])); \
        else if (KIND == 11) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(d[I]), "v"(d[(I + 1) & 7]) : "vcc"); \
        else if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[I]) : "v"(f[(I + 1) & 7])); \
        else if (KIND == 13) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[I]) : "v"(d[(I + 1) & 7]));
        OPD(0) OPD(1) OPD(2) OPD(3) OPD(4) OPD(5) OPD(6) OPD(7)
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> void run(const char *name, double *out, long long *cyc)
{
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * w), 0, 0, out, cyc, 2000);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * w), 0, 0, out, cyc, 2000);
        (void)hipDeviceSynchronize();
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-14s waves/SIMD %d: %6.2f cycles per instruction per wave, %6.2f per instruction on the SIMD\n", name, w, (double)c / 16000, (double)c / 16000 / w);
    }
}
int main()
{
    double *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 1024 * 8); (void)hipMalloc(&cyc, 8);
    run<0>("v_fma_f64", out, cyc); run<1>("v_mul_f64", out, cyc); run<2>("v_add_f64", out, cyc); run<3>("v_rsq_f64", out, cyc); run<4>("v_rcp_f64", out, cyc);
    run<5>("v_cvt_f64_f32", out, cyc); run<6>("v_cvt_f32_f64", out, cyc); run<7>("v_rsq_f32", out, cyc); run<8>("v_fma_f32", out, cyc); run<9>("v_exp_f32", out, cyc);
    run<10>("v_sqrt_f32", out, cyc); run<11>("v_cmp_gt_f64", out, cyc); run<12>("v_cndmask_b32", out, cyc); run<13>("v_max_f64", out, cyc);
    return 0;
}
