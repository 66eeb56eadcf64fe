// Does VALU work overlap the matrix pipe inside ONE wave on gfx950?  One workgroup per CU, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>   // NV VALU ops per MFMA; KIND 0 = v_fma_f32, 1 = v_exp_f32 (quarter rate), 2 = ds_read_b128
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters)
{
    __shared__ uint4 lds[4096];
    f32x16 acc = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.0f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i;
    uint4 r[8] = {};
    lds[threadIdx.x] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j % 8]));
            else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j % 8]));
            else asm volatile("ds_read_b128 %0, %1" : "=v"(r[j % 8]) : "v"((threadIdx.x & 63) * 16 + j * 1024));
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i] + r[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NV, int KIND>
void run(int waves_per_simd, float *out, long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("kind %d  valu/mfma %2d  waves/simd %d : %6.1f cycles per MFMA-iteration per wave, %6.1f per MFMA on the SIMD\n", KIND, NV, waves_per_simd,
           (double)c / iters, (double)c / iters / waves_per_simd);
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(w, out, cyc); run<2, 0>(w, out, cyc); run<4, 0>(w, out, cyc); run<6, 0>(w, out, cyc); run<8, 0>(w, out, cyc); run<12, 0>(w, out, cyc);
        run<1, 1>(w, out, cyc); run<2, 1>(w, out, cyc); run<4, 1>(w, out, cyc);
        run<1, 2>(w, out, cyc); run<2, 2>(w, out, cyc);
    }
    return 0;
}
