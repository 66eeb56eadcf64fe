// Round 3: what does ONE wave hide behind v_mfma_f32_32x32x16_bf16 when consecutive MFMAs go to DIFFERENT accumulators?
// (profiles/microbench/mfma_valu_overlap.hip chained every MFMA on one accumulator: 44 cycles per MFMA with nothing else,
//  every filler exposed.  The shipped kernels have exactly that shape: one accumulator per M-tile, 16-32 dependent MFMAs.)
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_multiacc profiles/microbench/mfma_multiacc.hip
// One workgroup per CU; W waves per SIMD; NACC accumulators used round-robin; per MFMA: RD ds_read_b128 (ring of 8, waited
// 8 behind) and TQ quarter-tanh groups (a group = the 4 instructions of one tanh: v_exp_f32, v_add_f32, v_rcp_f32, v_fma_f32;
// every second group is followed by one v_cvt_pk_bf16_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int NACC, int TANH4, int RD, int PLAINV>   // TANH4 = tanh per 4 MFMAs (0..8), RD = ds_read per MFMA (0/1), PLAINV = plain v_fma per MFMA
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters)
{
    __shared__ uint4 lds[8192];
    f32x16 acc[4] = {};
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + threadIdx.x; b[i] = 0x3f803f80u; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * threadIdx.x + i;
    unsigned pk[4] = {};
    u32x4 r[8];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    const unsigned base = (threadIdx.x & 63) * 16;
    if (RD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(base), "n"(j * 1024));
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a;
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {                        // 8 MFMAs per iteration
            if (RD) asm volatile("s_waitcnt lgkmcnt(7)");
            MFMA(acc[q % NACC], r[q], b);
            if (RD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[q]) : "v"(base), "n"(q * 1024 + 8192));
#pragma unroll
            for (int j = 0; j < PLAINV; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(q + j) % 8]));
            // tanh groups: TANH4 per 4 MFMAs, spread
            const int t_before = (q * TANH4) / 4, t_after = ((q + 1) * TANH4) / 4;
#pragma unroll
            for (int t = t_before; t < t_after; ++t) {
                float &x = v[t % 8];
                asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0\n\tv_fma_f32 %0, %0, -2.0, 1.0" : "+v"(x));
                if (t & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(t >> 1) % 4]) : "v"(v[t % 8]), "v"(v[(t - 1) % 8]));
            }
        }
    }
    if (RD) asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = clock64();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
    float s = 0;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    for (int i = 0; i < 8; ++i) s += v[i] + __uint_as_float(r[i][0]);
    for (int i = 0; i < 4; ++i) s += __uint_as_float(pk[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int TANH4, int RD, int PLAINV>
void run(int waves_per_simd, float *out, long long *cyc)
{
    const int iters = 500;
    hipLaunchKernelGGL((k<NACC, TANH4, RD, PLAINV>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NACC, TANH4, RD, PLAINV>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("acc %d  tanh/4mfma %d  ds_read/mfma %d  v_fma/mfma %d  waves/simd %d : %6.1f cycles per MFMA per wave, %6.1f per MFMA on the SIMD\n", NACC, TANH4, RD,
           PLAINV, waves_per_simd, (double)c / iters / 8, (double)c / iters / 8 / waves_per_simd);
}
#include <chrono>
#include <cstring>
// `mfma_multiacc long <mix>`: ~12 s of one mix back to back, for reading clock and socket power with rocm-smi meanwhile
// (profiles/r03_power.txt).  mix 0 = MFMA only, 1 = + one ds_read_b128 per MFMA, 2 = + half a tanh per MFMA as well; one wave per SIMD.
template <int NACC, int TANH4, int RD, int PLAINV>
void soak(float *out, long long *cyc, int waves)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 12.0) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<NACC, TANH4, RD, PLAINV>), dim3(256), dim3(256 * waves), 0, 0, out, cyc, 20000);
        hipDeviceSynchronize();
    }
}
int main(int argc, char **argv)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    if (argc >= 3 && !strcmp(argv[1], "long")) {
        const int mix = atoi(argv[2]), waves = argc >= 4 ? atoi(argv[3]) : 1;
        if (mix == 0) soak<2, 0, 0, 0>(out, cyc, waves);
        else if (mix == 1) soak<2, 0, 1, 0>(out, cyc, waves);
        else soak<2, 2, 1, 0>(out, cyc, waves);
        return 0;
    }
    for (int w = 1; w <= 2; ++w) {
        run<1, 0, 0, 0>(w, out, cyc); run<2, 0, 0, 0>(w, out, cyc); run<4, 0, 0, 0>(w, out, cyc);
        run<1, 0, 1, 0>(w, out, cyc); run<2, 0, 1, 0>(w, out, cyc); run<4, 0, 1, 0>(w, out, cyc);
        run<2, 0, 0, 2>(w, out, cyc); run<2, 0, 0, 4>(w, out, cyc); run<2, 0, 0, 6>(w, out, cyc); run<4, 0, 0, 4>(w, out, cyc);
        run<2, 0, 1, 2>(w, out, cyc); run<2, 0, 1, 4>(w, out, cyc); run<4, 0, 1, 4>(w, out, cyc);
        run<1, 2, 1, 0>(w, out, cyc); run<2, 1, 1, 0>(w, out, cyc); run<2, 2, 1, 0>(w, out, cyc); run<2, 3, 1, 0>(w, out, cyc); run<2, 4, 1, 0>(w, out, cyc);
        run<4, 2, 1, 0>(w, out, cyc); run<4, 3, 1, 0>(w, out, cyc); run<4, 4, 1, 0>(w, out, cyc);
        run<2, 2, 0, 0>(w, out, cyc); run<2, 4, 0, 0>(w, out, cyc);
    }
    return 0;
}
