// Round 3: is the FIRST MFMA read of a B operand slower than the second?  (dn_mlp_sac_lds_kernel: the first layer-2 tile and the
// head tile -- the first consumers of a freshly produced activation set -- ran 3x slower per MFMA than the tiles behind them.)
// One wave per SIMD; 16 B operands (u32x4) produced by method METHOD from DATA; then PASSES passes of 16 K-steps x 3 chained MFMAs
// (the float32-grade K-step: lo x hi, hi x lo, hi x hi on one accumulator), clock64() around each pass.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/r3/mfma_first_use profiles/microbench/mfma_first_use.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))

__device__ inline unsigned pack_bf16(float a, float b)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// DATA: 0 = ones, 1 = uniform random normal-range, 2 = ReLU-like (half zeros), 3 = tiny values (bf16 denormals), 4 = hi/lo split of ReLU-like
template <int DATA>
__device__ inline void make(const float *src, int lane, int i, u32x4 &h, u32x4 &l)
{
    for (int q = 0; q < 4; ++q) {
        float v0 = src[(i * 8 + 2 * q) * 64 + lane], v1 = src[(i * 8 + 2 * q + 1) * 64 + lane];
        if (DATA == 0) { v0 = 1.0f; v1 = 1.0f; }
        if (DATA == 2 || DATA == 4) { v0 = fmaxf(v0 - 0.5f, 0.0f); v1 = fmaxf(v1 - 0.5f, 0.0f); }
        if (DATA == 3) { v0 *= 1e-39f; v1 *= 1e-39f; }
        const unsigned hi = pack_bf16(v0, v1);
        h[q] = hi;
        if (DATA == 4) {
            const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xffff0000u);
            l[q] = pack_bf16(r0, r1);
        } else l[q] = hi;
    }
}
template <int DATA, int PASSES>
__global__ __launch_bounds__(256) void k(const float *src, float *out, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    u32x4 bh[16], bl[16], a0, a1;
    for (int q = 0; q < 4; ++q) { a0[q] = 0x3f803f80u; a1[q] = 0x3c003c00u; }
#pragma unroll
    for (int i = 0; i < 16; ++i) make<DATA>(src, lane, i, bh[i], bl[i]);
    f32x16 acc = {};
    long long t[PASSES + 1];
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        t[p] = clock64();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            MFMA(acc, a1, bh[kk]);
            MFMA(acc, a0, bl[kk]);
            MFMA(acc, a0, bh[kk]);
        }
    }
    t[PASSES] = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int p = 0; p < PASSES; ++p) cyc[p] = t[p + 1] - t[p];
}
template <int DATA>
void run(const char *what, const float *src, float *out, long long *cyc, int wgs)
{
    constexpr int P = 4;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<DATA, P>), dim3(wgs), dim3(256), 0, 0, src, out, cyc);
    hipDeviceSynchronize();
    long long c[P];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-44s %4d workgroups: cycles per MFMA, passes 1..4: %6.1f %6.1f %6.1f %6.1f\n", what, wgs, c[0] / 48.0, c[1] / 48.0, c[2] / 48.0, c[3] / 48.0);
}
int main()
{
    float *src, *out; long long *cyc;
    const int N = 16 * 8 * 64;
    float h[N];
    unsigned x = 12345u;
    for (int i = 0; i < N; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x >> 8) * (1.0f / 16777216.0f); }
    hipMalloc(&src, sizeof h); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    for (int wgs : {1, 256}) {
        run<0>("B = ones", src, out, cyc, wgs);
        run<1>("B = uniform (0,1)", src, out, cyc, wgs);
        run<2>("B = ReLU-like (half zeros)", src, out, cyc, wgs);
        run<3>("B = bf16 denormals", src, out, cyc, wgs);
        run<4>("B = hi / lo split of ReLU-like", src, out, cyc, wgs);
    }
    return 0;
}
