// LDS read throughput per CU for the fragment pattern (64 lanes x 16 B lane-linear ds_read_b128), 8 waves per CU,
// alone / with MFMAs consuming them / with an LDS-DMA stream landing in the other buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool MFMA, bool DMA, int WRITES>
__global__ __launch_bounds__(512) void k(const uint4 *w, float *out, long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * 2048 + 2048];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 6144; i += 512) lds[i] = make_uint4(i, 1, 2, 3);
    f32x16 acc = {};
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)1.0f;
    uint4 x = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const uint4 *cur = lds + (it & 1) * 2048;
        if (DMA) {
            const uint4 *gsrc = w + (size_t)(it % 25) * 2048 + wave * 4 * 64 + lane;
            const unsigned lds_dst = (unsigned)(uintptr_t)(lds + ((it + 1) & 1) * 2048 + wave * 4 * 64);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
        uint4 r[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) r[kk] = cur[((wave >> 2) * 16 + kk) * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r[kk]), b, acc, 0, 0, 0);
            else { x.x ^= r[kk].x; x.y ^= r[kk].y; x.z ^= r[kk].z; x.w ^= r[kk].w; }
        }
        if (WRITES) {
            float4 *xb = reinterpret_cast<float4 *>(lds + 4096) + (wave & 3) * 512 + (it & 1) * 256;
#pragma unroll
            for (int j = 0; j < WRITES; ++j) xb[j * 64 + lane] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    float s = (float)(x.x ^ x.y ^ x.z ^ x.w);
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <bool MFMA, bool DMA, int WRITES>
void run(const uint4 *w, float *out, long long *cyc)
{
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MFMA, DMA, WRITES>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mfma %d dma %d writes/wave %d: %7.1f cycles per interval (8 waves x 16 ds_read_b128 [+16 MFMA each]; MFMA floor 1024)\n", MFMA, DMA, WRITES, (double)c / iters);
}
int main()
{
    uint4 *w; float *out; long long *cyc;
    hipMalloc(&w, 8 << 20); hipMemset(w, 1, 8 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run<false, false, 0>(w, out, cyc); run<false, true, 0>(w, out, cyc); run<true, false, 0>(w, out, cyc); run<true, true, 0>(w, out, cyc);
    run<true, false, 4>(w, out, cyc); run<true, true, 4>(w, out, cyc); run<true, true, 2>(w, out, cyc);
    return 0;
}
