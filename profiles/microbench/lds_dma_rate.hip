// LDS-DMA throughput per CU when every CU streams the same L2-resident 0.8 MB, in 32 KB chunks, 8 waves x 4 pieces.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
template <int MODE>   // 0: LDS-DMA, barrier per chunk (vmcnt 0); 1: LDS-DMA, one chunk ahead kept in flight (vmcnt 4); 2: global_load_dwordx4 -> regs (no LDS); 3: LDS-DMA issued by 2 of the 8 waves (16 pieces each)
__global__ __launch_bounds__(512) void k(const uint4 *w, float *out, long long *cyc, int chunks, int total_chunks, int same)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[3 * 2048 + 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4 *base = w + (same ? 0 : (size_t)(blockIdx.x % 8) * 0);
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = clock64();
    for (int c = 0; c < chunks; ++c) {
        const uint4 *src = base + (size_t)(c % total_chunks) * 2048;
        uint4 *dst = lds + (c % 3) * 2048;
        if (MODE == 2) {
            uint4 r[4];
            for (int j = 0; j < 4; ++j) r[j] = src[(wave * 4 + j) * 64 + lane];
            for (int j = 0; j < 4; ++j) { acc.x ^= r[j].x; acc.y ^= r[j].y; acc.z ^= r[j].z; acc.w ^= r[j].w; }
        } else if (MODE == 3) {
            if (wave < 2) {
                for (int q = 0; q < 4; ++q) {
                    const int f = wave * 16 + q * 4;
                    const uint4 *gsrc = src + f * 64 + lane;
                    const unsigned lds_dst = (unsigned)(uintptr_t)(dst + f * 64);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            const int f = wave * 4;
            const uint4 *gsrc = src + f * 64 + lane;
            const unsigned lds_dst = (unsigned)(uintptr_t)(dst + f * 64);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    __syncthreads();
    const uint4 v = lds[threadIdx.x];
    out[blockIdx.x * 512 + threadIdx.x] = (float)(v.x ^ acc.x ^ acc.y ^ acc.z ^ acc.w);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const uint4 *w, float *out, long long *cyc, int blocks, int total_chunks)
{
    const int chunks = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, w, out, cyc, chunks, total_chunks, 1);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d  blocks %4d  footprint %5d KB: %7.1f cycles per 32 KB chunk = %5.1f B/clk/CU\n", MODE, blocks, total_chunks * 32, (double)c / chunks, 32768.0 * chunks / c);
}
int main()
{
    uint4 *w; float *out; long long *cyc;
    hipMalloc(&w, 64 << 20); hipMemset(w, 1, 64 << 20); hipMalloc(&out, 2048 * 512 * 4); hipMalloc(&cyc, 8);
    for (int blocks : {1, 32, 256, 512}) {
        run<0>(w, out, cyc, blocks, 25); run<1>(w, out, cyc, blocks, 25); run<2>(w, out, cyc, blocks, 25); run<3>(w, out, cyc, blocks, 25);
    }
    run<0>(w, out, cyc, 256, 1); run<1>(w, out, cyc, 256, 1); run<1>(w, out, cyc, 256, 50); run<1>(w, out, cyc, 256, 2000);
    return 0;
}
