// Streaming structure: three weight buffers, DMA two chunks ahead, an 8-deep fragment ring that runs across the chunk
// boundary (never drained), one barrier per chunk that waits only for the older DMA pieces and the partial stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RING = 8;
template <bool DMA, int WRITES, int VALU, bool DEFER, bool SPREAD, int KEEP>
__global__ __launch_bounds__(512) void k(const uint4 *w, float *out, long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[3 * 2048 + 2048];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave >> 2;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = make_uint4(i, 1, 2, 3);
    f32x16 acc = {}, prev = {};
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)1.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 1e-3f;
    __syncthreads();
    uint4 ring[RING];
#pragma unroll
    for (int kk = 0; kk < RING; ++kk) ring[kk] = lds[(half * 16 + kk) * 64 + lane];
    const long long t0 = clock64();
    for (int it = 0; it < iters; it += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {                 // chunk it + u in buffer u
            const uint4 *cur = lds + u * 2048;
            const uint4 *nxt = lds + ((u + 1) % 3) * 2048;
            const uint4 *gsrc = w + (size_t)((it + u) % 25) * 2048 + wave * 4 * 64 + lane;
            const unsigned lds_dst = (unsigned)(uintptr_t)(lds + ((u + 2) % 3) * 2048 + wave * 4 * 64);
            if (DMA && !SPREAD) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const uint4 a = ring[kk % RING];
                ring[kk % RING] = kk + RING < 16 ? cur[(half * 16 + kk + RING) * 64 + lane] : nxt[(half * 16 + kk + RING - 16) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, acc, 0, 0, 0);
                if (DMA && SPREAD && (kk & 3) == 1) {
                    unsigned keep;
                    const int q = kk >> 2;
                    if (q == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
                    if (q == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
                    if (q == 2) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
                    if (q == 3) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
                }
                if (WRITES && DEFER && kk >= 2 && kk < 2 + WRITES && (WRITES == 2 || (wave >> 2) != (u & 1))) {
                    float4 *xb = reinterpret_cast<float4 *>(lds + 6144) + (wave & 3) * 512 + (u & 1) * 256;
                    const int j = kk - 2;
                    xb[j * 64 + lane] = make_float4(prev[4 * j], prev[4 * j + 1], prev[4 * j + 2], prev[4 * j + 3]);
                }
#pragma unroll
                for (int j = 0; j < VALU; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(kk + j) % 8]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DEFER) { prev = acc; for (int i = 0; i < 16; ++i) acc[i] = 0.0f; }
            if (WRITES && !DEFER) {
                float4 *xb = reinterpret_cast<float4 *>(lds + 6144) + (wave & 3) * 512 + (u & 1) * 256;
                if (WRITES == 2 || (wave >> 2) == (u & 1)) {
#pragma unroll
                    for (int j = 0; j < WRITES; ++j) xb[j * 64 + lane] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
                }
            }
            if (DMA && DEFER && KEEP == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(8)\n\ts_barrier" ::: "memory");
            else if (DMA && DEFER) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(8)\n\ts_barrier" ::: "memory");
            else if (DMA && KEEP == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (DMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i] + prev[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < RING; ++i) s += ring[i].x;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <bool DMA, int WRITES, int VALU, bool DEFER = false, bool SPREAD = false, int KEEP = 4>
void run(const uint4 *w, float *out, long long *cyc)
{
    const int iters = 1998;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<DMA, WRITES, VALU, DEFER, SPREAD, KEEP>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("stream: dma %d  writes %d  v_exp/MFMA %d  defer %d  spread %d  vmcnt(%d): %7.1f cycles per interval (MFMA floor 1024)\n", DMA, WRITES, VALU, DEFER, SPREAD, KEEP, (double)c / iters);
}
int main()
{
    uint4 *w; float *out; long long *cyc;
    hipMalloc(&w, 8 << 20); hipMemset(w, 1, 8 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run<true, 0, 0, false, false, 4>(w, out, cyc); run<true, 0, 0, false, false, 0>(w, out, cyc);
    run<true, 2, 1, false, false, 4>(w, out, cyc); run<true, 2, 1, false, false, 0>(w, out, cyc);
    run<true, 2, 1, true, false, 4>(w, out, cyc); run<true, 2, 1, true, false, 0>(w, out, cyc);
    run<true, 2, 1, true, true, 4>(w, out, cyc); run<true, 2, 1, true, true, 0>(w, out, cyc);
    return 0;
}
