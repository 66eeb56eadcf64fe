// copy-kernel sweep for dn_stream_copy: GB/s (read + write) of 1 GiB float copies, by variant
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_stride(const v4f *__restrict__ src, v4f *__restrict__ dst, long long n16)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// contiguous chunk per workgroup (each block walks its own 1/grid of the buffer)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const v4f *__restrict__ src, v4f *__restrict__ dst, long long n16)
{
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long long b0 = (long long)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    long long i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < b1; i += UNROLL * 256) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
    }
    for (; i < b1; i += 256) dst[i] = src[i];
}
template <typename F> double run(F launch, long long bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * 20 / (ms * 1e-3) / 1e9;
}
int main()
{
    const long long bytes = 1ll << 30, n16 = bytes / 16;
    v4f *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
    for (int wg : {4, 8, 16, 32, 64}) {
        const unsigned g = 256 * wg;
        printf("wg/CU %2d: stride u4 nt %5.0f | u4 %5.0f | u8 nt %5.0f | u8 %5.0f | u2 %5.0f | u1 %5.0f || chunk u4 nt %5.0f | u4 %5.0f | u8 %5.0f\n", wg,
               run([&] { hipLaunchKernelGGL((k_stride<4, true>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_stride<4, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_stride<8, true>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_stride<8, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_stride<2, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_stride<1, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_chunk<4, true>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_chunk<4, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
               run([&] { hipLaunchKernelGGL((k_chunk<8, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes));
    }
    // one element per thread, no loop
    const unsigned gfull = (unsigned)(n16 / 256);
    printf("one float4 per thread (grid %u): %5.0f\n", gfull, run([&] { hipLaunchKernelGGL((k_stride<1, false>), dim3(gfull), dim3(256), 0, 0, s, d, n16); }, bytes));
    double hm = run([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
    printf("hipMemcpyAsync D2D: %5.0f\n", hm);
    return 0;
}
