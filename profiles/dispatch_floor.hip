// Launch-floor microbenchmark (round 2): launch-to-launch time of an empty kernel, of stamp kernels and of the memory skeleton of dn_step over
// grid / block shapes, and the spread of wave start times.  Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dispatch_floor profiles/dispatch_floor.hip
// Result: profiles/r02_dispatch_floor.txt (an empty kernel costs 2.7-3.3 us launch to launch in a stream; that is the floor under dn_step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void k_empty(long long *ts) {}
template <int VG>
__global__ __launch_bounds__(1024) void k_stamp(long long *ts)
{
    long long t = wall_clock64();
    // keep VG registers alive so the wave is allocated that many VGPRs
    float v[VG];
#pragma unroll
    for (int k = 0; k < VG; ++k) v[k] = (float)(threadIdx.x + k);
    float acc = 0;
#pragma unroll
    for (int k = 0; k < VG; ++k) acc += v[k] * v[(k * 7 + 3) % VG];
    if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ts[w] = t + (acc > 1e30f ? 1 : 0); }
}
// memory skeleton of one step over 64-drone tiles; a WG has `blockDim/64` waves, only wave 0 does the memory work when ALL==0
struct P { float4 *g[6]; const float4* act; float* obs; float* rew; unsigned char* done; unsigned char* tr; int* found; long long n; long long *ts; };
template<int WORK>
__global__ __launch_bounds__(256) void k_skel(P p)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv != 0) return;
    long long i = (long long)blockIdx.x*64 + lane;
    float4 a = p.act[i];
    float4 v[6];
    for (int k=0;k<6;++k) v[k]=p.g[k][i];
    double acc = a.x + a.y;
    for (int k=0;k<6;++k) acc += v[k].x*v[k].y + v[k].z*v[k].w;
    for (int j=0;j<WORK;++j) acc = acc*1.0000001 + 1e-9;
    float f=(float)acc;
    for (int k=0;k<6;++k) p.g[k][i]=make_float4(v[k].x+f*1e-20f,v[k].y,v[k].z,v[k].w);
    float* row = p.obs + i*13;
    for (int k=0;k<13;++k) row[k]=f+k;
    p.rew[i]=f; p.done[i]=f>1e30f; p.tr[i]=0; p.found[i]=(int)f;
}
template<typename F> float timeit(F f, int iters){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); for(int i=0;i<300;++i) f(); hipDeviceSynchronize(); hipEventRecord(a,0); for(int i=0;i<iters;++i) f(); hipEventRecord(b,0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); return ms*1e3f/iters; }
int main(int argc,char**argv){
    long long n = 32768;
    P p; p.n=n;
    for(int k=0;k<6;++k){ CK(hipMalloc(&p.g[k], n*16)); CK(hipMemset(p.g[k],0,n*16)); }
    float4* act; CK(hipMalloc(&act,n*16)); CK(hipMemset(act,0,n*16)); p.act=act;
    CK(hipMalloc(&p.obs,n*52)); CK(hipMalloc(&p.rew,n*4)); CK(hipMalloc(&p.done,n)); CK(hipMalloc(&p.tr,n)); CK(hipMalloc(&p.found,n*4));
    long long *ts; CK(hipMalloc(&ts, 1<<20)); p.ts = ts;
    std::vector<long long> h(1<<17);
    int it = 5000;
    // warm the clocks
    timeit([&]{ hipLaunchKernelGGL(k_skel<2000>, dim3(512), dim3(64), 0, 0, p); }, 20000);
    const int shapes[][2] = {{512,64},{512,128},{512,192},{512,256},{256,128},{256,256},{128,256},{1024,64},{2048,64},{256,64},{128,64},{64,64},{8,64},{1,64}};
    for (auto &s : shapes) {
        int grid=s[0], blk=s[1];
        float te = timeit([&]{ hipLaunchKernelGGL(k_empty, dim3(grid), dim3(blk), 0, 0, ts); }, it);
        float t8 = timeit([&]{ hipLaunchKernelGGL(k_stamp<8>, dim3(grid), dim3(blk), 0, 0, ts); }, it);
        float t96 = timeit([&]{ hipLaunchKernelGGL(k_stamp<96>, dim3(grid), dim3(blk), 0, 0, ts); }, it);
        hipDeviceSynchronize();
        int waves = grid * (blk/64);
        long long spread8=0, spread96=0;
        for (int rep=0; rep<5; ++rep) {
            hipLaunchKernelGGL(k_stamp<8>, dim3(grid), dim3(blk), 0, 0, ts); hipDeviceSynchronize();
            hipMemcpy(h.data(), ts, waves*8, hipMemcpyDeviceToHost);
            auto mm = std::minmax_element(h.begin(), h.begin()+waves); spread8 = std::max(spread8, *mm.second - *mm.first);
            hipLaunchKernelGGL(k_stamp<96>, dim3(grid), dim3(blk), 0, 0, ts); hipDeviceSynchronize();
            hipMemcpy(h.data(), ts, waves*8, hipMemcpyDeviceToHost);
            mm = std::minmax_element(h.begin(), h.begin()+waves); spread96 = std::max(spread96, *mm.second - *mm.first);
        }
        printf("grid %5d x %4d thr (%5d waves): empty %.3f us | stamp<8> %.3f us spread %lld ns | stamp<96> %.3f us spread %lld ns\n", grid, blk, waves, te, t8, spread8*10, t96, spread96*10);
    }
    // skeleton: 512 tiles, one working wave per WG, extra idle waves that exit at once
    for (int blk : {64,128,256}) {
        printf("skeleton 512 x %3d: work0 %.3f  work500 %.3f  work1000 %.3f  work2000 %.3f us\n", blk,
               timeit([&]{ hipLaunchKernelGGL(k_skel<0>, dim3(512), dim3(blk), 0, 0, p); }, it),
               timeit([&]{ hipLaunchKernelGGL(k_skel<500>, dim3(512), dim3(blk), 0, 0, p); }, it),
               timeit([&]{ hipLaunchKernelGGL(k_skel<1000>, dim3(512), dim3(blk), 0, 0, p); }, it),
               timeit([&]{ hipLaunchKernelGGL(k_skel<2000>, dim3(512), dim3(blk), 0, 0, p); }, it));
    }
    // Round 3: the same kernels as nodes of a hipGraph (64 dependent kernel nodes, replayed back to back): the period per node is the
    // GPU-side dependent-kernel boundary, with no host launch cadence in it (the eager loops above are host bound below ~3 us).
    {
        hipStream_t st; CK(hipStreamCreate(&st));
        auto graph_period = [&](auto launch, const char *name, int grid, int blk) -> int {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 64; ++i) launch(st);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            const int reps = 200;
            CK(hipEventRecord(a, st));
            for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("graph replay, 64 dependent nodes  %-28s grid %4d x %3d: %.3f us per node\n", name, grid, blk, ms * 1e3f / (reps * 64));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            return 0;
        };
        for (int blk : {64, 192, 256}) {
            if (graph_period([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(512), dim3(blk), 0, s, ts); }, "empty", 512, blk)) return 1;
            if (graph_period([&](hipStream_t s) { hipLaunchKernelGGL(k_stamp<8>, dim3(512), dim3(blk), 0, s, ts); }, "stamp<8>", 512, blk)) return 1;
            if (graph_period([&](hipStream_t s) { hipLaunchKernelGGL(k_skel<0>, dim3(512), dim3(blk), 0, s, p); }, "memory skeleton (work 0)", 512, blk)) return 1;
            if (graph_period([&](hipStream_t s) { hipLaunchKernelGGL(k_skel<500>, dim3(512), dim3(blk), 0, s, p); }, "memory skeleton (work 500)", 512, blk)) return 1;
        }
        if (graph_period([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, ts); }, "empty", 1, 64)) return 1;
        if (graph_period([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, ts); }, "empty", 256, 256)) return 1;
    }
    return 0;
}
