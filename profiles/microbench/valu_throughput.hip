// Round 5: chip-wide VALU throughput per instruction kind on gfx950, by WALL time (hipEvents), so that the answer does not
// depend on what s_memtime counts.  W waves per SIMD (W = 1, 2, 4, 8), 16 independent instructions of one kind per loop
// iteration (distinct registers), 4000 iterations.  Output: ns per wave-instruction per SIMD and, at the clock the
// chip reports, cycles; plus a single wave's own issue interval.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_throughput valu_throughput.hip
// Run:   ./valu_throughput            the instruction kinds of the step kernels (profiles/r05_valu_throughput.txt, first part)
//        PART2=1 ./valu_throughput    v_cndmask forms, integer / bit operations, scalar adds, pairs of kinds (second part)
//        PART3=1 ./valu_throughput    one kind with different register patterns and dependency distances (third part)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int KIND>
__global__ __launch_bounds__(1024) void k(double *out, int iters)
{
    double d[16]; float f[16];
    for (int i = 0; i < 16; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.0f + threadIdx.x * 1e-3f + i; }
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p[8];
    for (int i = 0; i < 8; ++i) { p[i].x = f[2 * i]; p[i].y = f[2 * i + 1]; }
    for (int it = 0; it < iters; ++it) {
#define OPD(I) \
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[I]) : "v"(d[(I + 5) & 15])); \
        else if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[I]) : "v"(d[(I + 5) & 15])); \
        else if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[I]) : "v"(d[(I + 5) & 15])); \
        else if (KIND == 3) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[I])); \
        else if (KIND == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[I])); \
        else if (KIND == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[I]) : "v"(f[I])); \
        else if (KIND == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[I]) : "v"(d[I])); \
        else if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[I])); \
        else if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 10) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[I])); \
        else if (KIND == 11) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(d[I]), "v"(d[(I + 1) & 15]) : "vcc"); \
        else if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[I]) : "v"(f[(I + 1) & 15])); \
        else if (KIND == 13) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[I]) : "v"(d[(I + 1) & 15])); \
        else if (KIND == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(I) & 7]) : "v"(p[(I + 3) & 7])); \
        else if (KIND == 15) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(I) & 7]) : "v"(p[(I + 3) & 7])); \
        else if (KIND == 16) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(I) & 7]) : "v"(p[(I + 3) & 7])); \
        else if (KIND == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(f[I]) : "v"(f[(I + 1) & 15])); \
        else if (KIND == 18) asm volatile("v_mov_b64 %0, %1" : "=v"(d[I]) : "v"(d[(I + 1) & 15])); \
        else if (KIND == 19) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[I]) : "v"(f[(I + 1) & 15]), "v"(f[(I + 2) & 15])); \
        else if (KIND == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[I]) : "v"(f[(I + 1) & 15])); \
        else if (KIND == 21) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(f[I]), "v"(f[(I + 1) & 15]) : "vcc"); \
        else if (KIND == 22) asm volatile("v_exp_f32 %0, %0" : "+v"(f[I])); \
        else if (KIND == 23) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 24) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc"); \
        else if (KIND == 25) asm volatile("v_fma_f64 %0, %0, %2, %0\n\tv_mov_b32 %1, %3" : "+v"(d[I]), "=v"(f[I]) : "v"(d[(I + 5) & 15]), "v"(f[(I + 1) & 15])); \
        else if (KIND == 26) asm volatile("v_fma_f64 %0, %0, %1, %0\n\ts_add_u32 s20, s20, 1" : "+v"(d[I]) : "v"(d[(I + 5) & 15]) : "s20", "scc"); \
        else if (KIND == 28) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[I]) : "v"(f[(I + 1) & 15]), "v"(f[(I + 2) & 15]) : "vcc"); \
        else if (KIND == 29) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[I]) : "v"(f[(I + 1) & 15]) : "s20", "s21"); \
        else if (KIND == 30) asm volatile("v_cmp_gt_f64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]" : "+v"(f[I]) : "v"(d[(I + 1) & 15]), "v"(d[(I + 2) & 15]), "v"(f[(I + 1) & 15]) : "s20", "s21"); \
        else if (KIND == 31) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(f[I]) : "v"(f[(I + 1) & 15]), "v"(f[(I + 2) & 15])); \
        else if (KIND == 32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 33) asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(d[I]) : "v"(d[(I + 5) & 15])); \
        else if (KIND == 34) asm volatile("v_fma_f64 %0, %0, s[22:23], %0" : "+v"(d[I]) :: "s22", "s23"); \
        else if (KIND == 35) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 36) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 37) asm volatile("v_cvt_f32_f64 %0, %1\n\tv_cvt_f64_f32 %1, %0" : "+v"(f[I]), "+v"(d[I])); \
        else if (KIND == 38) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[I]) : "v"(f[(I + 5) & 15])); \
        else if (KIND == 39) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(f[I]) : "s20"); \
        else if (KIND == 40) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[I]) : "v"(d[(I + 5) & 15]), "v"(d[(I + 7) & 15])); \
        else if (KIND == 41) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[I]) : "v"(d[(I + 5) & 15]), "v"(d[(I + 7) & 15])); \
        else if (KIND == 42) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[I]) : "v"(d[(I + 5) & 15]), "v"(d[(I + 7) & 15]), "v"(d[(I + 9) & 15])); \
        else if (KIND == 43) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[I]) : "v"(d[(I + 5) & 15]), "v"(d[(I + 7) & 15])); \
        else if (KIND == 44) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[I]) : "v"(d[(I + 5) & 15]), "v"(d[(I + 7) & 15])); \
        else if (KIND == 45) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[0]) : "v"(d[1])); \
        else if (KIND == 46) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[I]) : "v"(f[(I + 5) & 15]), "v"(f[(I + 7) & 15]), "v"(f[(I + 9) & 15])); \
        else if (KIND == 47) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[1])); \
        else if (KIND == 48) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[I]) : "v"(d[(I + 1) & 15]), "v"(d[(I + 2) & 15])); \
        else if (KIND == 49) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[I]) : "v"(d[(I + 15) & 15]), "v"(d[(I + 14) & 15])); \
        else if (KIND == 50) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[I]) : "v"(d[(I + 13) & 15]), "v"(d[(I + 12) & 15])); \
        else if (KIND == 51) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[I]) : "v"(d[(I + 8) & 15]), "v"(d[(I + 10) & 15])); \
        else if (KIND == 27) asm volatile("v_fma_f64 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %3, %1" : "+v"(d[I]), "+v"(f[I]) : "v"(d[(I + 5) & 15]), "v"(f[(I + 5) & 15]));
        OPD(0) OPD(1) OPD(2) OPD(3) OPD(4) OPD(5) OPD(6) OPD(7) OPD(8) OPD(9) OPD(10) OPD(11) OPD(12) OPD(13) OPD(14) OPD(15)
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += d[i] + f[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, double *out, int per_iter, double ghz)
{
    const int iters = 4000;
    printf("%-22s", name);
    for (int w : {1, 2, 4, 8}) {
        // w waves per SIMD: blocks of 256 threads (one wave per SIMD each), w blocks per CU
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)iters * 16 * per_iter * w;
        const double ns = ms * 1e6 / inst_per_simd;
        printf("  W=%d: %5.2f ns = %5.2f cyc", w, ns, ns * ghz);
    }
    printf("   (per wave-instruction per SIMD)\n"); fflush(stdout);
}
int main()
{
    double *out;
    (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 8);
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("# clock reported %.3f GHz; cycles = ns x that (the loaded chip may run lower)\n", ghz);
    if (getenv("PART3")) {
        run<42>("fma_f64 dst!=src (4 regs)", out, 1, ghz); run<43>("fma_f64 dst==src0", out, 1, ghz); run<44>("fma_f64 dst==src2", out, 1, ghz);
        run<41>("mul_f64 3 regs d5 d7", out, 1, ghz); run<48>("mul_f64 reads I+1,I+2 (old)", out, 1, ghz); run<49>("mul_f64 reads I-1,I-2 (RAW 1,2)", out, 1, ghz);
        run<50>("mul_f64 reads I-3,I-4", out, 1, ghz); run<51>("mul_f64 reads I-8,I-6", out, 1, ghz);
        run<45>("mul_f64 serial chain", out, 1, ghz); run<46>("fma_f32 dst!=src", out, 1, ghz); run<47>("mul_f32 serial chain", out, 1, ghz);
        return 0;
    }
    if (getenv("PART2")) {
        run<12>("v_cndmask vcc(stale)", out, 1, ghz); run<29>("v_cndmask_e64 sgpr", out, 1, ghz); run<28>("cmp_f32+cndmask (pair)", out, 1, ghz);
        run<30>("cmp_f64 s+cndmask (pair)", out, 1, ghz); run<31>("v_bfi_b32", out, 1, ghz); run<32>("v_max_f32", out, 1, ghz);
        run<33>("v_fma_f64 const1.0", out, 1, ghz); run<34>("v_fma_f64 sgpr src", out, 1, ghz); run<40>("v_fmac_f64 3 regs", out, 1, ghz); run<41>("v_mul_f64 3 regs", out, 1, ghz);
        run<35>("v_add_u32", out, 1, ghz); run<36>("v_lshl_add_u32", out, 1, ghz);
        run<37>("cvt f32<-f64<-f32 (pair)", out, 1, ghz); run<38>("v_mul_lo_u32", out, 1, ghz); run<39>("v_readlane_b32", out, 1, ghz);
        run<24>("s_add_u32", out, 1, ghz); run<25>("fma_f64+v_mov (pair)", out, 1, ghz); run<26>("fma_f64+s_add (pair)", out, 1, ghz); run<27>("fma_f64+fma_f32 (pair)", out, 1, ghz);
        return 0;
    }
    run<0>("v_fma_f64", out, 1, ghz); run<1>("v_mul_f64", out, 1, ghz); run<2>("v_add_f64", out, 1, ghz); run<13>("v_max_f64", out, 1, ghz);
    run<3>("v_rsq_f64", out, 1, ghz); run<4>("v_rcp_f64", out, 1, ghz);
    run<5>("v_cvt_f64_f32", out, 1, ghz); run<6>("v_cvt_f32_f64", out, 1, ghz); run<11>("v_cmp_gt_f64", out, 1, ghz);
    run<8>("v_fma_f32", out, 1, ghz); run<9>("v_mul_f32", out, 1, ghz); run<23>("v_add_f32", out, 1, ghz); run<19>("v_med3_f32", out, 1, ghz);
    run<14>("v_pk_fma_f32", out, 1, ghz); run<15>("v_pk_mul_f32", out, 1, ghz); run<16>("v_pk_add_f32", out, 1, ghz);
    run<7>("v_rsq_f32", out, 1, ghz); run<10>("v_sqrt_f32", out, 1, ghz); run<22>("v_exp_f32", out, 1, ghz);
    run<17>("v_mov_b32", out, 1, ghz); run<18>("v_mov_b64", out, 1, ghz); run<12>("v_cndmask_b32", out, 1, ghz); run<20>("v_and_b32", out, 1, ghz);
    run<21>("v_cmp_gt_f32", out, 1, ghz); run<24>("s_add_u32", out, 1, ghz);
    run<25>("fma_f64+v_mov (pair)", out, 1, ghz); run<26>("fma_f64+s_add (pair)", out, 1, ghz); run<27>("fma_f64+fma_f32 (pair)", out, 1, ghz);
    return 0;
}
