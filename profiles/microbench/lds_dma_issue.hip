// Round 3: what does ONE weight piece (64 lanes x 16 B = 1 KB, global memory -> LDS) cost the wave that requests it, by instruction form,
// in the policy kernels' setting: per "chunk" a wave runs 32 K-steps (one ds_read_b128 of an A fragment + one v_mfma_f32_32x32x16_bf16
// on one accumulator each) and requests 8 pieces of the workgroup's next 32 KB chunk (4 waves), one every fourth K-step; chunk end =
// s_waitcnt vmcnt(0) + s_barrier.  Every workgroup streams the same 1 MB of weights (L2-resident), as dn_mlp_lds_kernel's do.
//   FORM 0: no pieces (the MFMA / ds_read floor)
//   FORM 1: global_load_lds_dwordx4 v[lo:hi], off            (64-bit address per lane: what dn_mlp.hip issues)
//   FORM 2: global_load_lds_dwordx4 voff, s[base:base+1]     (scalar base + 32-bit lane offset)
//   FORM 3: buffer_load_dwordx4 voff, s[rsrc], 0 offen lds   (buffer resource + 32-bit lane offset)
//   FORM 4: global_load_dwordx4 into 4 VGPRs (three pieces in flight), ds_write_b128 ten K-steps later (register staging)
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/r3/lds_dma_issue profiles/microbench/lds_dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int FORM>
__global__ __launch_bounds__(512) void k(const uint4 *__restrict__ w, float *out, long long *cyc, int iters)
{
    extern __shared__ uint4 lds[];                          // 2 x 32 KB chunk buffers
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;                         // 4 or 8 waves
    const int per = 32 / nw;                                // pieces per wave per chunk
    f32x16 acc = {};
    u32x4 b;
    for (int i = 0; i < 4; ++i) b[i] = 0x3f803f80u;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    u32x4 r[4];
    const unsigned rd = lane * 16;
    // buffer resource over the 1 MB of weights
    const unsigned long long base = (unsigned long long)w;
    unsigned rs0 = (unsigned)base, rs1 = (unsigned)(base >> 32), rs2 = 1u << 20, rs3 = 0x00020000u;   // raw buffer, DATA_FORMAT = 32
    rs0 = __builtin_amdgcn_readfirstlane(rs0); rs1 = __builtin_amdgcn_readfirstlane(rs1);
    typedef unsigned u4s __attribute__((ext_vector_type(4)));
    u32x4 stage[3] = {};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const unsigned buf = (it & 1) * 32768u;             // read this buffer, fill the other
        const unsigned chunk = (unsigned)(it & 31) * 32768u;  // byte offset of the chunk in the weights
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(r[j]) : "v"(rd + buf + j * 1024));
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            asm volatile("s_waitcnt lgkmcnt(3)");
            MFMA(acc, r[kk & 3], b);
            asm volatile("ds_read_b128 %0, %1" : "=v"(r[kk & 3]) : "v"(rd + buf + ((kk + 4) & 31) * 1024));
            const int piece = kk >> 2;                      // 0..7; with 8 waves only the first 4 are issued
            if ((kk & 3) == 1 && piece < per) {
                const unsigned frag = (unsigned)(wave * per + piece);           // fragment (1 KB) of the chunk
                const unsigned goff = chunk + frag * 1024u + lane * 16u;
                const unsigned ldst = (buf ^ 32768u) + frag * 1024u;
                if (FORM == 1) {
                    const uint4 *src = (const uint4 *)((const char *)w + goff);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldst) : "memory");
                } else if (FORM == 2) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(goff), "s"(w), "s"(ldst) : "memory");
                } else if (FORM == 3) {
                    asm volatile("s_mov_b32 s40, %1\n\ts_mov_b32 s41, %2\n\ts_mov_b32 s42, %3\n\ts_mov_b32 s43, %4\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                                 "buffer_load_dwordx4 %0, s[40:43], 0 offen lds"
                                 :: "v"(goff), "s"(rs0), "s"(rs1), "s"(rs2), "s"(rs3), "s"(ldst) : "memory", "s40", "s41", "s42", "s43");
                } else if (FORM == 4) {
                    const uint4 *src = (const uint4 *)((const char *)w + goff);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage[piece % 3]) : "v"(src) : "memory");
                }
            }
            // FORM 4: piece j (requested at K-step 4j + 1) is written at K-step 4j + 11: two younger requests in flight -> vmcnt(2)
            if (FORM == 4 && (kk & 3) == 3 && kk >= 11 && (kk - 11) / 4 < per) {
                const int j = (kk - 11) / 4;
                const unsigned frag = (unsigned)(wave * per + j);
                if (j + 2 < per) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (j + 1 < per) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("ds_write_b128 %0, %1" :: "v"((buf ^ 32768u) + frag * 1024u + lane * 16u), "v"(stage[j % 3]) : "memory");
            }
        }
        if (FORM == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < per && 4 * j + 11 > 31) {
                    const unsigned frag = (unsigned)(wave * per + j);
                    if (j + 1 < per) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("ds_write_b128 %0, %1" :: "v"((buf ^ 32768u) + frag * 1024u + lane * 16u), "v"(stage[j % 3]) : "memory");
                }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(r[0][0]) + __uint_as_float(stage[0][0]) + __uint_as_float(stage[1][0]) + __uint_as_float(stage[2][0]);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int FORM>
void run(const char *what, const uint4 *w, float *out, long long *cyc, int waves)
{
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<FORM>), dim3(256), dim3(64 * waves), 65536, 0, w, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-62s %d waves/workgroup: %7.1f cycles per chunk (32 MFMA = 1024)\n", what, waves, (double)c / iters);
}
int main()
{
    uint4 *w; float *out; long long *cyc;
    (void)hipMalloc(&w, 1 << 20); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(w, 0x3c, 1 << 20);
    for (int waves : {4, 8}) {
        run<0>("no pieces", w, out, cyc, waves);
        run<1>("global_load_lds_dwordx4 v[a:a+1], off", w, out, cyc, waves);
        run<2>("global_load_lds_dwordx4 voff, s[base:base+1]", w, out, cyc, waves);
        run<3>("buffer_load_dwordx4 voff, s[rsrc:rsrc+3], 0 offen lds", w, out, cyc, waves);
        run<4>("global_load_dwordx4 -> VGPR, ds_write_b128 10 K-steps later", w, out, cyc, waves);
    }
    return 0;
}
