// dn_mlp.hip -- the policy / value network of the reference as ONE fused MFMA kernel for gfx950.
//
// Reference: PBDroneSimulator.setup_agent builds SB3 PPO(ActorCriticPolicy, net_arch=dict(pi=[512,512,256],
// vf=[512,512,256]), activation_fn=Tanh) (Sol/Model/PBDroneSimulator.py:251-286): per network
//     obs[13] -> Linear 512 -> Tanh -> Linear 512 -> Tanh -> Linear 256 -> Tanh -> Linear out   (out = 4 actions | 1 value)
// evaluated for every drone at every rollout step.  Through a BLAS library that is four GEMM launches plus three
// element-wise launches per network with the [N, 512] activations making a round trip through HBM between them.
//
// Here one wavefront carries a tile of 32 drones through the whole network and the activations never leave its
// registers, not even for LDS:
//   * the product is formed transposed, H^T = W . X^T, with v_mfma_f32_32x32x16_bf16: the weights are the A
//     operand (rows = output features), the activations the B operand (columns = drones);
//   * the accumulator tile D[feature][drone] leaves lane l holding 16 features of drone (l & 31) -- and a B operand
//     wants lane l to hold 8 K-values of column (l & 31): after bias + tanh + bf16 packing the accumulator registers
//     ARE the next layer's B operands.  The only thing that has to agree is the order of the K index, and that is
//     absorbed by permuting the weights' K columns once on the host (pack_mlp in policy_mfma.py): fragment (mo, kk)
//     of a layer is stored as the 64 x 16 B the 64 lanes load, so every A fragment is one global_load_dwordx4.
// No LDS, no barriers, no inter-wave traffic; the weights (0.8 MB per network in bf16) stream from L2.
//
// Arithmetic: bf16 inputs/weights, float32 accumulation, float32 bias, tanh(x) = 1 - 2 / (1 + e^{2x}) on
// v_exp_f32 / v_rcp_f32 (absolute error ~1e-7, far inside bf16; the 2 log2(e) of the exponent is pre-multiplied into
// the hidden layers' packed weights and biases), float32 output of the head.
#include "dn_internal.h"

#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // a B operand as the four dwords it occupies
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define MLP_DEV __device__ __forceinline__

// A policy kernel may carry a TAIL: work its workgroup does with the head's output before it leaves (dn_fused.hip: the workgroup that
// evaluated the actor for 128 / 64 drones draws their actions and runs their control step -- one launch per closed-loop step).
// NoTail = the plain forward pass (dn_mlp_forward).
struct NoTail {
    static constexpr bool active = false;
};

constexpr int H1 = 512, H2 = 512, H3 = 256;        // net_arch of PBDroneSimulator.py:251-258
constexpr int TILE = 32;                           // drones per wavefront (the N of the 32x32x16 MFMA)

struct MlpNetDev {
    const uint4 *w1, *w2, *w3, *wh;
    const float *b1, *b2, *b3, *bh;
    float *out;
    int out_dim;
};
// (fp32-grade networks, dn_mlp_net.grade = 1: the w* streams hold, per M-tile, the KS hi fragments then the KS lo fragments)
struct MlpArgs {
    MlpNetDev net[2];
    const float *obs;
    const uint8_t *row_mask;       // optional: a tile none of whose drones is flagged writes zeros and skips the network
    long long n;
    int obs_dim;
};

// tanh(h) = 1 - 2 / (1 + 2^(2 log2(e) h)).  The factor 2 log2(e) is folded into the hidden layers' weights and biases
// when they are packed (policy_mfma.pack_layer), so the accumulator already holds x = 2 log2(e) h and the activation is
// v_exp_f32, v_add, v_rcp_f32, v_fma.
MLP_DEV float tanh_fast(float x)
{
    const float e = __builtin_amdgcn_exp2f(x);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// This lane's eight layer-1 inputs k = 8g .. 8g + 7 of drone `row` (zero beyond obs_dim): eight independent loads at a
// clamped index, selected afterwards (a load under `if (k < obs_dim)` compiles to a branch with its own s_waitcnt).
MLP_DEV void load_obs8(const MlpArgs &a, const long long row, const int g, float (&v)[8])
{
    const float *o = a.obs + row * a.obs_dim;
    const int last = a.obs_dim - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const float x = o[k < last ? k : last];
        v[j] = k <= last ? x : 0.0f;
    }
}
// Row (output feature) of accumulator register r in lane group g of M-tile m: the C/D map of the 32x32 MFMA.
MLP_DEV int acc_row(int m, int g, int r) { return 32 * m + 4 * g + (r & 3) + 8 * (r >> 2); }

// The epilogue of an M-tile (16 accumulator values per lane: bias already in, tanh, bf16) costs ~700 VALU cycles, two
// thirds of the tile's 32 MFMAs (1024 cycles on the matrix pipe).  Issued after the MFMAs it would idle the matrix pipe;
// it is spread over the NEXT tile's MFMA stream, one pair of elements every fourth MFMA, so that the matrix pipe is
// never left without a queued MFMA while the wave grinds through 100 VALU instructions.
MLP_DEV unsigned pack2(float a, float b)
{
    bf16x2 pk;
    pk[0] = (__bf16)a;
    pk[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, pk);                 // one v_cvt_pk_bf16_f32
}
// The machine scheduler sinks every LDS fragment read next to the MFMA that consumes it (it minimises live ranges), which
// collapses the software ring below to a depth of one or two: the ISA then reads `ds_read_b128; s_waitcnt lgkmcnt(0); v_mfma`
// and every MFMA waits out a full LDS round trip (measured: 92 cycles per MFMA in the four-wave shape against 32.6 for the same
// stream with the ring kept, profiles/microbench/mfma_multiacc.hip).  A zero-mask sched_barrier after every K-step pins the
// source order across K-steps -- fragment read issued RING steps ahead of its MFMA, one slice of the previous tile's epilogue
// per step -- and leaves the order inside a step to the compiler.
#define MLP_PIN() __builtin_amdgcn_sched_barrier(0)
#ifndef DN_X3_LAZY_MERGE
#define DN_X3_LAZY_MERGE 1    // float32-grade kernel: the partner's partial sums are ADDED where the epilogue consumes them, inside the K-loop, not before it
#endif
#ifndef DN_X3_DMA_SPREAD
#define DN_X3_DMA_SPREAD 2    // float32-grade kernel: LDS-DMA pieces issued this many per K-step, between the MFMAs, instead of in one burst at the
                              // top of the tile (0 = burst).  Measured, interleaved A/B at 32 768 drones: 197.7 (burst) / 191 (1) / 187.7 us (2)
#endif

MLP_DEV void epilogue_pair(const f32x16 &acc, const int q, u32x4 &lo, u32x4 &hi)
{   // accumulator elements 2q, 2q+1 -> one dword of the packed B operand
    const unsigned u = pack2(tanh_fast(acc[2 * q]), tanh_fast(acc[2 * q + 1]));
    if (q < 4) lo[q] = u; else hi[q - 4] = u;
}

MLP_DEV void epilogue(const f32x16 &acc, const bool tanh_on, u32x4 &lo, u32x4 &hi)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        lo[q] = tanh_on ? pack2(tanh_fast(acc[2 * q]), tanh_fast(acc[2 * q + 1])) : pack2(acc[2 * q], acc[2 * q + 1]);
        hi[q] = tanh_on ? pack2(tanh_fast(acc[8 + 2 * q]), tanh_fast(acc[9 + 2 * q])) : pack2(acc[8 + 2 * q], acc[9 + 2 * q]);
    }
}

// The 16-bit operand format of a grade: bfloat16 (grade 0: 8 mantissa bits, float32's range) or float16 (grade 2: 11 mantissa
// bits -- an eighth of bf16's rounding error at the same MFMA rate; tanh activations, |w| << 1 and observations sit far inside
// its range).  Same fragment layout, same MFMA shape.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16> MLP_DEV unsigned pack2t(const float a, const float b)
{
    if (F16) {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        f16x2 pk;
        pk[0] = (_Float16)a;
        pk[1] = (_Float16)b;
        return __builtin_bit_cast(unsigned, pk);             // round to nearest even (v_cvt_pk_f16_f32 on gfx950)
    }
    return pack2(a, b);
}
template <bool F16> MLP_DEV f32x16 mfma16(const uint4 a, const u32x4 b, const f32x16 c)
{
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16> MLP_DEV void epilogue_pair_t(const f32x16 &acc, const int q, u32x4 &lo, u32x4 &hi)
{
    const unsigned u = pack2t<F16>(tanh_fast(acc[2 * q]), tanh_fast(acc[2 * q + 1]));
    if (q < 4) lo[q] = u; else hi[q - 4] = u;
}
template <bool F16> MLP_DEV void epilogue_t(const f32x16 &acc, u32x4 &lo, u32x4 &hi)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) epilogue_pair_t<F16>(acc, q, lo, hi);
}

// One Linear(+Tanh) layer for this wave's 32 drones.  KS = K-steps of 16 input features, MT = M-tiles of 32 output
// features.  in[kk] is the B operand of K-step kk; out[2m], out[2m+1] become K-steps 2m, 2m+1 of the next layer.
// The layer's MT*KS weight fragments are consumed in storage order; a ring of RING fragments (16 B per lane each)
// is kept in flight ahead of the MFMA that uses them, because one fragment is one L2 round trip (~200+ cycles) and an
// MFMA only 32: without the ring the wave waits on every load.
template <int KS, int MT, bool TANH, int RING>
MLP_DEV void layer(const uint4 *__restrict__ w, const float *__restrict__ bias, const u32x4 (&in)[KS], u32x4 (&out)[2 * MT],
                   const int lane)
{
    const int g = lane >> 5;
    constexpr int T = KS * MT;
    constexpr int P = RING < T ? RING : T;
    const uint4 *wl = w + lane;
    uint4 ring[P];
#pragma unroll
    for (int t = 0; t < P; ++t) ring[t] = wl[t * 64];
    MLP_PIN();
    f32x16 prev;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[acc_row(m, g, r)];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int t = m * KS + kk;
            const uint4 a = ring[t % P];
            if (t + P < T) ring[t % P] = wl[(t + P) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, in[kk]), acc, 0, 0, 0);
            // the previous tile's epilogue, one pair of elements every fourth MFMA (KS = 32)
            if (TANH && KS == 32 && m > 0 && (kk & 3) == 3) epilogue_pair(prev, kk >> 2, out[2 * (m - 1)], out[2 * (m - 1) + 1]);
            MLP_PIN();
        }
        if (TANH && KS == 32 && m + 1 < MT) prev = acc;
        else epilogue(acc, TANH, out[2 * m], out[2 * m + 1]);
    }
}

// grid = (tiles, nets), one wavefront per block: every wave is alone on its SIMD and may use the whole register file.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void dn_mlp_kernel(const MlpArgs a)
{
    const int lane = threadIdx.x;
    const int g = lane >> 5, col = lane & 31;
    const MlpNetDev &net = a.net[blockIdx.y];
    const long long row0 = (long long)blockIdx.x * TILE;
    const long long row = row0 + col < a.n ? row0 + col : a.n - 1;      // ragged last tile: shadow the last drone
    if (a.row_mask) {
        // masked forward (e.g. V(terminal_observation), needed only for drones that hit the time limit): one byte per
        // drone; the wave votes, and a tile without a flagged drone costs a 32-byte read and an out_dim-float store
        const bool wanted = row0 + col < a.n && a.row_mask[row0 + col] != 0;
        if (__ballot(wanted) == 0ull) {
            if (g == 0 && row0 + col < a.n)
                for (int j = 0; j < net.out_dim; ++j) net.out[(row0 + col) * net.out_dim + j] = 0.0f;
            return;
        }
    }

    // layer-1 B operand: this drone's observation, K = 16 (obs_dim <= 16, zero padded), lane group g holds k = 8g..8g+7
    u32x4 x0[1];
    {
        const float *o = a.obs + row * a.obs_dim;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * g + 2 * q;
            x0[0][q] = pack2(k < a.obs_dim ? o[k] : 0.0f, k + 1 < a.obs_dim ? o[k + 1] : 0.0f);
        }
    }
    u32x4 h1[H1 / 16];
    layer<1, H1 / 32, true, 16>(net.w1, net.b1, x0, h1, lane);
    u32x4 h2[H2 / 16];
    layer<H1 / 16, H2 / 32, true, 16>(net.w2, net.b2, h1, h2, lane);
    u32x4 h3[H3 / 16];
    layer<H2 / 16, H3 / 32, true, 16>(net.w3, net.b3, h2, h3, lane);

    // head: one M-tile (out_dim <= 32 rows, zero-padded weights), float32 result straight from the accumulator
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = net.bh[acc_row(0, g, r)];
#pragma unroll
    for (int kk = 0; kk < H3 / 16; ++kk) {
        const uint4 w = net.wh[kk * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, h3[kk]), acc, 0, 0, 0);
    }
    if (row0 + col < a.n) {
        float *o = net.out + (row0 + col) * net.out_dim;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = acc_row(0, g, r);
            if (j < net.out_dim) o[j] = acc[r];
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// Shared-weights shape: 4 wavefronts (128 drones) per workgroup stream every weight fragment through LDS ONCE.
//
// In the one-wave shape each of the four waves of a CU pulls every 1 KB fragment through the CU's vector memory path
// (64 B/clk): 4 KB per 32-cycle MFMA step is 128 B/clk, twice what the path delivers, and the kernel runs at a
// quarter of the MFMA rate however deep the loads are pipelined (measured: 41 us per network at 32768 drones with 2
// or with 10 fragments in flight).  Here a fragment crosses that path once per workgroup -- an LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 B land lane-linear, which IS the fragment layout) issued by one of the
// four waves -- and the four waves read it from LDS (ds_read_b128, 128 B/clk).  Weights move in chunks of up to 32
// fragments (one M-tile of a 512-wide layer), double-buffered: the DMA of chunk c+1 is in flight while chunk c is
// multiplied; one `s_waitcnt vmcnt(0)` + `s_barrier` per chunk.
// -----------------------------------------------------------------------------------------------------
constexpr int CHUNK = 32;                                  // fragments (KB) per LDS buffer
constexpr int WAVES = 4;
constexpr int LDS_RING = 8;
constexpr int NBIAS = H1 + H2 + H3 + 32;                      // float32 biases of the four layers, staged in LDS
// accumulator of an M-tile starts from its bias (staged in LDS): rows acc_row(m, g, 4j .. 4j + 3) are four consecutive floats
MLP_DEV void bias_init(const float *lbias, const int m, const int g, f32x16 &acc)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 b = *reinterpret_cast<const float4 *>(lbias + 32 * m + 4 * g + 8 * j);
        acc[4 * j] = b.x; acc[4 * j + 1] = b.y; acc[4 * j + 2] = b.z; acc[4 * j + 3] = b.w;
    }
}

// N (1, 2 or 4) weight pieces of 1 KB, 1 KB apart in global memory and in LDS alike (the instruction's immediate offset applies to both
// addresses): `gbase` = the first piece's fragment (wave-uniform), `lds_dst` = its LDS byte address (wave-uniform -> M0).  The scalar-base
// form (SGPR pair + one 32-bit lane offset) costs the requesting wave ~48 cycles a piece among ds_reads and MFMAs, the 64-bit-address-
// per-lane form ~57 (profiles/r03_lds_dma_issue.txt), and needs no 64-bit vector add per group.
// Inline asm on purpose: hipcc drains a builtin LDS-DMA (s_waitcnt vmcnt(0)) before the next ds_read because it cannot tell the two LDS
// buffers apart, which would expose the whole L2 round trip at the start of every chunk; chunk_barrier() waits for the pieces explicitly.
template <int N>
MLP_DEV void dma_pieces(const uint4 *gbase, const unsigned lds_dst, const int lane)
{
    static_assert(N == 1 || N == 2 || N == 4, "");
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
    const unsigned long long gb = (unsigned long long)(uintptr_t)gbase;
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gb), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
    const unsigned long long gs = ((unsigned long long)ghi << 32) | glo;
    unsigned keep;
    if (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(gs) : "memory");
    else if (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(gs) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(gs) : "memory");
}

MLP_DEV void dma_chunk(const uint4 *__restrict__ src, uint4 *lds, const int nfrag, const int wave, const int lane)
{   // wave w brings in fragments [w*nfrag/4, (w+1)*nfrag/4) of the chunk, contiguous both in global memory and in LDS
    const int per = nfrag / WAVES;                          // 8 (32-fragment chunk) or 4 (16-fragment chunk)
#pragma unroll
    for (int j0 = 0; j0 < CHUNK / WAVES; j0 += 4) {
        if (j0 < per) {
            const int f = wave * per + j0;
            dma_pieces<4>(src + f * 64, (unsigned)(uintptr_t)(lds + f * 64), lane);
        }
    }
}
MLP_DEV void dma_piece(const uint4 *gfrag, uint4 *lds_frag, const int lane)
{   // ONE 1-KB fragment: both addresses wave-uniform
    dma_pieces<1>(gfrag, (unsigned)(uintptr_t)lds_frag, lane);
}
MLP_DEV void chunk_barrier()
{
    __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) & lgkmcnt(0): my DMA pieces landed, my LDS reads done
    __builtin_amdgcn_s_barrier();
}
// Timing builds (-DDN_MLP_STAMP; never shipped): s_memtime before and after every chunk barrier of workgroup 0's waves, dumped by
// dn_launch_mlp to the file named by DN_MLP_STAMP_FILE.  STP_PARAM / STP_ARG thread the per-wave counter through the bodies.
#ifdef DN_MLP_STAMP
__device__ long long g_stamp[8][256];
struct Stamp { int n, wave; bool on; };
MLP_DEV void stamp(Stamp &s)
{
    if (s.on && s.n < 256) g_stamp[s.wave][s.n] = (long long)__builtin_readcyclecounter();
    ++s.n;
}
MLP_DEV void chunk_barrier_stamped(Stamp &s)
{
    stamp(s);
    __builtin_amdgcn_s_waitcnt(0x0070);
    stamp(s);
    __builtin_amdgcn_s_barrier();
    stamp(s);
}
#define STP_PARAM , Stamp &stp
#define STP_ARG , stp
#define CHUNK_BARRIER() chunk_barrier_stamped(stp)
#else
#define STP_PARAM
#define STP_ARG
#define CHUNK_BARRIER() chunk_barrier()
#endif

// A 512-input layer, weights through LDS: one chunk = the 32 fragments of one M-tile, THREE chunk buffers.  BASE = the buffer that holds this
// layer's chunk 0 (requested by the previous layer, landed before its last barrier): chunk m lives in buffer (BASE + m) % 3; `next` = the
// following layer's weights, whose first NEXT_FR fragments are requested during this layer's last chunk into buffer (BASE + MT) % 3.
// The layer is ONE fragment stream: the register ring (LDS_RING fragments ahead of the MFMA that consumes them -- an LDS round trip is
// ~64-128 cycles, an MFMA 32) runs on across the tile boundary.
//
// Until round 6 every M-tile ended with `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` and the next one started with eight ring reads: between a
// tile's last MFMA and the next tile's first the matrix pipe waited out the barrier and an LDS round trip.  Here the chunk barrier sits
// INSIDE the K-loop, at the K-step whose ring refill is the first to read the next chunk (K-step CHUNK - LDS_RING): by then the pieces
// requested at the top of the tile have had ~24 MFMAs to land, the barrier only says so (vmcnt(0); LDS reads stay in flight), the refills
// of the last LDS_RING K-steps read the NEXT chunk, and the next tile's first MFMA follows this tile's last.
// Why three buffers: tile m requests chunk m + 1 at its top.  With two buffers that is the buffer of chunk m - 1, whose last reads a slow
// wave has issued (before the barrier inside tile m - 1) but perhaps not yet been served -- safe only because a DMA piece lands an L2 round
// trip later.  With three it is the buffer of chunk m - 2: every wave consumed its last fragment of that chunk (K-step CHUNK - 1 of tile
// m - 2) before it arrived at the barrier inside tile m - 1, which every wave has passed before any reaches the top of tile m.
// Biases come from LDS (bias_init), one tile ahead, straight into the registers the next tile accumulates in.
MLP_DEV void dma_landed_barrier()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) only
    __builtin_amdgcn_s_barrier();
}
constexpr int NBUF = 3;
template <bool F16, int MT, int BASE, int NEXT_FR>
MLP_DEV void layer_lds_c(const uint4 *__restrict__ w, const float *lbias, const uint4 *__restrict__ next,
                         const u32x4 (&in)[CHUNK], u32x4 (&out)[2 * MT], uint4 *lds, const int wave, const int lane)
{
    constexpr int SYNC = CHUNK - LDS_RING;
    static_assert(CHUNK % LDS_RING == 0, "the ring slot of K-step kk is kk % LDS_RING in EVERY tile: the ring must divide the chunk");
    const int g = lane >> 5;
    f32x16 bnext, prev;
    bias_init(lbias, 0, g, bnext);
    uint4 ring[LDS_RING];
    {
        const uint4 *c0 = lds + (BASE % NBUF) * (CHUNK * 64);
#pragma unroll
        for (int kk = 0; kk < LDS_RING; ++kk) ring[kk] = c0[kk * 64 + lane];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const uint4 *cur = lds + ((BASE + m) % NBUF) * (CHUNK * 64);
        uint4 *nxt = lds + ((BASE + m + 1) % NBUF) * (CHUNK * 64);
        f32x16 acc = bnext;
        if (m + 1 < MT) dma_chunk(w + (size_t)(m + 1) * CHUNK * 64, nxt, CHUNK, wave, lane);
        else dma_chunk(next, nxt, NEXT_FR, wave, lane);
        MLP_PIN();
        // The previous tile's epilogue as a three-stage software pipeline over the K-steps (one wave per SIMD: a dependent
        // exp -> add -> rcp -> fma chain inside one K-step stalls the wave past its MFMA's 32 cycles): element e has its exp at K-step 2e,
        // add + rcp at 2e + 1, fma (and, for odd e, the pack) at 2e + 2, so no K-step waits on a transcendental it issued itself.
        float te = 0.0f, tr = 0.0f, tdone_even = 0.0f;       // in flight: exp result, rcp result, finished even element of a pair
#pragma unroll
        for (int kk = 0; kk < CHUNK; ++kk) {
            if (kk == SYNC) dma_landed_barrier();
            if (kk == CHUNK / 2 && m + 1 < MT) bias_init(lbias, m + 1, g, bnext);
            const uint4 a = ring[kk % LDS_RING];
            if (kk + LDS_RING < CHUNK) ring[kk % LDS_RING] = cur[(kk + LDS_RING) * 64 + lane];
            else if (m + 1 < MT) ring[kk % LDS_RING] = nxt[(kk + LDS_RING - CHUNK) * 64 + lane];
            acc = mfma16<F16>(a, in[kk], acc);
            if (m > 0) {
                if (kk >= 2 && !(kk & 1)) {                  // stage C of element e = kk / 2 - 1
                    const int e = kk / 2 - 1;
                    const float t = __builtin_fmaf(-2.0f, tr, 1.0f);
                    if (e & 1) {
                        const unsigned u = pack2t<F16>(tdone_even, t);
                        const int q = e >> 1;
                        if (q < 4) out[2 * (m - 1)][q] = u; else out[2 * (m - 1) + 1][q - 4] = u;
                    } else tdone_even = t;
                }
                if (!(kk & 1)) te = __builtin_amdgcn_exp2f(prev[kk / 2]);       // stage A of element kk / 2
                else tr = __builtin_amdgcn_rcpf(te + 1.0f);                      // stage B of element (kk - 1) / 2
            }
            MLP_PIN();
        }
        if (m > 0) {                                         // stage C of element 15
            const unsigned u = pack2t<F16>(tdone_even, __builtin_fmaf(-2.0f, tr, 1.0f));
            out[2 * (m - 1) + 1][3] = u;
        }
        if (m + 1 < MT) prev = acc;
        else epilogue_t<F16>(acc, out[2 * m], out[2 * m + 1]);
    }
}

// grid = (workgroups of 128 drones, networks)
template <bool F16>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(1, 1))) void dn_mlp_lds_kernel(const MlpArgs a)
{
    // ONE __shared__ object (a second one makes hipcc drain the LDS-DMA before every first ds_read of a chunk):
    // 96 KB = three chunks (layer_lds_c), the staged biases, one uint4 for the masked-forward vote
    __shared__ __attribute__((aligned(16))) uint4 lds[NBUF * CHUNK * 64 + (NBIAS + 3) / 4 + 1];
    float *lbias = reinterpret_cast<float *>(lds + NBUF * CHUNK * 64);
    int *s_any = reinterpret_cast<int *>(lds + NBUF * CHUNK * 64 + (NBIAS + 3) / 4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 5, col = lane & 31;
    const MlpNetDev &net = a.net[blockIdx.y];
    const long long row0 = ((long long)blockIdx.x * WAVES + wave) * TILE;
    const bool live = row0 + col < a.n;
    const long long row = live ? row0 + col : a.n - 1;      // ragged tail: shadow the last drone, never store
    bool tile_wanted = true;
    if (a.row_mask) {
        // masked forward: the four waves vote; a workgroup none of whose 128 drones is flagged writes zeros and leaves;
        // inside a workgroup that stays, a wave whose own 32 drones are all unflagged still computes (it shares the
        // barriers and the DMA duty) but stores zeros
        const bool wanted = live && a.row_mask[row0 + col] != 0;
        tile_wanted = __ballot(wanted) != 0ull;
        if (lane == 0) s_any[wave] = tile_wanted;
        __syncthreads();
        if ((s_any[0] | s_any[1] | s_any[2] | s_any[3]) == 0) {
            if (g == 0 && live)
                for (int j = 0; j < net.out_dim; ++j) net.out[(row0 + col) * net.out_dim + j] = 0.0f;
            return;
        }
    }
    // biases -> LDS once (as in the pair / x3 / SAC kernels): the layers read them a tile ahead among the ring's LDS reads, in the counter the
    // compiler tracks; a global load per tile shares the vector-memory counter with the LDS-DMA pieces, which it does not see
    for (int i = threadIdx.x; i < NBIAS; i += 64 * WAVES)
        lbias[i] = i < H1 ? net.b1[i] : i < H1 + H2 ? net.b2[i - H1] : i < H1 + H2 + H3 ? net.b3[i - H1 - H2] : net.bh[i - H1 - H2 - H3];
    dma_chunk(net.w1, lds, H1 / 32, wave, lane);            // layer 1 = one chunk of 16 fragments, into buffer 0
    u32x4 x0[1];
    {
        const float *o = a.obs + row * a.obs_dim;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * g + 2 * q;
            x0[0][q] = pack2t<F16>(k < a.obs_dim ? o[k] : 0.0f, k + 1 < a.obs_dim ? o[k + 1] : 0.0f);
        }
    }
    chunk_barrier();
    // LDS buffers: layer 1 (one chunk of 16 fragments) in buffer 0; layer 2's 16 chunks start in buffer 1; layer 3's 8 chunks at
    // (1 + 16) % 3 = 2; the head (one chunk of 16 fragments) at (2 + 8) % 3 = 1
    constexpr int B2 = 1, B3 = (B2 + H2 / 32) % NBUF, BH = (B3 + H3 / 32) % NBUF;
    u32x4 h1[H1 / 16];
    dma_chunk(net.w2, lds + B2 * CHUNK * 64, CHUNK, wave, lane);             // layer 2, chunk 0 -> buffer 1
#pragma unroll
    for (int m = 0; m < H1 / 32; ++m) {                                      // K = 16: one fragment per M-tile
        f32x16 acc;
        bias_init(lbias, m, g, acc);
        const uint4 w = lds[m * 64 + lane];
        acc = mfma16<F16>(w, x0[0], acc);
        epilogue_t<F16>(acc, h1[2 * m], h1[2 * m + 1]);
    }
    chunk_barrier();
    u32x4 h2[H2 / 16];
    layer_lds_c<F16, H2 / 32, B2, CHUNK>(net.w2, lbias + H1, net.w3, h1, h2, lds, wave, lane);
    u32x4 h3[H3 / 16];
    layer_lds_c<F16, H3 / 32, B3, H3 / 16>(net.w3, lbias + H1 + H2, net.wh, h2, h3, lds, wave, lane);
    // head: one M-tile of H3/16 = 16 fragments; float32 result straight from the accumulator
    f32x16 acc;
    bias_init(lbias + H1 + H2 + H3, 0, g, acc);
    const uint4 *cur = lds + BH * (CHUNK * 64);
#pragma unroll
    for (int kk = 0; kk < H3 / 16; ++kk) {
        const uint4 w = cur[kk * 64 + lane];
        acc = mfma16<F16>(w, h3[kk], acc);
    }
    if (live) {
        float *o = net.out + (row0 + col) * net.out_dim;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = acc_row(0, g, r);
            if (j < net.out_dim) o[j] = tile_wanted ? acc[r] : 0.0f;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// Pair shape: 8 wavefronts per workgroup, TWO per SIMD, each pair sharing one tile of 32 drones by splitting K.
//
// Measured on MI355X (scratch micro-benchmark): a chain of v_mfma_f32_32x32x16_bf16 issues every 32.3 cycles, but with
// five VALU instructions per MFMA in the same wave it takes 50-63 -- inside ONE wave the VALU work is not hidden behind
// the matrix pipe.  The four-wave shape above issues ~9 non-MFMA instructions per MFMA (ds_read, accumulator reads,
// tanh, packing) and runs at ~95 cycles per MFMA.  Two waves on a SIMD do overlap (one's MFMAs run while the other
// issues VALU), so here waves w and w + 4 (same SIMD) work on the same 32 drones:
//   * each holds HALF of a layer's input activations (64 registers instead of 128: that is what makes two waves fit
//     in the register file) and multiplies its half of K for every M-tile (16 MFMAs instead of 32);
//   * per M-tile one of the two is the owner (tiles of the lower half of the layer's outputs belong to wave-half 0,
//     the rest to wave-half 1 = exactly the K-half each needs for the NEXT layer): the other one parks its partial
//     accumulator in LDS (4 KB), the owner adds it after the chunk barrier, applies bias + tanh and packs the result
//     into its own next-layer operands.  No activation ever needs to be re-distributed.
//   * weights stream through LDS exactly as in the four-wave shape (each fragment now read by four waves, one per
//     pair), biases are staged in LDS once, and the only global traffic inside the loop is the LDS-DMA.
// -----------------------------------------------------------------------------------------------------
constexpr int PWAVES = 8;
constexpr int DN_MLP_DEFAULT_SHAPE = 4;       // pi + vf at 32 768 drones, sustained (400 launches): 57.5 us (four waves, round 3: ring pinned, epilogue
                                              // software-pipelined) vs 59.0 us (pair) vs 79 us (one wave from L2); rounds 1-2: 77.9 / 68.9 / 79
constexpr int XB_U4 = 4 * 2 * 4 * 64;                         // exchange: 4 pairs x 2 parities x (16 f32 per lane = 4 uint4) x 64 lanes
constexpr int LDS_PAIR_U4 = 2 * CHUNK * 64 + XB_U4 + (NBIAS + 3) / 4 + 1;

template <int NF>
MLP_DEV void dma_pair(const uint4 *__restrict__ src, uint4 *lds, const int wave, const int lane)
{   // NF fragments shared by 8 waves: 4 (NF = 32) or 2 (NF = 16) consecutive fragments each
    constexpr int per = NF / PWAVES;
    const int f = wave * per;
    dma_pieces<per>(src + f * 64, (unsigned)(uintptr_t)(lds + f * 64), lane);
}

MLP_DEV void park_partial(float4 *xb, const int parity, const int lane, const f32x16 &acc)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) xb[(parity * 4 + j) * 64 + lane] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
}
// (owner: the accumulator of an owned tile starts from the bias -- bias_init above -- so that merging costs one add per value)
MLP_DEV void merge_partial(const float4 *xb, const int parity, const int lane, f32x16 &acc)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 p = xb[(parity * 4 + j) * 64 + lane];
        acc[4 * j] += p.x; acc[4 * j + 1] += p.y; acc[4 * j + 2] += p.z; acc[4 * j + 3] += p.w;
    }
}

// A 512-input layer for wave-half HALF: K-steps [16 HALF, 16 HALF + 16) of every M-tile, owner of tiles
// [HALF MT/2, (HALF + 1) MT/2).  inh = this half's 16 B operands; outh = the owned tiles' outputs = this half's B
// operands of the next layer.
template <bool F16, int HALF, int MT, int PAR, int NEXT_FR>
MLP_DEV void layer_pair(const uint4 *__restrict__ w, const float *lbias, const uint4 *__restrict__ next, const u32x4 (&inh)[16],
                        u32x4 (&outh)[MT], uint4 *wbuf, float4 *xb, const int wave, const int lane STP_PARAM)
{
    const int g = lane >> 5;
    f32x16 prev;
#ifdef DN_ABL_PARK_STALE
    f32x16 stale;
#pragma unroll
    for (int r = 0; r < 16; ++r) stale[r] = (float)lane;
#endif
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        uint4 *cur = wbuf + ((PAR + m) & 1) * (CHUNK * 64);
        uint4 *nxt = wbuf + ((PAR + m + 1) & 1) * (CHUNK * 64);
#if !defined(DN_MLP_ABLATE_DMA)
        if (m + 1 < MT) dma_pair<CHUNK>(w + (size_t)(m + 1) * CHUNK * 64, nxt, wave, lane);
        else dma_pair<NEXT_FR>(next, nxt, wave, lane);
#endif
        const bool fin = m > 0 && (((m - 1) >= MT / 2) == (HALF == 1));      // I own tile m-1: finish it under this tile's MFMAs
        if (fin) merge_partial(xb, (m - 1) & 1, lane, prev);
        f32x16 acc;
        if ((m >= MT / 2) == (HALF == 1)) bias_init(lbias, m, g, acc);       // mine: start from the bias
        else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        }
        uint4 ring[LDS_RING];
#pragma unroll
        for (int kk = 0; kk < LDS_RING; ++kk) ring[kk] = cur[(HALF * 16 + kk) * 64 + lane];
        MLP_PIN();
        float t_even = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const uint4 a = ring[kk % LDS_RING];
            if (kk + LDS_RING < 16) ring[kk % LDS_RING] = cur[(HALF * 16 + kk + LDS_RING) * 64 + lane];
            acc = mfma16<F16>(a, inh[kk], acc);
            if (fin) {                                       // one element of the owned previous tile per MFMA, packed in pairs
                const int ml = (m - 1) - HALF * (MT / 2), q = kk >> 1;
                if (!(kk & 1)) t_even = tanh_fast(prev[2 * q]);
                else {
                    const unsigned u = pack2t<F16>(t_even, tanh_fast(prev[2 * q + 1]));
                    if (q < 4) outh[2 * ml][q] = u; else outh[2 * ml + 1][q - 4] = u;
                }
            }
            MLP_PIN();
        }
        if ((m >= MT / 2) == (HALF == 1)) prev = acc;                         // mine: keep, finish next round
#ifdef DN_ABL_PARK_STALE
        else { park_partial(xb, m & 1, lane, stale); stale[0] += acc[0]; }   // timing ablation: the parked registers do not wait for the MFMAs
#else
        else park_partial(xb, m & 1, lane, acc);                             // partner's: hand over through LDS
#endif
        CHUNK_BARRIER();
    }
    if (HALF == 1) {                                                         // the last tile belongs to half 1
        merge_partial(xb, (MT - 1) & 1, lane, prev);
        epilogue_t<F16>(prev, outh[MT - 2], outh[MT - 1]);
    }
}

template <bool F16, int HALF>
MLP_DEV void mlp_pair_body(const MlpArgs &a, const MlpNetDev &net, uint4 *wbuf, float4 *xb, const float *lbias, const int wave,
                           const int lane, const long long row0, const bool live, const long long row, const bool tile_wanted,
                           f32x16 &head STP_PARAM)
{
    const int g = lane >> 5, col = lane & 31;
    u32x4 x0;
    {
        const float *o = a.obs + row * a.obs_dim;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * g + 2 * q;
            x0[q] = pack2t<F16>(k < a.obs_dim ? o[k] : 0.0f, k + 1 < a.obs_dim ? o[k + 1] : 0.0f);
        }
    }
    // layer 1 (K = 16: one K-step, no split): this half computes its own 8 tiles outright.  Its 16 fragments are in buffer 0.
    u32x4 h1[16];
    dma_pair<CHUNK>(net.w2, wbuf + CHUNK * 64, wave, lane);                  // layer 2, chunk 0 -> buffer 1
#pragma unroll
    for (int ml = 0; ml < 8; ++ml) {
        const int m = HALF * 8 + ml;
        f32x16 acc;
        bias_init(lbias, m, g, acc);
        const uint4 w = wbuf[m * 64 + lane];
        acc = mfma16<F16>(w, x0, acc);
        epilogue_t<F16>(acc, h1[2 * ml], h1[2 * ml + 1]);
    }
    CHUNK_BARRIER();
    u32x4 h2[16];
    layer_pair<F16, HALF, H2 / 32, 1, CHUNK>(net.w2, lbias + H1, net.w3, h1, h2, wbuf, xb, wave, lane STP_ARG);
    u32x4 h3[8];
    layer_pair<F16, HALF, H3 / 32, 1, H3 / 16>(net.w3, lbias + H1 + H2, net.wh, h2, h3, wbuf, xb, wave, lane STP_ARG);
    // head: one tile, K = 256 = 16 K-steps, 8 per half; fragments in buffer 1 (parity 1 + 16 + 8 -> 1)
    f32x16 acc;
    if (HALF == 0) bias_init(lbias + H1 + H2 + H3, 0, g, acc);
    else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    }
    const uint4 *cur = wbuf + CHUNK * 64;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const uint4 w = cur[(HALF * 8 + kk) * 64 + lane];
        acc = mfma16<F16>(w, h3[kk], acc);
    }
    if (HALF == 1) park_partial(xb, 0, lane, acc);
    CHUNK_BARRIER();
    if (HALF == 0) {
        merge_partial(xb, 0, lane, acc);
        head = acc;
        if (live) {
            float *o = net.out + (row0 + col) * net.out_dim;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = acc_row(0, g, r);
                if (j < net.out_dim) o[j] = tile_wanted ? acc[r] : 0.0f;
            }
        }
    }
}

template <bool F16, typename TAIL = NoTail>
__global__ __launch_bounds__(64 * PWAVES) void dn_mlp_pair_kernel(const MlpArgs a, const TAIL tail)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_PAIR_U4];         // ONE __shared__ object (see dn_mlp_lds_kernel)
    uint4 *wbuf = lds;
    float *lbias = reinterpret_cast<float *>(lds + 2 * CHUNK * 64 + XB_U4);
    int *s_any = reinterpret_cast<int *>(lds + LDS_PAIR_U4 - 1);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 3, half = wave >> 2;                              // waves w and w + 4 share a SIMD
    float4 *xb = reinterpret_cast<float4 *>(lds + 2 * CHUNK * 64) + pair * (2 * 4 * 64);
    const int col = lane & 31;
    const MlpNetDev &net = a.net[blockIdx.y];
    const long long row0 = ((long long)blockIdx.x * 4 + pair) * TILE;
    const bool live = row0 + col < a.n;
    const long long row = live ? row0 + col : a.n - 1;
    bool tile_wanted = true;
    if (a.row_mask) {
        const bool wanted = live && a.row_mask[row0 + col] != 0;
        tile_wanted = __ballot(wanted) != 0ull;
        if (lane == 0 && half == 0) s_any[pair] = tile_wanted;
        __syncthreads();
        if ((s_any[0] | s_any[1] | s_any[2] | s_any[3]) == 0) {
            if (half == 0 && (lane >> 5) == 0 && live)
                for (int j = 0; j < net.out_dim; ++j) net.out[(row0 + col) * net.out_dim + j] = 0.0f;
            return;
        }
    }
    // biases -> LDS (once), layer-1 fragments -> buffer 0
    for (int i = threadIdx.x; i < NBIAS; i += 64 * PWAVES)
        lbias[i] = i < H1 ? net.b1[i] : i < H1 + H2 ? net.b2[i - H1] : i < H1 + H2 + H3 ? net.b3[i - H1 - H2] : net.bh[i - H1 - H2 - H3];
    dma_pair<H1 / 32>(net.w1, wbuf, wave, lane);
#ifdef DN_MLP_STAMP
    Stamp stp{0, wave, blockIdx.x == 0 && blockIdx.y == 0 && lane == 0};
#endif
    CHUNK_BARRIER();
    f32x16 head;
    if (half == 0) mlp_pair_body<F16, 0>(a, net, wbuf, xb, lbias, wave, lane, row0, live, row, tile_wanted, head STP_ARG);
    else mlp_pair_body<F16, 1>(a, net, wbuf, xb, lbias, wave, lane, row0, live, row, tile_wanted, head STP_ARG);
    // the tail of network 0 (the actor): its workgroup holds the action means of drones [128 b, 128 b + 128) in the half-0 waves'
    // accumulators (lane group 0: rows 0..3 of drone `col`); every weight buffer is free by now
    if constexpr (TAIL::active) {
        if (blockIdx.y == 0) tail.template run<2>(lds, wave, lane, half == 0, pair * TILE + col, head, (long long)blockIdx.x);
    }
}

// -----------------------------------------------------------------------------------------------------
// fp32-grade shape ("x3"): the same network at the reference's precision.
//
// The reference's policy runs in float32 (SB3 ActorCriticPolicy, PBDroneSimulator.py:251-286), and PBDroneEnv.rescale_action
// leaves a band only 0.0073 wide in which an action is not saturated (PBDroneEnv.py:949-971): bf16 activations and weights
// (8 mantissa bits) move the action mean by ~1e-3, a seventh of that band.  gfx950 has no fast float32 matrix path (the
// float32 MFMA runs at 1/16 of the bf16 rate), so float32 grade is reached by splitting both operands into two bf16 words,
// w = w_hi + w_lo, x = x_hi + x_lo (16 mantissa bits each), and forming  w_hi x_hi + w_hi x_lo + w_lo x_hi  with three
// MFMAs into ONE float32 accumulator (the dropped w_lo x_lo term is 2^-16 of a product whose error budget is 2^-16).
// Measured against the torch float32 network: ~3e-5 on the action mean (tests: <= 1e-4).
//
// Registers decide the shape: activations now cost two words per value, 128 KB per 32-drone tile for a layer's input plus
// output -- a whole SIMD's register file.  So a tile is shared by TWO waves that split K exactly as in the pair shape, but
// on two SIMDs (one wave per SIMD: 512 registers each), four waves = two tiles = 64 drones per workgroup.  Weights stream
// through LDS as before, one chunk = the 32 hi + 32 lo fragments of an M-tile (64 KB, double-buffered: 128 KB of the 160).
// -----------------------------------------------------------------------------------------------------
constexpr int XWAVES = 4;
constexpr int CH3 = 2 * CHUNK;                               // fragments per chunk: hi then lo
constexpr int XB3_U4 = 2 * 2 * 4 * 64;                       // exchange: 2 pairs x 2 parities x 4 uint4 x 64 lanes
constexpr int LDS_X3_U4 = 2 * CH3 * 64 + XB3_U4 + (NBIAS + 3) / 4 + 1;

template <int NF>
MLP_DEV void dma_x3(const uint4 *__restrict__ src, uint4 *lds, const int wave, const int lane)
{   // NF fragments (64 or 32) shared by 4 waves: 16 or 8 consecutive fragments each, four per group
    constexpr int per = NF / XWAVES;
#pragma unroll
    for (int j0 = 0; j0 < per; j0 += 4) {
        const int f = wave * per + j0;
        dma_pieces<4>(src + f * 64, (unsigned)(uintptr_t)(lds + f * 64), lane);
    }
}
MLP_DEV float bf16_hi_as_float(const unsigned packed, const int which)      // element 0 / 1 of a packed bf16 pair, widened
{
    return __uint_as_float(which ? (packed & 0xFFFF0000u) : (packed << 16));
}
// two float32 values -> one dword of the hi operand and one of the lo operand (v = hi + lo to 16 mantissa bits)
MLP_DEV void split2(const float a, const float b, unsigned &hi, unsigned &lo)
{
    hi = pack2(a, b);
    lo = pack2(a - bf16_hi_as_float(hi, 0), b - bf16_hi_as_float(hi, 1));
}
MLP_DEV void epilogue3_pair(const f32x16 &acc, const int q, u32x4 &hlo, u32x4 &hhi, u32x4 &llo, u32x4 &lhi)
{   // accumulator elements 2q, 2q+1 -> tanh -> dword q of the hi and of the lo operand
    unsigned h, l;
    split2(tanh_fast(acc[2 * q]), tanh_fast(acc[2 * q + 1]), h, l);
    if (q < 4) { hlo[q] = h; llo[q] = l; } else { hhi[q - 4] = h; lhi[q - 4] = l; }
}
MLP_DEV void epilogue3(const f32x16 &acc, u32x4 &hlo, u32x4 &hhi, u32x4 &llo, u32x4 &lhi)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) epilogue3_pair(acc, q, hlo, hhi, llo, lhi);
}
MLP_DEV f32x16 mfma3(const uint4 whi, const uint4 wlo, const u32x4 xhi, const u32x4 xlo, f32x16 acc)
{   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wlo), __builtin_bit_cast(bf16x8, xhi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, whi), __builtin_bit_cast(bf16x8, xlo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, whi), __builtin_bit_cast(bf16x8, xhi), acc, 0, 0, 0);
    return acc;
}

// A 512-input layer for wave-half HALF in the x3 shape (see layer_pair): K-steps [16 HALF, 16 HALF + 16) of every M-tile,
// owner of tiles [HALF MT/2, (HALF + 1) MT/2).  A chunk holds fragments [0, 32) = hi, [32, 64) = lo of one M-tile.
template <int HALF, int MT, int PAR, int NEXT_FR>
MLP_DEV void layer_x3(const uint4 *__restrict__ w, const float *lbias, const uint4 *__restrict__ next, const u32x4 (&inh)[16],
                      const u32x4 (&inl)[16], u32x4 (&outh)[MT], u32x4 (&outl)[MT], uint4 *wbuf, float4 *xb, const int wave, const int lane STP_PARAM)
{
    const int g = lane >> 5;
    f32x16 prev;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        uint4 *cur = wbuf + ((PAR + m) & 1) * (CH3 * 64);
        uint4 *nxt = wbuf + ((PAR + m + 1) & 1) * (CH3 * 64);
#if DN_X3_DMA_SPREAD && !defined(DN_MLP_ABLATE_DMA)
        // the wave's pieces of the next chunk go out DN_X3_DMA_SPREAD per K-step, between the MFMAs: with one wave per SIMD a burst of
        // 16 pieces at the top of the tile is ~1 000 cycles in which the matrix pipe idles (every wave of the CU queues on the one
        // 64 B/clk vector-memory path right after the barrier); one piece behind an MFMA hides in that MFMA's 32 cycles
        const int NPER = (m + 1 < MT ? CH3 : NEXT_FR) / XWAVES;
        const uint4 *nsrc = (m + 1 < MT ? w + (size_t)(m + 1) * CH3 * 64 : next) + (size_t)wave * NPER * 64;
        uint4 *ndst = nxt + (size_t)wave * NPER * 64;
#elif !defined(DN_MLP_ABLATE_DMA)
        if (m + 1 < MT) dma_x3<CH3>(w + (size_t)(m + 1) * CH3 * 64, nxt, wave, lane);
        else dma_x3<NEXT_FR>(next, nxt, wave, lane);
#endif
        const bool fin = m > 0 && (((m - 1) >= MT / 2) == (HALF == 1));      // I own tile m-1: finish it under this tile's MFMAs
#if DN_X3_LAZY_MERGE
        // Nothing but LDS reads may stand between the chunk barrier and the tile's first MFMA: a merge here (four reads, their round trip, 16
        // accumulator moves and adds) and an accumulator initialised from the bias (16 moves) held the matrix pipe idle for ~400 cycles of every
        // ~2 400-cycle tile (ablation without the exchange: 154 -> 127 us).  So the partial sums stay in the registers their reads land in and
        // are added where the epilogue consumes them, the OWNER's accumulator starts from zero (the first MFMA takes C = 0: no moves), and the
        // bias rides in the partner's accumulator, whose initialisation sits in that wave's slack before the barrier.
        float4 part[4];
        if (fin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] = xb[(((m - 1) & 1) * 4 + j) * 64 + lane];
        }
        f32x16 acc;
        if ((m >= MT / 2) == (HALF == 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        } else bias_init(lbias, m, g, acc);                                  // the partner's partial sum carries the bias
#else
#ifndef DN_ABL_X3_NOXCHG
        if (fin) merge_partial(xb, (m - 1) & 1, lane, prev);
#endif
        f32x16 acc;
        if ((m >= MT / 2) == (HALF == 1)) bias_init(lbias, m, g, acc);       // mine: start from the bias
        else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        }
#endif
        constexpr int RG = 4;                                                // ring depth (fragment PAIRS in flight)
        uint4 rh[RG], rl[RG];
#pragma unroll
        for (int kk = 0; kk < RG; ++kk) {
            rh[kk] = cur[(HALF * 16 + kk) * 64 + lane];
            rl[kk] = cur[(CHUNK + HALF * 16 + kk) * 64 + lane];
        }
        MLP_PIN();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const uint4 ah = rh[kk % RG], al = rl[kk % RG];
#ifndef DN_MLP_ABLATE_READS
            if (kk + RG < 16) {
                rh[kk % RG] = cur[(HALF * 16 + kk + RG) * 64 + lane];
                rl[kk % RG] = cur[(CHUNK + HALF * 16 + kk + RG) * 64 + lane];
            }
#endif
            acc = mfma3(ah, al, inh[kk], inl[kk], acc);
#if DN_X3_DMA_SPREAD && !defined(DN_MLP_ABLATE_DMA)
#pragma unroll
            for (int j = 0; j < DN_X3_DMA_SPREAD; ++j) {
                const int pc = kk * DN_X3_DMA_SPREAD + j;
                if (pc < NPER) dma_piece(nsrc + pc * 64, ndst + pc * 64, lane);
            }
#endif
#ifndef DN_ABL_X3_NOEPI
            if (fin && (kk & 1)) {
                const int ml = (m - 1) - HALF * (MT / 2);
#if DN_X3_LAZY_MERGE
                const int q = kk >> 1;
                const float4 pp = part[q >> 1];
                prev[2 * q] += (q & 1) ? pp.z : pp.x;
                prev[2 * q + 1] += (q & 1) ? pp.w : pp.y;
#endif
                epilogue3_pair(prev, kk >> 1, outh[2 * ml], outh[2 * ml + 1], outl[2 * ml], outl[2 * ml + 1]);
            }
#else
            if (fin && kk == 15) {                                            // timing ablation: no tanh, no split -- one dependence on prev
                const int ml = (m - 1) - HALF * (MT / 2);
                outh[2 * ml][0] = __float_as_uint(prev[0]); outh[2 * ml + 1][0] = __float_as_uint(prev[8]);
                outl[2 * ml][0] = __float_as_uint(prev[1]); outl[2 * ml + 1][0] = __float_as_uint(prev[9]);
            }
#endif
            MLP_PIN();
        }
        if ((m >= MT / 2) == (HALF == 1)) prev = acc;                         // mine: keep, finish next round
#ifndef DN_ABL_X3_NOXCHG
        else park_partial(xb, m & 1, lane, acc);                             // partner's: hand over through LDS
#else
        else prev[0] += acc[0];
#endif
#ifdef DN_ABL_X3_NOBAR
        __builtin_amdgcn_s_waitcnt(0x0070);
#else
        CHUNK_BARRIER();
#endif
    }
    if (HALF == 1) {                                                         // the last tile belongs to half 1
        merge_partial(xb, (MT - 1) & 1, lane, prev);
        epilogue3(prev, outh[MT - 2], outh[MT - 1], outl[MT - 2], outl[MT - 1]);
    }
}

template <int HALF>
MLP_DEV void mlp_x3_body(const MlpArgs &a, const MlpNetDev &net, uint4 *wbuf, float4 *xb, const float *lbias, const int wave,
                         const int lane, const long long row0, const bool live, const long long row, const bool tile_wanted,
                         f32x16 &head STP_PARAM)
{
    const int g = lane >> 5, col = lane & 31;
    u32x4 x0h, x0l;
    {
        const float *o = a.obs + row * a.obs_dim;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * g + 2 * q;
            unsigned h, l;
            split2(k < a.obs_dim ? o[k] : 0.0f, k + 1 < a.obs_dim ? o[k + 1] : 0.0f, h, l);
            x0h[q] = h; x0l[q] = l;
        }
    }
    // layer 1 (K = 16: one K-step, no split): this half computes its own 8 tiles outright.  Its 16 (hi, lo) fragment pairs are in
    // buffer 0, stored per M-tile as hi then lo.
    u32x4 h1h[16], h1l[16];
    dma_x3<CH3>(net.w2, wbuf + CH3 * 64, wave, lane);                        // layer 2, chunk 0 -> buffer 1
#pragma unroll
    for (int ml = 0; ml < 8; ++ml) {
        const int m = HALF * 8 + ml;
        f32x16 acc;
        bias_init(lbias, m, g, acc);
        const uint4 wh = wbuf[(2 * m) * 64 + lane], wl = wbuf[(2 * m + 1) * 64 + lane];
        acc = mfma3(wh, wl, x0h, x0l, acc);
        epilogue3(acc, h1h[2 * ml], h1h[2 * ml + 1], h1l[2 * ml], h1l[2 * ml + 1]);
    }
    CHUNK_BARRIER();
    u32x4 h2h[16], h2l[16];
    layer_x3<HALF, H2 / 32, 1, CH3>(net.w2, lbias + H1, net.w3, h1h, h1l, h2h, h2l, wbuf, xb, wave, lane STP_ARG);
    u32x4 h3h[8], h3l[8];
    layer_x3<HALF, H3 / 32, 1, 2 * (H3 / 16)>(net.w3, lbias + H1 + H2, net.wh, h2h, h2l, h3h, h3l, wbuf, xb, wave, lane STP_ARG);
    // head: one tile, K = 256 = 16 K-steps, 8 per half; chunk in buffer 1 as [16 hi][16 lo]
    f32x16 acc;
    if (HALF == 0) bias_init(lbias + H1 + H2 + H3, 0, g, acc);
    else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    }
    const uint4 *cur = wbuf + CH3 * 64;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const uint4 wh = cur[(HALF * 8 + kk) * 64 + lane], wl = cur[(H3 / 16 + HALF * 8 + kk) * 64 + lane];
        acc = mfma3(wh, wl, h3h[kk], h3l[kk], acc);
    }
    if (HALF == 1) park_partial(xb, 0, lane, acc);
    CHUNK_BARRIER();
    if (HALF == 0) {
        merge_partial(xb, 0, lane, acc);
        head = acc;
        if (live) {
            float *o = net.out + (row0 + col) * net.out_dim;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = acc_row(0, g, r);
                if (j < net.out_dim) o[j] = tile_wanted ? acc[r] : 0.0f;
            }
        }
    }
}

template <typename TAIL = NoTail>
__global__ __launch_bounds__(64 * XWAVES) __attribute__((amdgpu_waves_per_eu(1, 1))) void dn_mlp_x3_kernel(const MlpArgs a, const TAIL tail)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_X3_U4];           // ONE __shared__ object (see dn_mlp_lds_kernel)
    uint4 *wbuf = lds;
    float *lbias = reinterpret_cast<float *>(lds + 2 * CH3 * 64 + XB3_U4);
    int *s_any = reinterpret_cast<int *>(lds + LDS_X3_U4 - 1);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 1, half = wave >> 1;                              // waves p and p + 2 share a tile, on two SIMDs
    float4 *xb = reinterpret_cast<float4 *>(lds + 2 * CH3 * 64) + pair * (2 * 4 * 64);
    const int col = lane & 31;
    const MlpNetDev &net = a.net[blockIdx.y];
    const long long row0 = ((long long)blockIdx.x * 2 + pair) * TILE;
    const bool live = row0 + col < a.n;
    const long long row = live ? row0 + col : a.n - 1;
    bool tile_wanted = true;
    if (a.row_mask) {
        const bool wanted = live && a.row_mask[row0 + col] != 0;
        tile_wanted = __ballot(wanted) != 0ull;
        if (lane == 0 && half == 0) s_any[pair] = tile_wanted;
        __syncthreads();
        if ((s_any[0] | s_any[1]) == 0) {
            if (half == 0 && (lane >> 5) == 0 && live)
                for (int j = 0; j < net.out_dim; ++j) net.out[(row0 + col) * net.out_dim + j] = 0.0f;
            return;
        }
    }
    // biases -> LDS (once), layer-1 fragments (16 hi/lo pairs) -> buffer 0
    for (int i = threadIdx.x; i < NBIAS; i += 64 * XWAVES)
        lbias[i] = i < H1 ? net.b1[i] : i < H1 + H2 ? net.b2[i - H1] : i < H1 + H2 + H3 ? net.b3[i - H1 - H2] : net.bh[i - H1 - H2 - H3];
    dma_x3<2 * (H1 / 32)>(net.w1, wbuf, wave, lane);
#ifdef DN_MLP_STAMP
    Stamp stp{0, wave, blockIdx.x == 0 && blockIdx.y == 0 && lane == 0};
#endif
    CHUNK_BARRIER();
    f32x16 head;
    if (half == 0) mlp_x3_body<0>(a, net, wbuf, xb, lbias, wave, lane, row0, live, row, tile_wanted, head STP_ARG);
    else mlp_x3_body<1>(a, net, wbuf, xb, lbias, wave, lane, row0, live, row, tile_wanted, head STP_ARG);
    if constexpr (TAIL::active) {                            // see dn_mlp_pair_kernel: here the workgroup holds 64 drones = one step tile
        if (blockIdx.y == 0) tail.template run<1>(lds, wave, lane, half == 0, pair * TILE + col, head, (long long)blockIdx.x);
    }
}


// -----------------------------------------------------------------------------------------------------
// The SAC actor (dn_mlp_net.arch = DN_MLP_ARCH_SAC): obs -> 256 -> 256 -> {mu[4] | log_std[4]}, ReLU.
//
// SB3's SAC Actor with policy_kwargs net_arch = dict(pi=[256, 256]), activation_fn = ReLU (PBDroneSimulator.py:297-303):
// latent_pi = ReLU(W2 ReLU(W1 obs + b1) + b2), mu = Wm latent + bm, log_std = Ws latent + bs; the two heads are stacked
// into one [8, 256] matrix by the host, so the whole actor is three layers of the same transposed product as above and a
// 32-drone tile needs 71 k multiply-adds per drone against the PPO pair's 2 x 400 k.  First shape (DN_MLP_SAC_SHAPE=1): one
// wavefront per tile, the 142 KB of weights (284 KB in the fp32 grade) straight from L2 through an 8-deep register ring,
// activations in registers (64 per layer, 128 with the hi / lo split).  Both grades in one body: X3 = the split-bf16
// float32 grade of dn_mlp_x3_kernel (three MFMAs per fragment pair, w_lo x_hi + w_hi x_lo + w_hi x_hi).
// -----------------------------------------------------------------------------------------------------
constexpr int S1 = 256, S2 = 256;
constexpr int SAC_RING = 8;

MLP_DEV void split_pair(const float a, const float b, unsigned &hi, unsigned &lo)
{   // two float32 values -> one dword of the hi operand and one of the lo operand (v = hi + lo to 16 mantissa bits)
    bf16x2 h, l;
    h[0] = (__bf16)a; h[1] = (__bf16)b;
    l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// One Linear(+ReLU) layer of the SAC actor for this wave's 32 drones.  Weight stream per M-tile: KS fragments (bf16
// grade) or KS hi fragments then KS lo fragments (X3), consumed in storage order through a ring.
template <int KS, int MT, bool X3>
MLP_DEV void sac_layer(const uint4 *__restrict__ w, const float *__restrict__ bias, const u32x4 (&inh)[KS], const u32x4 (&inl)[KS],
                       u32x4 (&outh)[2 * MT], u32x4 (&outl)[2 * MT], const int lane)
{
    const int g = lane >> 5;
    constexpr int PER = X3 ? 2 : 1;
    constexpr int T = KS * MT;                              // K-steps in the layer; step t = (m, kk) reads PER fragments
    constexpr int P = SAC_RING < T ? SAC_RING : T;
    const uint4 *wl = w + lane;
    uint4 rh[P], rl[P];
    auto frag = [&](const int t, const int part) { return wl[(size_t)(((t / KS) * PER + part) * KS + t % KS) * 64]; };
#pragma unroll
    for (int t = 0; t < P; ++t) {
        rh[t] = frag(t, 0);
        if (X3) rl[t] = frag(t, 1);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[acc_row(m, g, r)];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int t = m * KS + kk;
            const uint4 ah = rh[t % P], al = rl[t % P];
            if (t + P < T) {
                rh[t % P] = frag(t + P, 0);
                if (X3) rl[t % P] = frag(t + P, 1);
            }
            if (X3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, inh[kk]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, inl[kk]), acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, inh[kk]), acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {                       // ReLU, pack (and split) into the next layer's K-steps 2m, 2m + 1
            const float v0 = fmaxf(acc[2 * q], 0.0f), v1 = fmaxf(acc[2 * q + 1], 0.0f);
            unsigned hi, lo = 0u;
            if (X3) split_pair(v0, v1, hi, lo); else hi = pack2(v0, v1);
            if (q < 4) { outh[2 * m][q] = hi; if (X3) outl[2 * m][q] = lo; }
            else { outh[2 * m + 1][q - 4] = hi; if (X3) outl[2 * m + 1][q - 4] = lo; }
        }
    }
}

template <bool X3>
__global__ __launch_bounds__(64) void dn_mlp_sac_kernel(const MlpArgs a)
{
    const int lane = threadIdx.x;
    const int g = lane >> 5, col = lane & 31;
    const MlpNetDev &net = a.net[blockIdx.y];
    const long long row0 = (long long)blockIdx.x * TILE;
    const bool live = row0 + col < a.n;
    const long long row = live ? row0 + col : a.n - 1;      // ragged last tile: shadow the last drone
    if (a.row_mask) {
        const bool wanted = live && a.row_mask[row0 + col] != 0;
        if (__ballot(wanted) == 0ull) {
            if (g == 0 && live)
                for (int j = 0; j < net.out_dim; ++j) net.out[(row0 + col) * net.out_dim + j] = 0.0f;
            return;
        }
    }
    float ob[8];
    load_obs8(a, row, g, ob);
    u32x4 x0h[1], x0l[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hi, lo = 0u;
        if (X3) split_pair(ob[2 * q], ob[2 * q + 1], hi, lo); else hi = pack2(ob[2 * q], ob[2 * q + 1]);
        x0h[0][q] = hi; x0l[0][q] = lo;
    }
    u32x4 h1h[S1 / 16], h1l[S1 / 16];
    sac_layer<1, S1 / 32, X3>(net.w1, net.b1, x0h, x0l, h1h, h1l, lane);
    u32x4 h2h[S2 / 16], h2l[S2 / 16];
    sac_layer<S1 / 16, S2 / 32, X3>(net.w2, net.b2, h1h, h1l, h2h, h2l, lane);
    // heads: one M-tile (8 rows used), float32 straight from the accumulator
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = net.bh[acc_row(0, g, r)];
#pragma unroll
    for (int kk = 0; kk < S2 / 16; ++kk) {
        const uint4 wh = net.wh[kk * 64 + lane];
        if (X3) {
            const uint4 wlo = net.wh[(S2 / 16 + kk) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wlo), __builtin_bit_cast(bf16x8, h2h[kk]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, h2l[kk]), acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, h2h[kk]), acc, 0, 0, 0);
    }
    if (live) {
        float *o = net.out + (row0 + col) * net.out_dim;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = acc_row(0, g, r);
            if (j < net.out_dim) o[j] = acc[r];
        }
    }
}


// The same actor with the weight stream shared by four wavefronts through LDS (default).  Straight from L2 every wave pulls the
// whole 142 KB (284 KB) itself: 1 024 waves x 284 KB in 22 us is 13 TB/s out of L2 (rocprof: 10.5 / 22.0 us for 32 768 drones
// in the bf16 / float32 grade; with the stream shared: 10.8 / 19.5 us -- at this size launch, first round trip and the chunk
// barriers weigh as much as the 2.5 / 7.7 us of MFMA time).  Here a workgroup of four waves
// (128 drones) brings every fragment in once by LDS-DMA, chunks of 32 fragments double-buffered exactly as dn_mlp_lds_kernel
// does: chunk 0 = layer 1 (8 fragments; 16 with the hi / lo split), then layer 2 in M-tile order (float32 grade: one M-tile =
// 16 hi + 16 lo fragments = one chunk; bf16 grade: two M-tiles per chunk), then the stacked heads (16 / 32 fragments).
constexpr int SAC_WAVES = 4;
template <int NF>
MLP_DEV void sac_dma(const uint4 *__restrict__ src, uint4 *lds, const int wave, const int lane)
{   // NF fragments (8, 16 or 32) shared by 4 waves: NF / 4 consecutive fragments each
    constexpr int per = NF / SAC_WAVES;
    const int f = wave * per;
    const uint4 *g = src + f * 64;
    const unsigned lds_dst = (unsigned)(uintptr_t)(lds + f * 64);
    if (per == 8) {
        dma_pieces<4>(g, lds_dst, lane);
        dma_pieces<4>(g + 4 * 64, lds_dst + 4096u, lane);
    } else if (per == 4) dma_pieces<4>(g, lds_dst, lane);
    else dma_pieces<2>(g, lds_dst, lane);
}
// ReLU, pack (and split) the 16 accumulator values of an M-tile into the next layer's K-steps 2m, 2m + 1
template <bool X3, bool F16>
MLP_DEV void sac_epilogue(const f32x16 &acc, u32x4 &h0, u32x4 &h1, u32x4 &l0, u32x4 &l1)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float v0 = fmaxf(acc[2 * q], 0.0f), v1 = fmaxf(acc[2 * q + 1], 0.0f);
        unsigned hi, lo = 0u;
        if (X3) split_pair(v0, v1, hi, lo); else hi = pack2t<F16>(v0, v1);
        if (q < 4) { h0[q] = hi; if (X3) l0[q] = lo; }
        else { h1[q - 4] = hi; if (X3) l1[q - 4] = lo; }
    }
}
// one M-tile over KS K-steps from an LDS chunk: hi fragments at chunk[off + kk], lo fragments at chunk[off + KS + kk]
template <int KS, bool X3, bool F16>
MLP_DEV f32x16 sac_tile(const uint4 *chunk, const int off, const float *lbias, const int m, const int g, const int lane,
                        const u32x4 (&inh)[KS], const u32x4 (&inl)[KS])
{
    // The biases come from LDS (staged once, as in the pair kernel).  Read from global memory per tile -- as this kernel did until round
    // 3 -- the load's s_waitcnt vmcnt(0) sits right behind the LDS-DMA of the NEXT chunk, issued a moment earlier, and every chunk of
    // this latency-bound kernel (0.3-0.8 us of MFMA work per chunk) waited out a full DMA round trip (~1 us) before its first MFMA.
    f32x16 acc;
    bias_init(lbias, m, g, acc);
    constexpr int RG = KS < 4 ? KS : 4;
    uint4 rh[RG], rl[RG];
#pragma unroll
    for (int kk = 0; kk < RG; ++kk) {
        rh[kk] = chunk[(off + kk) * 64 + lane];
        if (X3) rl[kk] = chunk[(off + KS + kk) * 64 + lane];
    }
    MLP_PIN();
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const uint4 ah = rh[kk % RG], al = rl[kk % RG];
        if (kk + RG < KS) {
            rh[kk % RG] = chunk[(off + kk + RG) * 64 + lane];
            if (X3) rl[kk % RG] = chunk[(off + KS + kk + RG) * 64 + lane];
        }
        if (X3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, inh[kk]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, inl[kk]), acc, 0, 0, 0);
        }
        acc = mfma16<F16>(ah, inh[kk], acc);
        MLP_PIN();
    }
    return acc;
}

// Layer 2 and the stacked heads of the SAC actor as ONE stream of 9 M-tiles x 16 K-steps (the heads' K is layer 2's: 256): the register ring
// runs on across tiles and chunks, the chunk barrier sits at the K-step whose refill is the first to read the next chunk, the next tile's bias
// lands in its accumulator registers half a tile ahead, and a tile's ReLU / pack (/ split) is spread under the next tile's first eight
// K-steps (layer-2 tile 7's under the heads', which need its output at K-steps 14 and 15).  Until round 6 every tile was bias reads -> ring
// fill -> MFMAs -> epilogue -> barrier, one after the other: ~1 000 cycles a tile beside 512 (bf16) or 1 536 (float32 grade) of MFMAs, in a
// kernel that is one dependent chain per SIMD at 32 768 drones.  Chunk c lives in buffer (1 + c) % NB (buffer 0 = layer 1); with three
// buffers the DMA of chunk c + 1, requested at the top of chunk c, lands where chunk c - 2 was (see layer_lds_c); with two, where chunk
// c - 1 was, behind a barrier of its own.
template <bool X3, bool F16, int NB>
MLP_DEV f32x16 sac_stream(const MlpNetDev &net, const float *lbias, uint4 *lds, const u32x4 (&h1h)[S1 / 16], const u32x4 (&h1l)[S1 / 16],
                          const int wave, const int lane)
{
    constexpr int KS = S1 / 16, L2T = S2 / 32, NT = L2T + 1;               // 16 K-steps a tile; 8 layer-2 tiles + the heads
    constexpr int PER = X3 ? 2 : 1, TPC = X3 ? 1 : 2, NL2 = L2T / TPC;     // layer-2 tiles per chunk, layer-2 chunks; the heads are chunk NL2
    constexpr int RG = 4;                                                  // ring depth in K-steps
    static_assert(S2 / 16 == KS, "the heads are streamed as a ninth tile of layer 2's K");
    static_assert(KS % RG == 0, "the ring slot of K-step kk is kk % RG in every tile");
    const int g = lane >> 5;
    auto chunk_of = [](const int t) { return t < L2T ? t / TPC : NL2; };
    auto off_of = [](const int t) { return t < L2T ? (t % TPC) * KS * PER : 0; };   // hi fragment kk at off + kk, lo at off + KS + kk
    auto buf = [&](const int c) { return lds + ((1 + c) % NB) * (CHUNK * 64); };
    u32x4 h2h[S2 / 16], h2l[S2 / 16];
    uint4 rh[RG], rl[RG];
#pragma unroll
    for (int kk = 0; kk < RG; ++kk) {
        rh[kk] = buf(0)[kk * 64 + lane];
        if (X3) rl[kk] = buf(0)[(KS + kk) * 64 + lane];
    }
    f32x16 bnext, prev, acc;
    bias_init(lbias + S1, 0, g, bnext);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = chunk_of(t);
        acc = bnext;
        if (off_of(t) == 0 && c + 1 <= NL2) {                              // first tile of chunk c: request chunk c + 1
            // two buffers: the target held chunk c - 1, whose last fragments every wave has CONSUMED only once it stands here (the barrier
            // inside chunk c - 1 came four K-steps before its end)
            if (NB == 2 && c > 0) __builtin_amdgcn_s_barrier();
            if (c + 1 < NL2) sac_dma<CHUNK>(net.w2 + (size_t)(c + 1) * CHUNK * 64, buf(c + 1), wave, lane);
            else sac_dma<(S2 / 16) * PER>(net.wh, buf(c + 1), wave, lane);
        }
        MLP_PIN();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const uint4 ah = rh[kk % RG], al = rl[kk % RG];
            const int tr = t + (kk + RG) / KS, kr = (kk + RG) % KS;        // the K-step the ring is refilled for
            if (tr < NT) {
                if (chunk_of(tr) != c && kr == 0) dma_landed_barrier();    // the first read of the next chunk: its pieces have landed
                const uint4 *src = buf(chunk_of(tr)) + (off_of(tr) + kr) * 64 + lane;
                rh[kk % RG] = src[0];
                if (X3) rl[kk % RG] = src[KS * 64];
            }
            if (kk == KS / 2 && t + 1 < NT) bias_init(t + 1 < L2T ? lbias + S1 : lbias + S1 + S2, t + 1 < L2T ? t + 1 : 0, g, bnext);
            const u32x4 &bh = t < L2T ? h1h[kk] : h2h[kk], &bl = t < L2T ? h1l[kk] : h2l[kk];
            if (X3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
            }
            acc = mfma16<F16>(ah, bh, acc);
            if (t > 0 && kk < 8) {                                         // the previous tile's ReLU / pack / split, one pair a K-step
                const int q = kk, m = t - 1;
                const float v0 = fmaxf(prev[2 * q], 0.0f), v1 = fmaxf(prev[2 * q + 1], 0.0f);
                unsigned hi, lo = 0u;
                if (X3) split_pair(v0, v1, hi, lo); else hi = pack2t<F16>(v0, v1);
                if (q < 4) { h2h[2 * m][q] = hi; if (X3) h2l[2 * m][q] = lo; }
                else { h2h[2 * m + 1][q - 4] = hi; if (X3) h2l[2 * m + 1][q - 4] = lo; }
            }
            MLP_PIN();
        }
        if (t + 1 < NT) prev = acc;
    }
    return acc;
}

template <bool X3, bool F16>
__global__ __launch_bounds__(64 * SAC_WAVES) void dn_mlp_sac_lds_kernel(const MlpArgs a)
{
    constexpr int NB_SAC = S1 + S2 + 32;                   // float32 biases of the three layers, staged in LDS
    // chunk buffers of sac_stream: three in the float32 grade (one wave per SIMD by its registers anyway), two in the 16-bit grades, where a
    // second workgroup per CU is what large fleets run on (131 072 drones: 25 us with two buffers, 31 with three)
    constexpr int NB = X3 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) uint4 lds[NB * CHUNK * 64 + NB_SAC / 4 + 1];    // ONE __shared__ object (see dn_mlp_lds_kernel)
    float *lbias = reinterpret_cast<float *>(lds + NB * CHUNK * 64);
    int *s_any = reinterpret_cast<int *>(lds + NB * CHUNK * 64 + NB_SAC / 4);
    constexpr int PER = X3 ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 5, col = lane & 31;
    const MlpNetDev &net = a.net[blockIdx.y];
    const long long row0 = ((long long)blockIdx.x * SAC_WAVES + wave) * TILE;
    const bool live = row0 + col < a.n;
    const long long row = live ? row0 + col : a.n - 1;
    bool tile_wanted = true;
    if (a.row_mask) {
        const bool wanted = live && a.row_mask[row0 + col] != 0;
        tile_wanted = __ballot(wanted) != 0ull;
        if (lane == 0) s_any[wave] = tile_wanted;
        __syncthreads();
        if ((s_any[0] | s_any[1] | s_any[2] | s_any[3]) == 0) {
            if (g == 0 && live)
                for (int j = 0; j < net.out_dim; ++j) net.out[(row0 + col) * net.out_dim + j] = 0.0f;
            return;
        }
    }
    for (int i = threadIdx.x; i < NB_SAC; i += 64 * SAC_WAVES)
        lbias[i] = i < S1 ? net.b1[i] : i < S1 + S2 ? net.b2[i - S1] : net.bh[i - S1 - S2];
    sac_dma<(S1 / 32) * PER>(net.w1, lds, wave, lane);      // chunk 0 = layer 1 -> buffer 0
    float ob[8];
    load_obs8(a, row, g, ob);
    u32x4 x0h[1], x0l[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hi, lo = 0u;
        if (X3) split_pair(ob[2 * q], ob[2 * q + 1], hi, lo); else hi = pack2t<F16>(ob[2 * q], ob[2 * q + 1]);
        x0h[0][q] = hi; x0l[0][q] = lo;
    }
    chunk_barrier();
    u32x4 h1h[S1 / 16], h1l[S1 / 16];
    sac_dma<CHUNK>(net.w2, lds + CHUNK * 64, wave, lane);   // layer 2, chunk 0 -> buffer 1
#pragma unroll
    for (int m = 0; m < S1 / 32; ++m) {
        const f32x16 acc = sac_tile<1, X3, F16>(lds, m * PER, lbias, m, g, lane, x0h, x0l);
        sac_epilogue<X3, F16>(acc, h1h[2 * m], h1h[2 * m + 1], h1l[2 * m], h1l[2 * m + 1]);
    }
    chunk_barrier();
    const f32x16 acc = sac_stream<X3, F16, NB>(net, lbias, lds, h1h, h1l, wave, lane);     // layer 2 and the heads (float32 from the accumulator)
    if (live) {
        float *o = net.out + (row0 + col) * net.out_dim;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = acc_row(0, g, r);
            if (j < net.out_dim) o[j] = tile_wanted ? acc[r] : 0.0f;
        }
    }
}

}  // namespace

#ifndef DN_MLP_NO_LAUNCHER
#ifdef DN_MLP_STAMP
#include <cstdio>
static void dump_stamps(const char *kernel)
{
    const char *path = getenv("DN_MLP_STAMP_FILE");
    if (!path) return;
    long long h[8][256];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof h) != hipSuccess) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    for (int w = 0; w < 8; ++w) {
        fprintf(f, "%s wave %d:", kernel, w);
        for (int i = 0; i < 256 && h[w][i]; ++i) fprintf(f, " %lld", h[w][i] - h[w][0]);
        fprintf(f, "\n");
    }
    fclose(f);
    static long long zero[8][256];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), zero, sizeof zero);
}
#else
static void dump_stamps(const char *) {}
#endif

hipError_t dn_launch_mlp(const dn_mlp_net *nets, int num_nets, const float *obs, const uint8_t *row_mask, long long n, int obs_dim,
                         hipStream_t stream)
{
    MlpArgs a;
    for (int k = 0; k < 2; ++k) {
        const dn_mlp_net &s = nets[k < num_nets ? k : 0];
        a.net[k].w1 = (const uint4 *)s.w1; a.net[k].w2 = (const uint4 *)s.w2; a.net[k].w3 = (const uint4 *)s.w3;
        a.net[k].wh = (const uint4 *)s.wh;
        a.net[k].b1 = s.b1; a.net[k].b2 = s.b2; a.net[k].b3 = s.b3; a.net[k].bh = s.bh;
        a.net[k].out = s.out; a.net[k].out_dim = s.out_dim;
    }
    a.obs = obs; a.row_mask = row_mask; a.n = n; a.obs_dim = obs_dim;
    const unsigned tiles = (unsigned)((n + TILE - 1) / TILE);
    if (nets[0].arch == DN_MLP_ARCH_SAC) {
        const char *es = getenv("DN_MLP_SAC_SHAPE");        // 1 = one wave per workgroup straight from L2 | 4 = four waves sharing the stream through LDS
        const bool direct = es && atoi(es) == 1;
        const dim3 grid4((tiles + SAC_WAVES - 1) / SAC_WAVES, num_nets);
        if (nets[0].grade == 2) {                            // float16 operands: the shared-stream shape only
            hipLaunchKernelGGL((dn_mlp_sac_lds_kernel<false, true>), grid4, dim3(64 * SAC_WAVES), 0, stream, a);
        } else if (nets[0].grade == 1) {
            if (direct) hipLaunchKernelGGL(dn_mlp_sac_kernel<true>, dim3(tiles, num_nets), dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((dn_mlp_sac_lds_kernel<true, false>), grid4, dim3(64 * SAC_WAVES), 0, stream, a);
        } else {
            if (direct) hipLaunchKernelGGL(dn_mlp_sac_kernel<false>, dim3(tiles, num_nets), dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((dn_mlp_sac_lds_kernel<false, false>), grid4, dim3(64 * SAC_WAVES), 0, stream, a);
        }
        return hipGetLastError();
    }
    if (nets[0].grade == 1) {                                // fp32-grade networks (split-bf16 x3): their own kernel and packing
        hipLaunchKernelGGL(dn_mlp_x3_kernel<NoTail>, dim3((tiles + 1) / 2, num_nets), dim3(64 * XWAVES), 0, stream, a, NoTail());
        dump_stamps("x3");
        return hipGetLastError();
    }
    const char *e = getenv("DN_MLP_SHAPE");                  // 1 | 4 | 8 waves per workgroup (A/B measurements, tests)
    const int shape = e ? atoi(e) : DN_MLP_DEFAULT_SHAPE;

    const bool f16 = nets[0].grade == 2;                     // float16 operands: the two LDS-fed shapes
    if (shape == 1 && !f16) hipLaunchKernelGGL(dn_mlp_kernel, dim3(tiles, num_nets), dim3(64), 0, stream, a);
    else if (shape == 8) {
        if (f16) hipLaunchKernelGGL((dn_mlp_pair_kernel<true, NoTail>), dim3((tiles + 3) / 4, num_nets), dim3(64 * PWAVES), 0, stream, a, NoTail());
        else hipLaunchKernelGGL((dn_mlp_pair_kernel<false, NoTail>), dim3((tiles + 3) / 4, num_nets), dim3(64 * PWAVES), 0, stream, a, NoTail());
        dump_stamps("pair");
    } else {
        if (f16) hipLaunchKernelGGL(dn_mlp_lds_kernel<true>, dim3((tiles + WAVES - 1) / WAVES, num_nets), dim3(64 * WAVES), 0, stream, a);
        else hipLaunchKernelGGL(dn_mlp_lds_kernel<false>, dim3((tiles + WAVES - 1) / WAVES, num_nets), dim3(64 * WAVES), 0, stream, a);
    }
    return hipGetLastError();
}
#endif  // DN_MLP_NO_LAUNCHER
