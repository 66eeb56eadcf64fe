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
// v_exp_f32 / v_rcp_f32 (absolute error ~1e-7, far inside bf16), float32 output of the head.
#include "dn_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_DEV __device__ __forceinline__

constexpr int H1 = 512, H2 = 512, H3 = 256;        // net_arch of PBDroneSimulator.py:251-258
constexpr int TILE = 32;                           // drones per wavefront (the N of the 32x32x16 MFMA)

struct MlpNetDev {
    const uint4 *w1, *w2, *w3, *wh;
    const float *b1, *b2, *b3, *bh;
    float *out;
    int out_dim;
};
struct MlpArgs {
    MlpNetDev net[2];
    const float *obs;
    const uint8_t *row_mask;       // optional: a tile none of whose drones is flagged writes zeros and skips the network
    long long n;
    int obs_dim;
};

MLP_DEV float tanh_fast(float x)
{
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// Row (output feature) of accumulator register r in lane group g of M-tile m: the C/D map of the 32x32 MFMA.
MLP_DEV int acc_row(int m, int g, int r) { return 32 * m + 4 * g + (r & 3) + 8 * (r >> 2); }

// One Linear(+Tanh) layer for this wave's 32 drones.  KS = K-steps of 16 input features, MT = M-tiles of 32 output
// features.  in[kk] is the B operand of K-step kk; out[2m], out[2m+1] become K-steps 2m, 2m+1 of the next layer.
// The layer's MT*KS weight fragments are consumed in storage order; a ring of RING fragments (16 B per lane each)
// is kept in flight ahead of the MFMA that uses them, because one fragment is one L2 round trip (~200+ cycles) and an
// MFMA only 32: without the ring the wave waits on every load.
template <int KS, int MT, bool TANH, int RING>
MLP_DEV void layer(const uint4 *__restrict__ w, const float *__restrict__ bias, const bf16x8 (&in)[KS], bf16x8 (&out)[2 * MT],
                   const int lane)
{
    const int g = lane >> 5;
    constexpr int T = KS * MT;
    constexpr int P = RING < T ? RING : T;
    const uint4 *wl = w + lane;
    uint4 ring[P];
#pragma unroll
    for (int t = 0; t < P; ++t) ring[t] = wl[t * 64];
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
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), in[kk], acc, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 o;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float v = acc[8 * h + s];
                o[s] = (__bf16)(TANH ? tanh_fast(v) : v);
            }
            out[2 * m + h] = o;
        }
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
    bf16x8 x0[1];
    {
        const float *o = a.obs + row * a.obs_dim;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = 8 * g + s;
            x0[0][s] = (__bf16)(k < a.obs_dim ? o[k] : 0.0f);
        }
    }
    bf16x8 h1[H1 / 16];
    layer<1, H1 / 32, true, 16>(net.w1, net.b1, x0, h1, lane);
    bf16x8 h2[H2 / 16];
    layer<H1 / 16, H2 / 32, true, 16>(net.w2, net.b2, h1, h2, lane);
    bf16x8 h3[H3 / 16];
    layer<H2 / 16, H3 / 32, true, 16>(net.w3, net.b3, h2, h3, lane);

    // head: one M-tile (out_dim <= 32 rows, zero-padded weights), float32 result straight from the accumulator
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = net.bh[acc_row(0, g, r)];
#pragma unroll
    for (int kk = 0; kk < H3 / 16; ++kk) {
        const uint4 w = net.wh[kk * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), h3[kk], acc, 0, 0, 0);
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

}  // namespace

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
    hipLaunchKernelGGL(dn_mlp_kernel, dim3(tiles, num_nets), dim3(64), 0, stream, a);
    return hipGetLastError();
}
