// dn_fused.hip -- ONE launch per closed-loop step: the policy network and the control step it feeds.
//
// Reference: SB3 OnPolicyAlgorithm.collect_rollouts, once per step for every env of the SubprocVecEnv
// (Sol/Model/PBDroneSimulator.py:261-286, :653-666): policy forward -> DiagGaussianDistribution.sample -> np.clip ->
// env.step.  dn_mlp_forward + dn_step_sampled are that as two launches: the action means make a round trip through
// memory and the step waits for the LAST workgroup of the policy kernel behind a kernel boundary (~1.7 us on this chip,
// profiles/r03_dispatch_floor.txt).  Here the workgroup of dn_mlp_pair_kernel / dn_mlp_x3_kernel that has just evaluated
// the actor for 128 (64) drones steps exactly those drones before it leaves: the head's accumulators go to LDS, and the
// three-wave single step (pqx_step, dn_kernels.hip) runs on the workgroup's first six (three) waves in the LDS the weight
// buffers no longer need.  Same device functions as dn_mlp_forward's pair shape (DN_MLP_SHAPE=8) and dn_step_sampled, hence the same
// bits (tests/test_gpu_round3.py::test_fused_policy_step_equals_forward_then_step_sampled); the critic's workgroups are
// untouched.  Built for the plain configuration of dn_step_sampled without the ground-contact term (what pqx_step covers).
//
// This translation unit includes the two kernel sources for their device functions; neither contributes a host launcher
// here (DN_TU == 3, DN_MLP_NO_LAUNCHER).
#define DN_TU 3
#include "dn_kernels.hip"
#define DN_MLP_NO_LAUNCHER
#include "dn_mlp.hip"

namespace {

// The tail of the actor's workgroups.  TILES = step tiles (64 drones) per workgroup: 2 for the pair kernel (128 drones, 8 waves),
// 1 for the float32-grade kernel (64 drones, 4 waves).  `slot` = this lane's drone within the workgroup when `has_head` (half-0
// waves; lane group 0 holds rows 0..3 of the head = the four action means of drone `slot`).
template <bool NORM>
struct StepTail {
    static constexpr bool active = true;
    DnParams p;
    DnStepIO io;

    template <int TILES>
    __device__ __forceinline__ void run(uint4 *lds, const int wave, const int lane, const bool has_head, const int slot, const f32x16 &head,
                                        const long long wg) const
    {
        // LDS the policy kernel is done with: [TILES x PqxShared<double>] [TILES x 64 float4 means]
        PqxShared<double> *sh = reinterpret_cast<PqxShared<double> *>(lds);
        float4 *s_mean = reinterpret_cast<float4 *>(sh + TILES);
        if (has_head && lane < 32) s_mean[slot] = make_float4(head[0], head[1], head[2], head[3]);
        __syncthreads();                                     // every wave of the workgroup: the weight buffers are free, the means are in
        if (wave >= 3 * TILES) return;                       // the waves without a role leave (a finished wave no longer counts at s_barrier)
        const int group = wave / 3, role = wave - 3 * group;
        const float4 m = s_mean[group * DN_BLOCK + lane];
        pqx_step<double, NORM, false, true, true>(p, io, sh[group], wg * TILES + group, role, (unsigned)lane, (unsigned)(role * DN_BLOCK + lane), m);
    }
};

template <typename TAIL>
void fill_args(MlpArgs &a, const dn_mlp_net *nets, int num_nets, const float *obs, long long n, int obs_dim)
{
    for (int k = 0; k < 2; ++k) {
        const dn_mlp_net &s = nets[k < num_nets ? k : 0];
        a.net[k].w1 = (const uint4 *)s.w1; a.net[k].w2 = (const uint4 *)s.w2; a.net[k].w3 = (const uint4 *)s.w3;
        a.net[k].wh = (const uint4 *)s.wh;
        a.net[k].b1 = s.b1; a.net[k].b2 = s.b2; a.net[k].b3 = s.b3; a.net[k].bh = s.bh;
        a.net[k].out = s.out; a.net[k].out_dim = s.out_dim;
    }
    a.obs = obs; a.row_mask = nullptr; a.n = n; a.obs_dim = obs_dim;
}

template <bool NORM>
hipError_t launch(const DnParams &p, const DnStepIO &io, const dn_mlp_net *nets, int num_nets, const float *obs, int obs_dim, hipStream_t stream)
{
    static_assert(2 * sizeof(PqxShared<double>) + 2 * DN_BLOCK * sizeof(float4) <= 2 * CHUNK * 64 * sizeof(uint4), "the tail must fit the weight buffers");
    MlpArgs a;
    fill_args<StepTail<NORM>>(a, nets, num_nets, obs, p.n, obs_dim);
    StepTail<NORM> tail;
    tail.p = p;
    tail.io = io;
    const unsigned tiles = (unsigned)((p.n + TILE - 1) / TILE);
    if (nets[0].grade == 1)
        hipLaunchKernelGGL((dn_mlp_x3_kernel<StepTail<NORM>>), dim3((tiles + 1) / 2, num_nets), dim3(64 * XWAVES), 0, stream, a, tail);
    else if (nets[0].grade == 2)
        hipLaunchKernelGGL((dn_mlp_pair_kernel<true, StepTail<NORM>>), dim3((tiles + 3) / 4, num_nets), dim3(64 * PWAVES), 0, stream, a, tail);
    else
        hipLaunchKernelGGL((dn_mlp_pair_kernel<false, StepTail<NORM>>), dim3((tiles + 3) / 4, num_nets), dim3(64 * PWAVES), 0, stream, a, tail);
    return hipGetLastError();
}

}  // namespace

// nets[0] = the actor (its workgroups carry the step), nets[1] (optional) = the critic; io as for dn_step_sampled with io.mean unused.
hipError_t dn_launch_mlp_step(const DnParams &p, const DnStepIO &io, const dn_mlp_net *nets, int num_nets, const float *obs, int obs_dim,
                              hipStream_t stream)
{
    return p.normalize_obs ? launch<true>(p, io, nets, num_nets, obs, obs_dim, stream) : launch<false>(p, io, nets, num_nets, obs, obs_dim, stream);
}
