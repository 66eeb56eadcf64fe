// dn_internal.h -- shared between the HIP kernels (dn_kernels.hip) and the C-ABI host side (dn_capi.cpp).
#ifndef DN_INTERNAL_H
#define DN_INTERNAL_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/dronenav.h"

// 64 drones per workgroup: N = 32768 drones is only 512 tiles, and 512 workgroups spread over all 256 CUs
// (2 per CU, block b -> XCD b % 8) where 128 workgroups of 256 drones would leave half the chip idle.  A
// workgroup is one wave (all phases) or two (flight wave + report wave over the same 64 drones).  Every
// wave-level idiom below (ballot, LDS tile transpose) assumes DN_BLOCK == 64.
#define DN_BLOCK 64

// Layout of one waypoint-table entry (entry k describes waypoint k and the corridor segment that ends
// at it), staged into LDS once per workgroup.  Precomputed on the host in float64 with the reference's
// operation order (PBDroneEnv.is_out_of_cylinder_bounds, PBDroneEnv.py:746-786).
enum {
    DN_T_WP = 0,    // target_points[k]                         (3)
    DN_T_U = 3,     // line_unit_vec of segment k               (3)
    DN_T_E1 = 6,    // extended_point1 = base1 - 0.2*unit       (3)
    DN_T_B1 = 9,    // base1 (spawn for k = 0, else wp[k-1])    (3)
    DN_T_LEXT = 12, // ||extended_point2 - extended_point1||    (1)
    DN_T_LL = 13,   // line_length (0 -> degenerate segment)    (1)
    DN_T_STRIDE = 14
};

// Per-workgroup episode statistics slot (workgroup b always owns drones [64b, 64b+64), so its lane 0
// read-modify-writes the slot without atomics: deterministic, no contended counter).
// The slot also carries the tile's vector-step counter (the Philox counter word of the noise streams and the
// source of dn_stats.env_steps): it advances on the device, so launches replayed from a hipGraph keep counting.
struct DnStatSlot {
    long long episodes, truncated, completed, sum_len, sum_found, sum_ret_fix;
    unsigned long long step_count;
    long long pad_;
};

// Persistent state in HBM: "SoA of float4 groups" -- each group is an array of N float4, lane i reads
// 16 contiguous bytes at i*16 (1 KiB per wave instruction, the full-width coalesced access), instead
// of 26 separate dword arrays.  Field order inside a group is chosen so one drone's step touches six
// groups read + six written; g6 (stale _current_position) is touched only around resets.
struct DnState {
    float4 *g0;  // pos.xyz, d (_distance_to_target)
    float4 *g1;  // quat.xyzw
    float4 *g2;  // vel.xyz, d_prev
    float4 *g3;  // ang_v.xyz, meta bits: steps[0:24) | idx[24:31) | just_found[31]
    float4 *g4;  // prev_vel.xyz, ep_ret (Monitor)
    float4 *g5;  // prev_ang_v.xyz, ep_len bits (Monitor)
    float4 *g6;  // _current_position.xyz (valid while steps == 0; otherwise it equals pos), pad
    float4 *g7;  // BaseAviary.last_clipped_action (previous step's rpm); allocated with Physics.PYB_DRAG only, else NULL
    double *rms_mean;   // [13][N]  normalize.RunningMeanStd.mean
    double *rms_m2;     // [13][N]  RunningMeanStd.var x .count: the second moment (dn_kernels.hip normalize_obs_cols; dn_get_state returns var)
    double *rms_count;  // [N]
    double *rr;         // [4][N]  NormalizeReward: returns, return_rms.mean, .var, .count (norm_rew only)
    double *pid;        // [9][N]  DSLPIDControl: integral_pos_e, last_rpy, integral_rpy_e (action types PID / VEL / ONE_D_PID only)
    DnStatSlot *stats;  // [ceil(N/64)]
};

struct DnStepIO {
    const float *actions;
    float *obs;
    float *reward;
    uint8_t *done;
    uint8_t *truncated;
    int32_t *found_targets;
    float *terminal_obs;
    float *ep_return;
    int32_t *ep_length;
    unsigned long long *done_mask;
    // dn_step_sampled (single-step launches only): the action is drawn in the kernel from the policy's mean instead of read
    const float *mean;                 // [N][4] or nullptr (then `actions` is read)
    float *act_out;                    // [N][4] the sampled, UNclipped action (what SB3 stores in the rollout buffer)
    float *logp_out;                   // [N]    log N(action; mean, std) summed over the four dims
    float log_std[4];
    unsigned long long sample_seed;
    int sample_deterministic;
    int sample_squash;                 // dn_step_squashed: `mean` holds [N][8] rows (mu[4], log_std[4]); action = tanh(mu + sigma z)
};

// Scalars of the environment, in both precisions (the float32 build must not touch float64).
template <typename R>
struct DnConsts {
    R dim[6];
    R spawn[3];
    R threshold;
    R thr_ext;          // threshold + 0.2
    R thr2, thr_ext2;   // squares of the two radii: corridor tests compare squared distances (no sqrt)
    R max_target_dist;
    R inv_max_target_dist;
    R inv_dim[3];       // 1 / (x_high, y_high, z_high): position normalisation as a multiply
    R reset_obs[12];    // observation of the freshly spawned body (BaseAviary.reset, BaseAviary.py:318)
    float reset_obs32[12];   // ... as the float32 words the observation row carries (what the kernels read: half the scalar registers of
                             // the R values, and no float64 -> float32 conversion per column in the reset observation's normaliser pass)
};

struct DnParams {
    DnState st;
    long long n;
    int num_waypoints;
    int max_steps;
    int num_cus;                    // CUs of the device: tile b lands on a CU beside tile b + num_cus (role orders of the multi-wave kernels)
    int circle, cylinder, include_distance, normalize_actions, normalize_obs, ground_contact, clip_rew, norm_rew;
    int gnd, drag, rpm_actions;     // N4: Physics.PYB_GND / PYB_DRAG force terms, ActionType.RPM (1) / ONE_D_RPM (2)
    int pid_mode;                   // N4: 0, or the dn_config.action_type of the DSLPIDControl family: 2 PID | 3 VEL | 5 ONE_D_PID
    int random_spawn;               // N4: episodes start at a Philox-drawn point around a random track line
    int zero_damping;               // N4: changeDynamics(linearDamping=0, angularDamping=0), BaseAviary.py:571-573 (commented out there)
    float act_noise_sigma, obs_noise_sigma;
    int exact_obs_noise;            // DN_EXACT_OBS_NOISE=1 (read by dn_create): the observation-noise draws in the exact float64 form as well
    unsigned long long seed;
    long long env_id_offset;
    const double *tab64;   // [W][DN_T_STRIDE] float64 table
    const float *tab32;    // same, float32
    DnConsts<double> c64;
    DnConsts<float> c32;
};

// dn_set_launch_events (ABI 8): the step kernel of the next dn_step / dn_step_many launch is dispatched with these two hipEvents attached to
// its own dispatch packet (hipExtLaunchKernelGGL) -- they time the kernel itself, like a profiler's kernel trace, where a pair of
// hipEventRecord around the call would also time the host's launch path and add two marker packets to the stream.  One shot: the
// C ABI clears them after the launch.  Thread-local: a dn_env is driven from one host thread at a time (include/dronenav.h).
extern thread_local hipEvent_t dn_tl_ev_start, dn_tl_ev_stop;
#define DN_KLAUNCH(kern, grid, blk, shm, stream, ...)                                                                                   \
    do {                                                                                                                                \
        if (dn_tl_ev_start || dn_tl_ev_stop) hipExtLaunchKernelGGL(kern, grid, blk, shm, stream, dn_tl_ev_start, dn_tl_ev_stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, blk, shm, stream, __VA_ARGS__);                                                             \
    } while (0)

int dn_norm_exact_compiled_in();      // 1 in libdronenav_exact.so (-DDN_NORM_EXACT=1: the normaliser's float64 output stage), else 0
hipError_t dn_launch_step_many(const DnParams &p, const DnStepIO &io, int k, bool f32, int waves, hipStream_t stream);
hipError_t dn_launch_step_many_mw(const DnParams &p, const DnStepIO &io, int k, bool f32, int waves, hipStream_t stream);   // dn_kernels_mw.hip
hipError_t dn_launch_reset(const DnParams &p, float *obs, bool f32, hipStream_t stream);
hipError_t dn_launch_eval_kinematics(const DnParams &p, const DnStepIO &io, const double *kin, bool f32, hipStream_t stream);
hipError_t dn_launch_gae(const float *rewards, const float *values, const uint8_t *dones,
                         const float *last_values, const uint8_t *last_dones, long long T, long long N,
                         float gamma, float gl, float *adv, float *ret, hipStream_t stream);
hipError_t dn_launch_action_chain(const float *actions, long long n, int normalize_actions, float *rpm, float *forces,
                                  float *z_torque, hipStream_t stream);
hipError_t dn_launch_fill4(float4 *dst, float4 v, long long n, hipStream_t stream);
hipError_t dn_launch_filld(double *dst, double v, long long n, hipStream_t stream);
hipError_t dn_launch_squashed_sample(const DnParams &p, const float *mu_log_std, unsigned long long seed, int deterministic,
                                     float *actions, float *log_prob, hipStream_t stream);
hipError_t dn_launch_policy_sample(const DnParams &p, const float *mean, const float *log_std4, unsigned long long seed, int deterministic,
                                   float *actions, float *clipped, float *log_prob, hipStream_t stream);
hipError_t dn_launch_add_bootstrap(float *reward, const float *terminal_value, const uint8_t *truncated, float gamma, long long n,
                                   hipStream_t stream);
hipError_t dn_launch_set_step_count(DnStatSlot *slots, long long blocks, unsigned long long value, hipStream_t stream);
hipError_t dn_launch_mlp(const dn_mlp_net *nets, int num_nets, const float *obs, const uint8_t *row_mask, long long n, int obs_dim,
                         hipStream_t stream);
hipError_t dn_launch_mlp_step(const DnParams &p, const DnStepIO &io, const dn_mlp_net *nets, int num_nets, const float *obs, int obs_dim,
                              hipStream_t stream);                                                                             // dn_fused.hip
hipError_t dn_launch_compact_pack(const unsigned long long *mask, long long n, const float *terminal_obs, const float *ep_return,
                                  const int32_t *ep_length, const uint8_t *truncated, const int32_t *found, int32_t *indices, int32_t *count,
                                  float *packed, hipStream_t stream);
hipError_t dn_launch_stream_copy(void *dst, const void *src, long long n16, int num_cus, hipStream_t stream);
hipError_t dn_launch_compact(const unsigned long long *mask, long long n, int32_t *indices, int32_t *count,
                             hipStream_t stream);

#endif
