/*
 * dronenav.h -- C ABI of libdronenav.so: the MI355X-native vectorised drone-navigation
 * environment (hand-written HIP for gfx950).
 *
 * This is the drop-in boundary for the reference's hot path.  In the reference
 * (eRGiBi/DRL-DroneNavigation, pure Python, paths relative to /root/reference) that path is
 *     SB3 SubprocVecEnv(...)                          Sol/Model/PBDroneSimulator.py:653-666
 *       -> Monitor(NormalizeObservation(PBDroneEnv))  Sol/Model/PBDroneSimulator.py:154-196
 *         -> PBDroneEnv.step / reset                  Sol/Model/Environments/PBDroneEnv.py:171,609
 *           -> BaseAviary.step / reset                Sol/PyBullet/BaseAviary.py:324,276
 *             -> pybullet.stepSimulation              Sol/PyBullet/BaseAviary.py:439-440
 * i.e. N worker processes with one PyBullet world and one drone each.  Here one dn_env holds all
 * N drones of one GPU; dn_step() advances every drone by one control step (240 Hz) in a single
 * kernel launch and applies the VecEnv auto-reset in the same launch.
 *
 * Conventions
 *   - plain C, no torch types; all *device* pointers are raw HIP device addresses (e.g.
 *     tensor.data_ptr()), `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - the caller owns every I/O buffer; the library owns the persistent per-drone state allocated
 *     by dn_create() and freed by dn_destroy(); no caller buffer is retained across calls.
 *   - every function returns DN_OK (0) or a negative dn_status; dn_last_error() returns a
 *     thread-local message.  HIP errors are surfaced, never abort()ed.
 *   - a dn_env is used from one host thread at a time; work is enqueued on the caller's stream and
 *     the calls do not synchronise unless stated.
 *   - there is NO CPU fallback: without a HIP device dn_create() fails with DN_ERR_NO_DEVICE.
 */
#ifndef DRONENAV_H
#define DRONENAV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_ABI_VERSION 9
#define DN_MAX_WAYPOINTS 64
#define DN_OBS_DIM 13      /* 12 kinematic + distance, PBDroneEnv._computeObs, PBDroneEnv.py:296-336 */
#define DN_ACT_DIM 4       /* four rotor thrust commands, PBDroneEnv._actionSpace, PBDroneEnv.py:225-243 */
#define DN_GROUND_CONTACT_AUTO 2   /* dn_config.ground_contact: resolved by dn_create (see the field) */

typedef enum dn_status {
    DN_OK = 0,
    DN_ERR_INVALID_ARGUMENT = -1,
    DN_ERR_HIP = -2,
    DN_ERR_OUT_OF_MEMORY = -3,
    DN_ERR_NO_DEVICE = -4,
    DN_ERR_BAD_STATE = -5
} dn_status;

/* Replaces the constructor arguments of PBDroneEnv (PBDroneEnv.py:41-65) as filled in by
 * PBDroneSimulator.make_env (PBDroneSimulator.py:154-171), plus the wrapper switches of
 * make_env (:181-196) and the SubprocVecEnv size (:653-666). */
typedef struct dn_config {
    int64_t num_envs;                           /* drones on this GPU (SubprocVecEnv's n_envs) */
    int32_t device_id;                          /* HIP device ordinal */
    int32_t num_waypoints;                      /* len(target_points), 1..DN_MAX_WAYPOINTS */
    double waypoints[DN_MAX_WAYPOINTS * 3];     /* target_points, row-major xyz */
    double spawn[3];                            /* initial_xyzs[0] */
    double aviary_dim[6];                       /* x_low y_low z_low x_high y_high z_high */
    double threshold;                           /* gate radius, 0.3 in PBDroneSimulator.py:116 */
    int32_t max_steps;                          /* --max_env_steps */
    int32_t circle;                             /* Track.is_circle: torus corridor around the unit circle at z=1 */
    int32_t cylinder;                           /* corridor check on (make_env passes True) */
    int32_t include_distance;                   /* obs[12] = distance/max_target_dist (True in the driver) */
    int32_t normalize_actions;                  /* PBDroneEnv.rescale_action (True in the driver) */
    int32_t normalize_obs;                      /* per-drone normalize.NormalizeObservation (always on in make_env) */
    int32_t ground_contact;                     /* len(p.getContactPoints())>0 vs plane.urdf (PBDroneEnv.py:699), which the reference always tests,
                                                   APPROXIMATED as: lowest point of the collision cylinder within Bullet's contact margin of z = 0
                                                   (the one term of the step that is neither pinned nor exact).  0 off | 1 on |
                                                   DN_GROUND_CONTACT_AUTO (2, the default): on unless the term is provably unreachable -- corridor
                                                   test on and every point low enough to touch the floor already outside the corridor of every track
                                                   segment, so that `terminated` cannot depend on it (true for the circle tracks at z = 1 and the
                                                   8-gate race track; false for the registry tracks that spawn at z = 0.1).  dn_create resolves it;
                                                   dn_get_config returns the resolved value */
    int32_t compute_f32;                        /* 0: float64 arithmetic in registers over the float32 state
                                                      (parity grade, default); 1: float32 arithmetic */
    float act_noise_sigma;                      /* sim-to-real: Gaussian action noise (0 = reference) */
    float obs_noise_sigma;                      /* sim-to-real: Gaussian observation noise (0 = reference) */
    uint64_t seed;                              /* Philox key for the noise streams */
    int64_t env_id_offset;                      /* global id of drone 0 (rank * num_envs when sharded) */
    int32_t clip_rew;                           /* --clip_rew: TransformReward(clip(r, -10, 10)), PBDroneSimulator.py:191-192 */
    int32_t norm_rew;                           /* --norm_rew: NormalizeReward(gamma .99, eps 1e-8), PBDroneSimulator.py:193-194
                                                   (normalize.py:100-147); both sit inside Monitor, clip first */
    int32_t physics;                            /* enums.Physics (enums.py:12-21) as dispatched by BaseAviary.step (BaseAviary.py:412-437):
                                                   0 PYB (what the reference always runs: :411 pins it) | 1 PYB_GND (_groundEffect, :800-832)
                                                   | 2 PYB_DRAG (_drag, :836-862) | 3 PYB_DW (_downwash: other drones of the same world,
                                                   none here -> as PYB) | 4 PYB_GND_DRAG_DW */
    int32_t action_type;                        /* 0 ActionType.THRUST (PBDroneEnv._preprocessAction, PBDroneEnv.py:872-895)
                                                   | 1 ActionType.RPM (BaseSingleAgentAviary.py:176-179: rpm = HOVER_RPM (1 + 0.05 a))
                                                   | 2 PID (a[0:3] = destination) | 3 VEL (a[0:3] direction, |a[3]| speed) | 4 ONE_D_RPM (a[0])
                                                   | 5 ONE_D_PID (a[0]): BaseSingleAgentAviary._preprocessAction (:180-222) with the
                                                   DSLPIDControl loop (Sol/PyBullet/DSLPIDControl.py) per drone; the action buffer stays [N, 4] */
    int32_t random_spawn;                       /* PBDroneEnv(random_spawn=True): every episode starts at a random point around a random track line
                                                   (PositionGenerator.generate_random_point_around_line, position_generator.py:121-152, max_distance
                                                   0.1, fed by the dormant block PBDroneEnv.py:622-627); draws are Philox words keyed by `seed`, the
                                                   global drone id and the vector step, so sharding does not move them.  0 = the reference as it runs */
    int32_t zero_damping;                       /* p.changeDynamics(linearDamping=0, angularDamping=0): the line the reference keeps commented out
                                                   (BaseAviary.py:571-573).  0 = Bullet's default damping 0.04 (1 + |v|), what the reference simulates */
} dn_config;

/* One drone's persistent state, host-side AoS view used by dn_get_state/dn_set_state (tests,
 * checkpointing).  Field names follow the reference's attributes. */
typedef struct dn_env_state {
    float pos[3], quat[4], vel[3], ang_v[3];    /* Bullet base state, BaseAviary.py:596-598 (quat = x,y,z,w) */
    float prev_vel[3], prev_ang_v[3];           /* PBDroneEnv.prev_vel / prev_ang_v */
    float cur_pos[3];                           /* PBDroneEnv._current_position (stale copy, quirk Q3) */
    float d, d_prev;                            /* _distance_to_target, _prev_distance_to_target */
    int32_t idx;                                /* _current_target_index */
    int32_t steps;                              /* _steps */
    int32_t just_found;                         /* just_found */
    float ep_ret;                               /* Monitor: running episode return (high word, see ep_ret_lo) */
    int32_t ep_len;                             /* Monitor: running episode length */
    double rms_mean[DN_OBS_DIM];                /* normalize.RunningMeanStd.mean  (normalize_obs only) */
    double rms_var[DN_OBS_DIM];                 /*                         .var  (the device carries the second moment var x count; these calls convert) */
    double rms_count;                           /*                         .count                     */
    double rr_returns;                          /* NormalizeReward.returns (discounted return, norm_rew only)  */
    double rr_mean, rr_var, rr_count;           /* NormalizeReward.return_rms                                  */
    float last_rpm[4];                          /* BaseAviary.last_clipped_action (physics with drag only; zeros otherwise) */
    double pid[9];                              /* DSLPIDControl.integral_pos_e, .last_rpy, .integral_rpy_e (action types PID / VEL / ONE_D_PID) */
    float ep_ret_lo;                            /* Monitor: low part of the running return -- the return is ep_ret + ep_ret_lo, ep_ret_lo a multiple k/256
                                                   (k a signed byte) of ep_ret's ulp, so that the float64 sum SB3's Monitor keeps is not re-rounded to
                                                   24 bits every step; dn_set_state rounds what it is given to that grid */
} dn_env_state;

/* Wave-reduced episode statistics accumulated on the device since dn_create / dn_reset_stats. */
typedef struct dn_stats {
    int64_t env_steps;                          /* drone steps simulated */
    int64_t episodes;                           /* episodes finished (done flags raised) */
    int64_t truncated;                          /* of which TimeLimit.truncated */
    int64_t completed;                          /* of which all waypoints reached (+200 branch) */
    int64_t sum_ep_len;                         /* sum of Monitor 'l' */
    int64_t sum_found_targets;                  /* sum of info['found_targets'] at episode end */
    double sum_ep_return;                       /* sum of Monitor 'r' (fixed-point 1e-6 accumulation) */
} dn_stats;

typedef struct dn_env dn_env;

int32_t dn_abi_version(void);
const char *dn_last_error(void);
int32_t dn_device_count(void);

/* Fills *cfg with the driver's literals (threshold 0.3, max_steps 4096, cylinder, include_distance,
 * normalize_actions on; ground_contact = DN_GROUND_CONTACT_AUTO; circle/normalize_obs/noise off) and an empty track. */
void dn_config_default(dn_config *cfg);

/* Replaces N x PBDroneEnv.__init__ + the env.reset(seed=seed+rank) of make_env
 * (PBDroneSimulator.py:154-173).  Allocates the device state. */
int32_t dn_create(const dn_config *cfg, dn_env **out);
int32_t dn_destroy(dn_env *env);
int64_t dn_num_envs(const dn_env *env);
/* The configuration the environment runs with: *out = the dn_config given to dn_create with ground_contact resolved
 * to 0 / 1 (ABI 6).  What `PBDroneEnv.__dict__` answers in the reference. */
int32_t dn_get_config(const dn_env *env, dn_config *out);
/* What dn_create makes of cfg->ground_contact (host arithmetic on the track geometry only, no device needed): 0 / 1, or a
 * negative dn_status for an invalid configuration (ABI 6). */
int32_t dn_resolve_ground_contact(const dn_config *cfg);
/* Compute units of the environment's device (hipDeviceProp_t.multiProcessorCount): the kernel-shape crossovers below are
 * tiles (64 drones) per CU, calibrated on the 256-CU MI355X (ABI 6). */
int32_t dn_get_num_cus(const dn_env *env);

/* Replaces VecEnv.reset() -> N x Monitor.reset/NormalizeObservation.reset/PBDroneEnv.reset
 * (PBDroneEnv.py:609-665, BaseAviary.py:276-320).  obs: device float[N*13]. */
int32_t dn_reset(dn_env *env, float *obs, void *stream);

/* Replaces VecEnv.step_async+step_wait -> N x worker step (PBDroneEnv.step, PBDroneEnv.py:171-199)
 * with SubprocVecEnv auto-reset and Monitor statistics.  All pointers are device pointers:
 *   actions       const float[N*4]   policy output in [-1,1] (action_space, PBDroneEnv.py:230-236)
 *   obs           float[N*13]        next observation (already the reset observation where done)
 *   reward        float[N]
 *   done          uint8[N]           terminated || truncated
 *   truncated     uint8[N]           info["TimeLimit.truncated"] = truncated && !terminated
 *   found_targets int32[N]           info["found_targets"] (PBDroneEnv.py:442)
 *   terminal_obs  float[N*13]|NULL   info["terminal_observation"]; rows written only where done
 *   ep_return     float[N]|NULL      Monitor info["episode"]["r"]; written only where done
 *   ep_length     int32[N]|NULL      Monitor info["episode"]["l"]; written only where done
 *   done_mask     uint64[ceil(N/64)]|NULL  one wave-ballot word per 64 drones (bit l = drone 64*w+l done) */
int32_t dn_step(dn_env *env, const float *actions, float *obs, float *reward, uint8_t *done,
                uint8_t *truncated, int32_t *found_targets, float *terminal_obs, float *ep_return,
                int32_t *ep_length, uint64_t *done_mask, void *stream);

/* k consecutive control steps enqueued back-to-back (open-loop action sequences: replays, random-action
 * collection, benchmarks).  Every buffer is step-major [k, N, ...] -- the (n_steps, n_envs, ...) layout of an
 * SB3 RolloutBuffer -- with the same meaning and optionality as in dn_step; done_mask is [k, ceil(N/64)].
 * N must be a multiple of 4 so that every step's obs slice stays 16-byte aligned. */
int32_t dn_step_many(dn_env *env, int64_t k, const float *actions, float *obs, float *reward, uint8_t *done,
                     uint8_t *truncated, int32_t *found_targets, float *terminal_obs, float *ep_return,
                     int32_t *ep_length, uint64_t *done_mask, void *stream);

/* Rows A5-A9 of a control step on their own (plus the A10/A11 wrappers): the rigid-body transition of the step -- what
 * p.stepSimulation (BaseAviary.py:439-440) leaves in Bullet -- is GIVEN, and everything the reference does with it runs
 * through the same device code as dn_step: _updateAndStoreKinematicInformation / getEulerFromQuaternion
 * (BaseAviary.py:588-598), _computeObs (PBDroneEnv.py:296-398), _computeReward (:475-607), _computeTerminated /
 * _computeTruncated (:444-473, :678-786), _update_state_post_step (:196-223), and on done the SubprocVecEnv auto-reset
 * (:609-665) with Monitor's record.  The persistent state advances exactly as in dn_step (the given pose and velocities
 * become the body state).  Exists so that fixtures which script a kinematic sequence reach the HIP path directly.
 *   kinematics    const double[N*13]  per drone pos(3) quat(4: x,y,z,w, unit) vel(3) ang_v(3), world frame, float64 as
 *                                     the reference holds them (the velocities are rounded to float32, the state's type)
 *   other buffers as in dn_step.  Reference configuration only (no noise / reward wrappers / extra physics). */
int32_t dn_eval_kinematics(dn_env *env, const double *kinematics, float *obs, float *reward, uint8_t *done,
                           uint8_t *truncated, int32_t *found_targets, float *terminal_obs, float *ep_return,
                           int32_t *ep_length, void *stream);

/* Episode-done compaction: expands the per-wave ballot words written by dn_step into the ordered
 * list of finished drones (what the host needs to build the per-env `infos` of SubprocVecEnv
 * without scanning N flags).  indices: device int32[N]; count: device int32[1]. */
int32_t dn_compact_done(const uint64_t *done_mask, int64_t num_envs, int32_t *indices, int32_t *count,
                        int32_t device_id, void *stream);

/* dn_compact_done plus, for every finished drone in index order, its episode-end record as one row of 16 float32 words:
 * terminal_observation[13] (SubprocVecEnv's info["terminal_observation"]), Monitor's episode return, its length (int32 bits) and
 * TimeLimit.truncated | found_targets << 8 (int32 bits) -- what SubprocVecEnv's worker puts into `info` when an episode ends
 * (PBDroneSimulator.py:653-666 with Monitor, :196).  The host copies count x 64 bytes instead of four whole per-drone arrays.
 * Inputs: the buffers the last dn_step wrote.  indices: device int32[N]; count: device int32[1]; packed: device float[N x 16]. */
int32_t dn_pack_done(const uint64_t *done_mask, int64_t num_envs, const float *terminal_obs, const float *ep_return, const int32_t *ep_length,
                     const uint8_t *truncated, const int32_t *found_targets, int32_t *indices, int32_t *count, float *packed,
                     int32_t device_id, void *stream);

/* Measurement helper ("a measured stream-copy ceiling on the box"): a hand-written float4 copy of `bytes` bytes
 * (a multiple of 16, both pointers 16-byte aligned, device memory) -- one 16-byte load and one 16-byte store per lane, one lane per
 * 16 bytes (the form that measured fastest: profiles/r04_copy_sweep.txt) -- enqueued on `stream`.  What bench.py quotes as `hbm_copy_ceiling` beside the
 * nominal 8 TB/s; it replaces no reference code. */
int32_t dn_stream_copy(void *dst, const void *src, int64_t bytes, int32_t device_id, void *stream);

/* Host <-> device copies of the whole persistent state (synchronous).  states: host array [N]. */
int32_t dn_get_state(dn_env *env, dn_env_state *states, int64_t count);
int32_t dn_set_state(dn_env *env, const dn_env_state *states, int64_t count);

/* Episode statistics (synchronises `stream`). */
int32_t dn_get_stats(dn_env *env, dn_stats *out, void *stream);
int32_t dn_reset_stats(dn_env *env, void *stream);

/* Kernel shape chosen for this environment's launches (fused != 0: dn_step_many with k > 1; fused == 0: dn_step), as waves
 * per 64-drone tile.  Fused: 8 = the role-pipelined kernel (thrust | linear + flags | angular | attitude | distance bookkeeping +
 * reward terms | scalars | the normaliser's two column halves; plain configuration with normalize_obs, no noise: up to 1 tile per CU
 * and from 2 to 3 tiles per CU), 5 = the four-wave shape
 * with the observation normaliser on a wave of its own (normalize_obs in the plain configuration, 1 to 2 tiles per CU -- 32 768 drones
 * on 256 CUs -- and with noise up to 3), 4 = the recurrence itself on two waves (linear + rules | angular + attitude) plus an observation
 * and a report wave (plain configuration up to 3 tiles per CU), 3 = flight + report + aux wave, 2 = a flight wave + a report wave
 * (mid-size fleets with the optional terms, where more waves cost occupancy), 1 = one wave.
 * Single step: 3 = three waves cut by dependency (dn_step_pqx_kernel, plain configuration up to 4 tiles per CU), 1 = one wave.
 * The multi-wave shapes win while the tiles alone leave SIMDs idle; crossovers are tiles per CU (dn_get_num_cus).  All shapes
 * produce identical bits -- on one device: the observation-noise draws (obs_noise_sigma > 0; 13 of a noisy step's 17 normals) use the
 * hardware's float32 log2 / sqrt / sin / cos (within 1.2e-6 of the float64 definition, tests/test_gpu_parity.py::
 * test_observation_noise_draws_match_their_definition), so noisy runs are bit-reproducible across kernel shapes, launches and shards
 * of one GPU generation, not across generations or against a CPU evaluation; everything that feeds the dynamics (action noise, policy
 * sampling, random spawn) and the whole noise-free configuration is exact arithmetic.  DN_EXACT_OBS_NOISE=1 (read by dn_create) draws the
 * observation noise in the exact form as well (bit-equal to the float64 definition for > 99.99 % of draws, one float32 ulp otherwise).
 * Environment variables DN_WAVES=1|2|3|4|5|8 and DN_WAVES_SINGLE=1|3 (read by dn_create) force a shape. */
int32_t dn_get_kernel_waves(const dn_env *env, int32_t fused);

/* The arithmetic switches dn_create resolved from the process environment (ABI 9), as a bit mask: a checkpoint or a shard restored in a
 * process with another setting would otherwise replay other observation noise / other last bits silently -- compare this value.
 *   DN_EXACT_FLAG_OBS_NOISE  DN_EXACT_OBS_NOISE=1: observation noise in the exact float64 Box-Muller form (see above).
 *   DN_EXACT_FLAG_NORM       DN_EXACT_NORM=1: the observation normaliser's OUTPUT stage in float64.  The running statistics are float64
 *                            and updated by the same expressions either way (normalize.py:34-47); the normalised value
 *                            (obs - mean) / sqrt(var + 1e-8) (normalize.py:94-97) leaves as a float32, and by default it is formed in
 *                            float32 on the hardware reciprocal square root: within 3 float32 ulp (3.6e-7 relative) of the float64
 *                            evaluation, 30x inside the 1e-5 parity bar, for a third less vector-ALU time in the normaliser.  With the
 *                            switch the output is the float32 nearest to the float64 evaluation (1/2 ulp).
 * Accepted values: 1 | true | on | yes and 0 | false | off | no (any case; unset or empty = off); anything else fails dn_create with
 * DN_ERR_INVALID_ARGUMENT. */
#define DN_EXACT_FLAG_OBS_NOISE 1
#define DN_EXACT_FLAG_NORM 2
int32_t dn_get_exact_flags(const dn_env *env);

/* Vector-step counter: the Philox counter word of the noise streams and the source of dn_stats.env_steps.  It
 * lives on the device and is advanced by the step kernels themselves, so dn_step / dn_step_many launches captured
 * into a hipGraph keep counting when the graph is replayed.  Both calls synchronise the device. */
int32_t dn_get_step_count(const dn_env *env, uint64_t *out);
int32_t dn_set_step_count(dn_env *env, uint64_t value);

/* The float32 action chain on its own: replaces N x PBDroneEnv._preprocessAction (PBDroneEnv.py:872-895, with
 * rescale_action :949-971 and env_utils.cmd2pwm / pwm2rpm, env_utils.py:8-59) plus the rotor force / torque lines of
 * BaseAviary._physics (BaseAviary.py:776-780).  dn_step runs exactly this code in-kernel; the entry point exists so
 * the chain can be checked bit for bit.  actions: device float[N*4]; rpm, forces: device float[N*4] or NULL;
 * z_torque: device float[N] or NULL (at least one output). */
int32_t dn_preprocess_action(const float *actions, int64_t num_envs, int32_t normalize_actions, float *rpm,
                             float *forces, float *z_torque, int32_t device_id, void *stream);

/* Generalised advantage estimation on device buffers laid out [n_steps, n_envs]
 * (the reference's only in-tree statement of the recursion: Sol/Model/Algorithms/cleanRLPPO.py:234-248).
 * dones[t] is the episode-start flag of step t (cleanRL: dones[t] = next_done before step t),
 * last_values/last_dones are V(s_T) and the done flag after the final step. */
int32_t dn_gae(const float *rewards, const float *values, const uint8_t *dones,
               const float *last_values, const uint8_t *last_dones, int64_t n_steps, int64_t n_envs,
               double gamma, double gae_lambda, float *advantages, float *returns,
               int32_t device_id, void *stream);

/* One network of the reference's agents, by `arch`:
 *   DN_MLP_ARCH_PPO (0)  actor or critic of SB3's ActorCriticPolicy with net_arch pi = vf = [512, 512, 256], Tanh
 *                        (Sol/Model/PBDroneSimulator.py:251-286): obs -> 512 -> 512 -> 256 -> out_dim;
 *   DN_MLP_ARCH_SAC (1)  latent_pi + heads of SB3's SAC Actor with net_arch pi = [256, 256], ReLU
 *                        (Sol/Model/PBDroneSimulator.py:297-303): obs -> 256 -> 256 -> out_dim, the mu and log_std heads stacked
 *                        into one [8, 256] matrix (rows 0..3 mu, 4..7 log_std); w3 / b3 are unused and may be NULL.
 * Weights are bfloat16 in the fragment order of the kernel (drl-dronenavigation_amd/policy_mfma.py::pack_mlp / pack_sac_actor
 * produce it from the [out, in] float32 matrices), biases float32 padded to a multiple of 32 (the head's to 32).  All pointers
 * are device pointers. */
#define DN_MLP_ARCH_PPO 0
#define DN_MLP_ARCH_SAC 1
typedef struct dn_mlp_net {
    const void *w1, *w2, *w3, *wh;              /* packed bf16 weights of the hidden layers and the head */
    const float *b1, *b2, *b3, *bh;             /* biases */
    float *out;                                 /* float[num_envs * out_dim] */
    int32_t out_dim;                            /* 1..32 (4 action means / 1 value / 8 = mu | log_std) */
    int32_t grade;                              /* 0: bf16 weights and activations, float32 accumulate (the speed option, ~1e-3 on the action mean)
                                                   1: float32 grade -- both operands split into two bf16 words, three MFMAs per product
                                                      (pack_mlp / pack_sac_actor(..., grade="fp32") pack the hi / lo fragment streams); matches the
                                                      reference's float32 networks to <= 1e-4.
                                                   2: float16 weights and activations, float32 accumulate: the speed of grade 0 with an
                                                      eighth of its rounding error (11 mantissa bits: ~1e-3 on the action mean).
                                                   All networks of one call share grade and arch. */
    int32_t arch;                               /* DN_MLP_ARCH_* (appended in ABI 5) */
    int32_t reserved_;                          /* 0 */
} dn_mlp_net;

/* Forward pass of one or two such networks over the same observations in one launch (replaces the mlp_extractor +
 * action_net / value_net part of SB3's ActorCriticPolicy.forward / predict_values): a fused MFMA kernel, one wavefront
 * per 32 drones, activations resident in registers.  obs: device float[num_envs * obs_dim], obs_dim <= 16.
 * row_mask: device uint8[num_envs] or NULL; with a mask, a 32-drone tile without a flagged drone writes zeros and
 * skips the network (V(terminal_observation) is needed only where an episode hit the time limit). */
int32_t dn_mlp_forward(const dn_mlp_net *nets, int32_t num_nets, const float *obs, const uint8_t *row_mask,
                       int64_t num_envs, int32_t obs_dim, int32_t device_id, void *stream);

/* The small element-wise steps of SB3's OnPolicyAlgorithm.collect_rollouts around the policy network, as kernels
 * [3P-recall of SB3]:
 *   dn_policy_sample  DiagGaussianDistribution.sample + log_prob and the np.clip(actions, -1, 1) handed to env.step:
 *                     actions = mean + exp(log_std) z with z ~ N(0,1) from the environment's Philox streams (seed, global
 *                     drone id, the vector-step counter; reproducible under hipGraph replay), `clipped` = the copy that goes
 *                     to dn_step, log_prob = sum over the 4 action dims.  mean/actions/clipped: device float[N*4];
 *                     log_std: HOST float[4]; log_prob: device float[N].
 *   dn_add_bootstrap  reward[i] += gamma * terminal_value[i] where truncated[i] (TimeLimit bootstrap). */
int32_t dn_policy_sample(dn_env *env, const float *mean, const float *log_std, uint64_t seed, int32_t deterministic,
                         float *actions, float *clipped, float *log_prob, void *stream);
/* The SAC actor's sampling step on top of dn_mlp_forward(arch = DN_MLP_ARCH_SAC) (replaces SB3's Actor.forward /
 * action_log_prob after latent_pi, Sol/Model/PBDroneSimulator.py:297-338) [3P-recall of SB3]: log_std clamped to [-20, 2],
 * actions = tanh(mu + exp(log_std) z), z ~ N(0,1) from the environment's Philox streams exactly as dn_policy_sample draws it
 * (seed, global drone id, vector-step counter; reproducible under hipGraph replay and independent of the sharding); the result
 * lies inside dn_step's action box.  mu_log_std: device float[N*8], rows (mu[4], log_std[4]); actions: device float[N*4];
 * log_prob: device float[N] or NULL (sum over the four dims of log N(pre; mu, sigma) - log(1 - a^2 + 1e-6)). */
int32_t dn_squashed_sample(dn_env *env, const float *mu_log_std, uint64_t seed, int32_t deterministic, float *actions,
                           float *log_prob, void *stream);
/* dn_squashed_sample + dn_step in one launch (the SAC collection loop's per-step pair): the action is drawn inside the step
 * kernel from the (mu | log_std) rows exactly as dn_squashed_sample draws it (same Philox stream, same bits); actions_out
 * receives it (the replay buffer's action), log_prob_out (may be NULL) its log-probability.  Other arguments as dn_step; same
 * configuration limits as dn_step_sampled. */
int32_t dn_step_squashed(dn_env *env, const float *mu_log_std, uint64_t seed, int32_t deterministic, float *actions_out,
                         float *log_prob_out, float *obs, float *reward, uint8_t *done, uint8_t *truncated,
                         int32_t *found_targets, float *terminal_obs, float *ep_return, int32_t *ep_length,
                         uint64_t *done_mask, void *stream);
int32_t dn_add_bootstrap(float *reward, const float *terminal_value, const uint8_t *truncated, double gamma,
                         int64_t num_envs, int32_t device_id, void *stream);
/* dn_policy_sample + dn_step in one launch (the rollout loop's per-step pair): the action is drawn inside the step kernel
 * from `mean` exactly as dn_policy_sample draws it (same Philox stream, same bits); actions_out receives the UNclipped
 * action (what SB3's collect_rollouts stores), log_prob_out its log-probability, the clipped action goes into the step.
 * Other arguments as dn_step.  Built for the configuration without reward wrappers / extra physics terms / RPM actions /
 * random spawn / zero damping (DN_ERR_INVALID_ARGUMENT otherwise: use dn_policy_sample + dn_step there). */
int32_t dn_step_sampled(dn_env *env, const float *mean, const float *log_std, uint64_t seed, int32_t deterministic,
                        float *actions_out, float *log_prob_out, float *obs, float *reward, uint8_t *done, uint8_t *truncated,
                        int32_t *found_targets, float *terminal_obs, float *ep_return, int32_t *ep_length,
                        uint64_t *done_mask, void *stream);

/* dn_mlp_forward + dn_step_sampled in ONE launch (ABI 6): the policy kernel's workgroup that evaluated the actor (nets[0], out_dim 4)
 * for 128 drones (64 in the float32 grade) draws their actions and runs their control step before it leaves -- no kernel boundary
 * and no round trip of the action means between the policy and the environment (SB3 collect_rollouts' forward -> sample -> clip ->
 * env.step, Sol/Model/PBDroneSimulator.py:261-286).  nets[1] (optional) is the critic, evaluated by the other workgroups of the same
 * launch; both write their `out` as dn_mlp_forward does.  policy_obs: device float[N * obs_dim], the observation the networks read
 * (must not alias `obs`, the step's output).  Same results, bit for bit, as the two calls it replaces when dn_mlp_forward runs its pair
 * shape (DN_MLP_SHAPE=8: the kernel this launch extends; the default four-wave shape sums K in another order).  Limits: PPO arch, the float64
 * reference configuration without noise / reward wrappers / extra physics / ground contact, fleets on which dn_create picked the
 * three-wave single step (dn_get_kernel_waves(env, 0) == 3), num_envs a multiple of 128 (64 in the float32 grade);
 * DN_ERR_INVALID_ARGUMENT otherwise (use the two calls). */
int32_t dn_mlp_step_sampled(dn_env *env, const dn_mlp_net *nets, int32_t num_nets, const float *policy_obs, int32_t obs_dim,
                            const float *log_std, uint64_t seed, int32_t deterministic, float *actions_out, float *log_prob_out,
                            float *obs, float *reward, uint8_t *done, uint8_t *truncated, int32_t *found_targets, float *terminal_obs,
                            float *ep_return, int32_t *ep_length, uint64_t *done_mask, void *stream);

/* Measurement hook (ABI 8; lifetime rules ABI 9).  The step kernel of the NEXT dn_step / dn_step_many / dn_step_sampled / dn_step_squashed call on `env` is dispatched with these two hipEvents
 * (hipEvent_t passed as void *, created with timing enabled; either may be NULL) attached to its own dispatch packet
 * (hipExtLaunchKernelGGL): hipEventElapsedTime(start, stop) is then the duration of that kernel alone -- what a profiler's kernel trace
 * reports -- where a pair of hipEventRecord around the call also times the host's launch path and puts two marker packets on the
 * stream.  One shot: consumed by the next step-family call on `env` whatever its outcome -- a call that fails validation, dn_eval_kinematics
 * and dn_mlp_step_sampled (no hook) DROP the events instead of leaving them armed for a later launch.  An armed launch cannot be
 * captured into a hipGraph (DN_ERR_INVALID_ARGUMENT if the stream is capturing).  It stands where the reference wraps its training loop in cProfile
 * (Sol/Utilities/Profiler.py:5-16), at the granularity this path has: one launch.  Used by bench.py's roofline figure. */
int32_t dn_set_launch_events(dn_env *env, void *start_event, void *stop_event);

/* Bytes of HBM the persistent state of `num_envs` drones occupies (capacity planning). */
int64_t dn_state_bytes(int64_t num_envs, int32_t normalize_obs);

#ifdef __cplusplus
}
#endif
#endif /* DRONENAV_H */
