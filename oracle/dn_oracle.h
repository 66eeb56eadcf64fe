/*
 * dn_oracle.h -- CPU ORACLE for the drone-navigation environment step.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (eRGiBi/DRL-DroneNavigation, pure Python + PyBullet) for the hot path
 * PBDroneEnv.step -> BaseAviary.step -> p.stepSimulation, plus the SB3
 * SubprocVecEnv/Monitor auto-reset semantics wrapped around it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (drl-dronenavigation_amd/, include/dronenav.h, libdronenav.so) never does.
 *
 * PARITY STATUS
 *   pinned   : rows A1 A2 A3 A6 A7 A8 A9 A10 A12 of SURVEY.md section 8(a) are checked
 *              against golden vectors produced by importing the reference's own Python
 *              (tests/golden/gen_golden.py, fixtures under tests/golden/).
 *   UNPINNED : row A4 (p.stepSimulation) and the Bullet half of A5
 *              (p.getEulerFromQuaternion) live in the third-party `pybullet` wheel
 *              (Bullet3 C++), which is neither vendored under /root/reference nor pinned
 *              by its requirements.txt / uv.lock, and cannot be installed here.  Those two
 *              functions restate Bullet's published algorithm from memory
 *              (btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof,
 *              btMultiBody::stepPositionsMultiDof, pybullet.c getEulerFromQuaternion) and
 *              are "parity unpinned".  Stand-in evidence (tests/test_bullet_invariants.py): closed-form one-step
 *              results, an independent world-frame integrator (1e-12), and through it the reference's own
 *              dead explicit model BaseAviary._dynamics (BaseAviary.py:899-973, fixture dead_dynamics.npz,
 *              1e-13 with the damping switched off) -- the structure of the step is tied to a statement the
 *              reference owns; Bullet's damping law and velocity clamp are recall.
 *
 * All arithmetic follows the reference's dtypes: float32 for the action chain
 * (A1-A3, numpy float32 arrays), float64 for everything else.
 */
#ifndef DN_ORACLE_H
#define DN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_WAYPOINTS 64
#define ORC_OBS_DIM 13

/* Environment configuration == the constructor arguments of PBDroneEnv
 * (Sol/Model/Environments/PBDroneEnv.py:41-65) as passed by
 * PBDroneSimulator.make_env (Sol/Model/PBDroneSimulator.py:154-171). */
typedef struct orc_config {
    int32_t num_waypoints;                      /* len(target_points) */
    double waypoints[ORC_MAX_WAYPOINTS * 3];    /* target_points */
    double spawn[3];                            /* initial_xyzs[0] */
    double dim[6];                              /* aviary_dim: x_low y_low z_low x_high y_high z_high */
    double threshold;                           /* 0.3, PBDroneSimulator.py:116 */
    int32_t max_steps;                          /* args.max_env_steps */
    int32_t circle;                             /* track.is_circle */
    int32_t cylinder;                           /* True in make_env */
    int32_t include_distance;                   /* True in run_full_training */
    int32_t normalize_actions;                  /* True in run_full_training */
    int32_t normalize_obs;                      /* normalize.NormalizeObservation wrapper (always on in make_env) */
    int32_t ground_contact;                     /* approximate len(p.getContactPoints())>0 against plane.urdf */
    int32_t f32_state;                          /* 1: round the stored state to float32 after every vec step
                                                      (mirrors the HIP build's float32 HBM state) */
    /* sim-to-real noise (BASELINE config 5; the reference has none: sigma = 0 is the reference) */
    float act_noise_sigma;
    float obs_noise_sigma;
    uint64_t seed;
    int64_t env_id_offset;                      /* global id of env 0 (rank sharding) */
    /* optional reward wrappers of make_env (PBDroneSimulator.py:191-194), inside Monitor:
     * TransformReward(clip(-10, 10)) if --clip_rew, then gym NormalizeReward() if --norm_rew */
    int32_t clip_rew;
    int32_t norm_rew;
    /* N4 -- options present but unreachable in the reference (BaseAviary.step forces Physics.PYB, BaseAviary.py:411;
     * PBDroneEnv overrides _preprocessAction for ActionType.THRUST only):
     *   physics     0 PYB | 1 PYB_GND | 2 PYB_DRAG | 3 PYB_DW | 4 PYB_GND_DRAG_DW   (enums.py:12-21, BaseAviary.py:412-437;
     *               _downwash sums over OTHER drones of the same Bullet world, NUM_DRONES = 1 -> no force)
     *   action_type 0 THRUST (PBDroneEnv._preprocessAction) | 1 RPM | 2 PID | 3 VEL | 4 ONE_D_RPM | 5 ONE_D_PID
     *               (BaseSingleAgentAviary._preprocessAction, BaseSingleAgentAviary.py:176-222) */
    int32_t physics;
    int32_t action_type;
    /* N4: spawn every episode at a random point around a random track line (PBDroneEnv.py:622-627, dormant in the reference) */
    int32_t random_spawn;
    /* N4: p.changeDynamics(linearDamping=0, angularDamping=0), the line the reference keeps commented out (BaseAviary.py:571-573) */
    int32_t zero_damping;
} orc_config;

/* Every per-env variable the reference keeps, under the reference's names. */
typedef struct orc_env {
    /* Bullet rigid body (world frame), BaseAviary.py:596-598 */
    double pos[3], quat[4], vel[3], ang_v[3];
    double rpy[3];
    /* PBDroneEnv bookkeeping, PBDroneEnv.py:122-145 */
    double cur_pos[3];                          /* _current_position */
    double cur_vel[3], cur_ang_v[3];            /* current_vel, current_ang_v */
    double prev_vel[3], prev_ang_v[3];
    double d, d_prev;                           /* _distance_to_target, _prev_distance_to_target */
    int32_t idx;                                /* _current_target_index */
    int32_t just_found;
    int32_t is_done;                            /* _is_done */
    int32_t steps;                              /* _steps */
    /* SB3 Monitor */
    double ep_ret;
    int32_t ep_len;
    /* normalize.RunningMeanStd, normalize.py:10-31 */
    double rms_mean[ORC_OBS_DIM], rms_var[ORC_OBS_DIM], rms_count;
    /* noise counter (the vector-step counter, 64 bits: it enters the Philox counter whole) */
    uint64_t step_count;
    /* NormalizeReward (normalize.py:100-147): discounted return and its RunningMeanStd(shape=()) */
    double rr_returns, rr_mean, rr_var, rr_count;
    /* BaseAviary.last_clipped_action (BaseAviary.py:442,545): the rpm of the previous control step, zeros after reset */
    double last_clipped_action[4];
    /* DSLPIDControl state (ActionType.PID / VEL / ONE_D_PID): integral_pos_e, last_rpy, integral_rpy_e; never reset */
    double pid[9];
    /* random spawn: global env id (Philox counter word), this episode's INIT_XYZS[0] */
    uint64_t gid;
    double spawn_pt[3];
    int32_t spawn_ready;
} orc_env;

/* Result of one gym-level env.step (PBDroneEnv.step), before vectorisation. */
typedef struct orc_step_out {
    float obs[ORC_OBS_DIM];
    double reward;
    int32_t terminated, truncated, found_targets;
} orc_step_out;

/* ---- A1-A3: action chain, float32 ---------------------------------------- */
void orc_constants(double *out /* [16] */);
void orc_action_bounds(float *a_low, float *a_high);
void orc_rescale_action(const float a[4], float out[4]);
void orc_preprocess_action(const float thrust_cmd[4], float rpm[4]);
void orc_rotor_forces(const float rpm[4], float forces[4], float *z_torque);

/* ---- A4/A5: rigid body (UNPINNED, Bullet recall) ------------------------- */
void orc_bullet_step(double pos[3], double quat[4], double vel[3], double ang_v[3],
                     const double forces[4], double z_torque);
/* same, with an extra LINK_FRAME force on link 4 (centre of mass, BaseAviary._drag) */
void orc_bullet_step_ex(double pos[3], double quat[4], double vel[3], double ang_v[3],
                        const double forces[4], double z_torque, const double body_force[3]);

/* ---- N4: extra force terms and the RPM action type (python halves pinned: extra_physics.npz) ---- */
/* BaseSingleAgentAviary._preprocessAction, ActionType.RPM (BaseSingleAgentAviary.py:176-179) + BaseAviary._physics (:776-780) */
void orc_rpm_action(const float a[4], double rpm[4], double forces[4], double *z_torque);
/* BaseAviary._groundEffect (:800-832): the four forceObj z values, or zeros when the attitude test fails.
 * rpm_is_f32: the rpm array is float32 (THRUST chain) -> numpy works in float32 up to the (PROP_RADIUS/(4h))^2 factor */
void orc_ground_effect(const double pos[3], const double quat[4], const double rpy[3], const double rpm[4], int rpm_is_f32,
                       double out[4]);
/* BaseAviary._drag (:836-862): forceObj handed to link 4 */
void orc_drag(const double quat[4], const double vel[3], const double last_rpm[4], int rpm_is_f32, double out[3]);
void orc_bullet_step_damp(double pos[3], double quat[4], double vel[3], double ang_v[3],
                          const double forces[4], double z_torque, const double body_force[3], double damp);
void orc_euler_from_quat(const double q[4], double rpy[3]);
/* ActionType.PID (2) / VEL (3) / ONE_D_RPM (4) / ONE_D_PID (5): BaseSingleAgentAviary._preprocessAction (:180-222) with
 * DSLPIDControl.computeControl; st[9] = integral_pos_e, last_rpy, integral_rpy_e (python half pinned: pid_control.npz) */
/* position_generator.py:121-152 with the draws supplied; and the Philox-keyed draw of one episode's spawn point */
void orc_point_around_line(const double frm[3], const double to[3], double t, const double rv[3], double offset,
                           const double bounds[6], double out[3]);
void orc_random_spawn(const orc_config *cfg, uint64_t env_id, uint64_t step, double out[3]);
void orc_pid_control(int32_t action_type, const double pos[3], const double quat[4], const double vel[3],
                     const float action[4], double st[9], double rpm[4]);

/* ---- A6-A9: gym-level env -------------------------------------------------- */
void orc_env_construct(const orc_config *cfg, orc_env *e);
void orc_env_reset(const orc_config *cfg, orc_env *e, float obs[ORC_OBS_DIM]);
void orc_env_step(const orc_config *cfg, orc_env *e, const float action[4], orc_step_out *out);
/* pieces, exposed for the golden-vector tests */
void orc_compute_obs(const orc_config *cfg, const orc_env *e, float obs[ORC_OBS_DIM]);
double orc_compute_reward(const orc_config *cfg, orc_env *e);
int32_t orc_compute_terminated(const orc_config *cfg, const orc_env *e);
int32_t orc_compute_truncated(const orc_config *cfg, const orc_env *e);
int32_t orc_has_collision(const orc_config *cfg, const orc_env *e);
void orc_post_step(const orc_config *cfg, orc_env *e);
void orc_normalize_obs(orc_env *e, const float obs_in[ORC_OBS_DIM], double obs_out[ORC_OBS_DIM]);
/* TransformReward(clip) + NormalizeReward.step for one env (normalize.py:132-147); returns the reward Monitor sees */
double orc_reward_wrappers(const orc_config *cfg, orc_env *e, double reward, int32_t done);

/* ---- A10/A11: vectorised (SubprocVecEnv + Monitor + NormalizeObservation) --- */
void orc_vec_create(const orc_config *cfg, orc_env *envs, int64_t n);
void orc_vec_reset(const orc_config *cfg, orc_env *envs, int64_t n, float *obs /* [n,13] */, int threads);
void orc_vec_step(const orc_config *cfg, orc_env *envs, int64_t n, const float *actions /* [n,4] */,
                  float *obs /* [n,13] */, float *reward /* [n] */, uint8_t *done /* [n] */,
                  uint8_t *truncated /* [n] TimeLimit.truncated */, int32_t *found_targets /* [n] */,
                  float *terminal_obs /* [n,13] rows valid where done, may be NULL */,
                  float *ep_ret /* [n] valid where done, may be NULL */,
                  int32_t *ep_len /* [n] valid where done, may be NULL */,
                  uint8_t *terminated /* [n] raw terminated flag, may be NULL */,
                  int threads);

void orc_vec_refresh_rpy(orc_env *envs, int64_t n);   /* teacher-forcing helper: rpy cache <- quat */

/* ---- N1: GAE (cleanRLPPO.py:234-248 + SB3 truncation bootstrap) ------------ */
void orc_gae(const float *rewards, const float *values, const uint8_t *dones,
             const float *last_values, const uint8_t *last_dones,
             int64_t n_steps, int64_t n_envs, double gamma, double lam,
             float *advantages, float *returns);

/* ---- noise: Philox4x32-10 counter RNG -------------------------------------- */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t k0, uint32_t k1, uint32_t out[4]);
void orc_noise4(uint64_t seed, uint64_t env_id, uint64_t step, uint32_t stream, float out[4]);
void orc_noise4_many(uint64_t seed, uint64_t env_id0, int64_t n, uint64_t step, uint32_t stream, float *out);   /* [n][4] */

int32_t orc_sizeof_env(void);
int32_t orc_sizeof_config(void);
int32_t orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
