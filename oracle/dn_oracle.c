/*
 * dn_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's per-drone environment step.  See dn_oracle.h for
 * the parity status: everything here is pinned against golden vectors captured from the
 * reference's own Python, EXCEPT orc_bullet_step() and orc_euler_from_quat(), which restate
 * Bullet3 (third-party, absent from /root/reference, un-pinned) from its published algorithm:
 * "parity unpinned".
 *
 * Citations are file:line relative to /root/reference.  Abbreviations:
 *   PBDroneEnv.py  = Sol/Model/Environments/PBDroneEnv.py
 *   BaseAviary.py  = Sol/PyBullet/BaseAviary.py
 *   env_utils.py   = Sol/Model/env_utils.py
 *   normalize.py   = Sol/Model/Environments/normalize.py
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fopenmp (see oracle/Makefile).
 * -ffp-contract=off matters: numpy evaluates one rounded operation at a time.
 */
#include "dn_oracle.h"

#include <math.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* Constants: Sol/resources/safegym/cf2x.urdf:5,11-12 parsed by               */
/* BaseAviary._parse_urdf_parameters (BaseAviary.py:1123-1163); G at :76.    */
/* ------------------------------------------------------------------------- */
#define ORC_M 0.027
#define ORC_KF 3.16e-10
#define ORC_KM 7.94e-12
#define ORC_IXX 1.4e-5
#define ORC_IYY 1.4e-5
#define ORC_IZZ 2.17e-5
#define ORC_PWM2RPM_SCALE 0.2685
#define ORC_PWM2RPM_CONST 4070.3
#define ORC_MIN_PWM 20000.0
#define ORC_MAX_PWM 65535.0
#define ORC_G 9.8
#define ORC_DT (1.0 / 240.0)      /* PYB_TIMESTEP, BaseAviary.py:82; pyb_freq=ctrl_freq=240 (PBDroneEnv.py:49-50) */
/* Prop-link COM offsets of the URDF that is actually loaded into Bullet:
 * Sol/resources/cf2x.urdf:42,54,66,78 (BaseAviary.py:562-570). */
static const double ORC_PROP_X[4] = {0.028, -0.028, -0.028, 0.028};
static const double ORC_PROP_Y[4] = {-0.028, -0.028, 0.028, 0.028};
/* Collision cylinder of base_link, Sol/resources/cf2x.urdf:34 */
#define ORC_COLL_R 0.06
#define ORC_COLL_H 0.025
/* btMultiBody defaults [3P-recall]: m_linearDamping = m_angularDamping = 0.04,
 * m_maxCoordinateVelocity = 100, gyroscopic term on. The reference leaves the
 * changeDynamics(linearDamping=0, angularDamping=0) line commented out (BaseAviary.py:571-573). */
#define ORC_LIN_DAMP 0.04
#define ORC_ANG_DAMP 0.04
#define ORC_MAX_COORD_VEL 100.0
#define ORC_PI 3.14159265358979323846

void orc_constants(double *out)
{
    out[0] = ORC_M; out[1] = ORC_KF; out[2] = ORC_KM; out[3] = ORC_IXX; out[4] = ORC_IYY; out[5] = ORC_IZZ;
    out[6] = ORC_PWM2RPM_SCALE; out[7] = ORC_PWM2RPM_CONST; out[8] = ORC_MIN_PWM; out[9] = ORC_MAX_PWM;
    out[10] = ORC_G; out[11] = ORC_DT;
    out[12] = ORC_G * ORC_M;                                   /* GRAVITY, BaseAviary.py:164 */
    out[13] = sqrt(ORC_G * ORC_M / (4 * ORC_KF));              /* HOVER_RPM, BaseAviary.py:165 */
    out[14] = ORC_LIN_DAMP; out[15] = ORC_MAX_COORD_VEL;
}

/* PBDroneEnv.py:113-116: a_low/a_high = KF*(SCALE*PWM + CONST)**2, stored as float32. */
void orc_action_bounds(float *a_low, float *a_high)
{
    double lo = ORC_PWM2RPM_SCALE * ORC_MIN_PWM + ORC_PWM2RPM_CONST;
    double hi = ORC_PWM2RPM_SCALE * ORC_MAX_PWM + ORC_PWM2RPM_CONST;
    *a_low = (float)(ORC_KF * (lo * lo));
    *a_high = (float)(ORC_KF * (hi * hi));
}

static inline float clipf(float x, float lo, float hi)
{   /* np.clip == minimum(maximum(x, lo), hi); NaN propagates */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
static inline double clipd(double x, double lo, double hi)
{
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* A1 -- PBDroneEnv.rescale_action, PBDroneEnv.py:949-971.  All operands are float32 arrays
 * (action_space.low/high = -1/+1, physical_action_bounds), so numpy works in float32:
 *   action = low + (high - low) * ((action - min_action) / (max_action - min_action))
 *   action = np.clip(action, low, high)                                            */
void orc_rescale_action(const float a[4], float out[4])
{
    float a_low, a_high;
    orc_action_bounds(&a_low, &a_high);
    const float low = -1.0f, high = 1.0f;
    for (int i = 0; i < 4; ++i) {
        float num = a[i] - a_low;
        float den = a_high - a_low;
        float q = num / den;
        float span = high - low;
        float m = span * q;
        float r = low + m;
        out[i] = clipf(r, low, high);
    }
}

/* A2 -- PBDroneEnv._preprocessAction (PBDroneEnv.py:872-895) + cmd2pwm (env_utils.py:8-41)
 * + pwm2rpm (env_utils.py:44-59).  float32 array (x) python float -> float32. */
void orc_preprocess_action(const float cmd[4], float rpm[4])
{
    float a_low, a_high;
    orc_action_bounds(&a_low, &a_high);
    const float ct = (float)ORC_KF, cst = (float)ORC_PWM2RPM_CONST, scl = (float)ORC_PWM2RPM_SCALE;
    const float pmin = (float)ORC_MIN_PWM, pmax = (float)ORC_MAX_PWM;
    for (int i = 0; i < 4; ++i) {
        float thrust = clipf(cmd[i], a_low, a_high);        /* PBDroneEnv.py:889 */
        if (thrust < 0.0f) thrust = 0.0f;                   /* env_utils.py:29  (NaN stays NaN) */
        float t = thrust / 1.0f;                            /* n_motor = 4 // 4, env_utils.py:28,30 */
        t = t / ct;
        float s = sqrtf(t);
        float pwm = (s - cst) / scl;                        /* env_utils.py:30 */
        pwm = clipf(pwm, pmin, pmax);                       /* env_utils.py:39 */
        float r = scl * pwm;
        rpm[i] = r + cst;                                   /* env_utils.py:58 */
    }
}

/* A3 -- BaseAviary._physics, BaseAviary.py:776-780 (float32 arithmetic on the rpm array). */
void orc_rotor_forces(const float rpm[4], float forces[4], float *z_torque)
{
    const float kf = (float)ORC_KF, km = (float)ORC_KM;
    float tq[4];
    for (int i = 0; i < 4; ++i) {
        float sq = rpm[i] * rpm[i];
        forces[i] = sq * kf;
        tq[i] = sq * km;
    }
    float z = -tq[0];
    z = z + tq[1];
    z = z - tq[2];
    z = z + tq[3];
    *z_torque = z;
}

/* ------------------------------------------------------------------------- */
/* A4 -- p.stepSimulation (BaseAviary.py:439-440).  UNPINNED, [3P-recall] of Bullet3:       */
/* btMultiBodyDynamicsWorld single step for a free-floating btMultiBody whose five child    */
/* links are massless and fixed, i.e. one rigid body.  World set-up: BaseAviary.py:556-570  */
/* (gravity (0,0,-9.8), dt 1/240, URDF_USE_INERTIA_FROM_FILE, default damping kept).        */
/*   solveExternalForces: base force += m*g; ABA for the base in the base frame with        */
/*   damping  m*v_b*(c + c|v_b|)  and  I*w_b*(c + c|w_b|)  and the gyroscopic term;         */
/*   applyDeltaVeeMultiDof: v += a*dt, w += wdot*dt, each coordinate clamped to +-100;      */
/*   stepPositionsMultiDof: x += v*dt; q <- normalize(dq(w*dt) * q)  (exponential map,      */
/*   world-frame angular velocity, Taylor branch below |w| < 1e-3, angle clamp at pi/4).    */
/* Forces: applyExternalForce(link i, [0,0,F_i], posObj=[0,0,0], LINK_FRAME) acts at the    */
/* link's inertial origin (BaseAviary.py:781-788), applyExternalTorque(link 4, [0,0,z],     */
/* LINK_FRAME) (BaseAviary.py:789-794); cleared after the step.                              */
/* ------------------------------------------------------------------------- */
static void quat_to_mat(const double q[4], double R[9])
{   /* btMatrix3x3::setRotation */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double d = x * x + y * y + z * z + w * w;
    double s = 2.0 / d;
    double xs = x * s, ys = y * s, zs = z * s;
    double wx = w * xs, wy = w * ys, wz = w * zs;
    double xx = x * xs, xy = x * ys, xz = x * zs;
    double yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
    R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
}
static inline void mat_vec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
static inline void matT_vec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}
static inline double norm3(const double v[3])
{   /* np.linalg.norm of a 3-vector: sqrt(dot(v, v)) */
    return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

void orc_bullet_step(double pos[3], double quat[4], double vel[3], double ang_v[3],
                     const double forces[4], double z_torque)
{
    const double none[3] = {0.0, 0.0, 0.0};
    orc_bullet_step_ex(pos, quat, vel, ang_v, forces, z_torque, none);
}

void orc_bullet_step_ex(double pos[3], double quat[4], double vel[3], double ang_v[3],
                        const double forces[4], double z_torque, const double body_force[3])
{
    orc_bullet_step_damp(pos, quat, vel, ang_v, forces, z_torque, body_force, ORC_LIN_DAMP);
}

/* damp: btMultiBody's m_linearDamping = m_angularDamping (0.04 by default; 0 is what the reference's commented-out
 * p.changeDynamics(..., linearDamping=0, angularDamping=0) would set, BaseAviary.py:571-573) */
void orc_bullet_step_damp(double pos[3], double quat[4], double vel[3], double ang_v[3],
                          const double forces[4], double z_torque, const double body_force[3], double damp)
{
    const double dt = ORC_DT;
    double R[9];
    quat_to_mat(quat, R);                       /* base -> world */
    double vb[3], wb[3];
    matT_vec(R, vel, vb);
    matT_vec(R, ang_v, wb);

    /* body-frame resultant of the four rotor forces and the yaw torque (A3) */
    double fz = forces[0] + forces[1] + forces[2] + forces[3];
    double tx = 0.0, ty = 0.0;
    for (int i = 0; i < 4; ++i) {               /* r x F with r=(x_i,y_i,0), F=(0,0,F_i) */
        tx += ORC_PROP_Y[i] * forces[i];
        ty -= ORC_PROP_X[i] * forces[i];
    }
    /* gravity is a world-frame base force m*g, rotated into the base frame */
    const double gw[3] = {0.0, 0.0, -ORC_G * ORC_M};
    double gb[3];
    matT_vec(R, gw, gb);

    double nv = norm3(vb), nw = norm3(wb);
    double kl = damp + damp * nv;
    double ka = damp + damp * nw;
    /* body_force: LINK_FRAME force on link 4, whose frame coincides with the base frame (cf2x.urdf:88-98) */
    double Fb[3] = { body_force[0] + gb[0] - ORC_M * vb[0] * kl,
                     body_force[1] + gb[1] - ORC_M * vb[1] * kl,
                     body_force[2] + fz + gb[2] - ORC_M * vb[2] * kl };
    const double I[3] = {ORC_IXX, ORC_IYY, ORC_IZZ};
    double Iw[3] = {I[0] * wb[0], I[1] * wb[1], I[2] * wb[2]};
    double gyro[3] = { wb[1] * Iw[2] - wb[2] * Iw[1],
                       wb[2] * Iw[0] - wb[0] * Iw[2],
                       wb[0] * Iw[1] - wb[1] * Iw[0] };
    double Tb[3] = { tx - gyro[0] - Iw[0] * ka,
                     ty - gyro[1] - Iw[1] * ka,
                     z_torque - gyro[2] - Iw[2] * ka };
    double ab[3] = {Fb[0] / ORC_M, Fb[1] / ORC_M, Fb[2] / ORC_M};
    double wdb[3] = {Tb[0] / I[0], Tb[1] / I[1], Tb[2] / I[2]};
    double aw[3], wdw[3];
    mat_vec(R, ab, aw);
    mat_vec(R, wdb, wdw);

    for (int i = 0; i < 3; ++i) {               /* applyDeltaVeeMultiDof */
        ang_v[i] = clipd(ang_v[i] + wdw[i] * dt, -ORC_MAX_COORD_VEL, ORC_MAX_COORD_VEL);
        vel[i] = clipd(vel[i] + aw[i] * dt, -ORC_MAX_COORD_VEL, ORC_MAX_COORD_VEL);
    }
    for (int i = 0; i < 3; ++i) pos[i] += dt * vel[i];   /* stepPositionsMultiDof */

    double fAngle = norm3(ang_v);
    if (fAngle * dt > 0.25 * ORC_PI) fAngle = 0.5 * (0.5 * ORC_PI) / dt;   /* ANGULAR_MOTION_THRESHOLD */
    double k;
    if (fAngle < 0.001)
        k = 0.5 * dt - (dt * dt * dt) * 0.020833333333 * fAngle * fAngle;
    else
        k = sin(0.5 * fAngle * dt) / fAngle;
    double ax = ang_v[0] * k, ay = ang_v[1] * k, az = ang_v[2] * k, aw_ = cos(fAngle * dt * 0.5);
    /* q <- dq * q (world-frame increment on the base->world quaternion) */
    double x = quat[0], y = quat[1], z = quat[2], w = quat[3];
    double nx = aw_ * x + ax * w + ay * z - az * y;
    double ny = aw_ * y + ay * w + az * x - ax * z;
    double nz = aw_ * z + az * w + ax * y - ay * x;
    double nw_ = aw_ * w - ax * x - ay * y - az * z;
    double inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw_ * nw_);   /* btQuaternion::normalize */
    quat[0] = nx * inv; quat[1] = ny * inv; quat[2] = nz * inv; quat[3] = nw_ * inv;
}

/* A5 -- p.getEulerFromQuaternion (BaseAviary.py:597). UNPINNED, [3P-recall] of pybullet.c. */
void orc_euler_from_quat(const double q[4], double rpy[3])
{
    double sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], squ = q[3] * q[3];
    double sarg = -2.0 * (q[0] * q[2] - q[3] * q[1]);
    if (sarg <= -0.99999) {
        rpy[0] = 0.0; rpy[1] = -0.5 * ORC_PI; rpy[2] = 2.0 * atan2(q[0], -q[1]);
    } else if (sarg >= 0.99999) {
        rpy[0] = 0.0; rpy[1] = 0.5 * ORC_PI; rpy[2] = 2.0 * atan2(-q[0], q[1]);
    } else {
        rpy[0] = atan2(2.0 * (q[1] * q[2] + q[3] * q[0]), squ - sqx - sqy + sqz);
        rpy[1] = asin(sarg);
        rpy[2] = atan2(2.0 * (q[0] * q[1] + q[3] * q[2]), squ + sqx - sqy - sqz);
    }
}

/* ------------------------------------------------------------------------- */
/* A6 -- _getDroneStateVector (BaseAviary.py:623-643) -> _clipAndNormalizeState             */
/* (PBDroneEnv.py:338-398) -> _computeObs (PBDroneEnv.py:296-336).  float64, cast at the end. */
/* ------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------- */
/* N4 -- force terms of Physics.PYB_GND / PYB_DRAG and ActionType.RPM.  The python halves   */
/* (numpy dtypes, operation order) are pinned by tests/golden/extra_physics.npz; the Bullet */
/* halves (p.getLinkStates, p.getMatrixFromQuaternion) are [3P-recall].                      */
/* ------------------------------------------------------------------------- */
#define ORC_GND_EFF_COEFF 11.36859      /* cf2x.urdf:5 */
#define ORC_PROP_RADIUS 2.31348e-2
#define ORC_DRAG_XY 9.1785e-7
#define ORC_DRAG_Z 10.311e-7
#define ORC_GRAVITY (ORC_G * ORC_M)     /* BaseAviary.py:129 */

static double hover_rpm(void) { return sqrt(ORC_GRAVITY / (4 * ORC_KF)); }              /* BaseAviary.py:164 */
static double max_rpm(void) { return sqrt((2.25 * ORC_GRAVITY) / (4 * ORC_KF)); }        /* :165, thrust2weight 2.25 */
static double gnd_eff_h_clip(void)
{   /* BaseAviary.py:175-176 */
    double mr = max_rpm();
    double max_thrust = 4 * ORC_KF * (mr * mr);                                          /* :167 */
    return 0.25 * ORC_PROP_RADIUS * sqrt((15 * (mr * mr) * ORC_KF * ORC_GND_EFF_COEFF) / max_thrust);
}

void orc_rpm_action(const float a[4], double rpm[4], double forces[4], double *z_torque)
{
    const double hr = hover_rpm();
    double tq[4];
    for (int i = 0; i < 4; ++i) {
        float s = 0.05f * a[i];                 /* python float (x) float32 array -> float32 */
        float u = 1.0f + s;
        rpm[i] = hr * (double)u;                /* np.float64 scalar (x) float32 array -> float64 */
        double sq = rpm[i] * rpm[i];            /* BaseAviary.py:776-777 */
        forces[i] = sq * ORC_KF;
        tq[i] = sq * ORC_KM;
    }
    double z = -tq[0];
    z = z + tq[1];
    z = z - tq[2];
    z = z + tq[3];
    *z_torque = z;
}

void orc_ground_effect(const double pos[3], const double quat[4], const double rpy[3], const double rpm[4], int rpm_is_f32,
                       double out[4])
{
    double R[9];
    quat_to_mat(quat, R);
    const double hclip = gnd_eff_h_clip();
    const int ok = fabs(rpy[0]) < ORC_PI / 2 && fabs(rpy[1]) < ORC_PI / 2;               /* :823 */
    for (int i = 0; i < 4; ++i) {
        /* p.getLinkStates(...)[i][0][2]: world z of the link's centre of mass = base + R (x_i, y_i, 0) */
        double h = pos[2] + (R[6] * ORC_PROP_X[i] + R[7] * ORC_PROP_Y[i]);
        if (h < hclip) h = hclip;                                                        /* :821 */
        double q = ORC_PROP_RADIUS / (4 * h);
        double g;
        if (rpm_is_f32) {
            float r = (float)rpm[i];
            float t = r * r;
            t = t * (float)ORC_KF;
            t = t * (float)ORC_GND_EFF_COEFF;
            g = (double)t * (q * q);
        } else {
            g = rpm[i] * rpm[i] * ORC_KF * ORC_GND_EFF_COEFF * (q * q);
        }
        out[i] = ok ? g : 0.0;
    }
}

void orc_drag(const double quat[4], const double vel[3], const double last_rpm[4], int rpm_is_f32, double out[3])
{
    double R[9];
    quat_to_mat(quat, R);                                   /* p.getMatrixFromQuaternion, :850 */
    double sum;
    if (rpm_is_f32) {                                       /* np.sum(np.array(2*np.pi*rpm/60)) in float32 */
        float w[4];
        for (int i = 0; i < 4; ++i) {
            float m = (float)(2 * ORC_PI) * (float)last_rpm[i];
            w[i] = m / 60.0f;
        }
        float s = w[0] + w[1];                              /* add.reduce over four elements: left to right */
        s = s + w[2];
        s = s + w[3];
        sum = (double)s;
    } else {
        double w[4];
        for (int i = 0; i < 4; ++i) w[i] = (2 * ORC_PI) * last_rpm[i] / 60;
        double s = w[0] + w[1];
        s = s + w[2];
        sum = s + w[3];
    }
    const double k[3] = {-1 * ORC_DRAG_XY * sum, -1 * ORC_DRAG_XY * sum, -1 * ORC_DRAG_Z * sum};   /* :852 */
    const double u[3] = {k[0] * vel[0], k[1] * vel[1], k[2] * vel[2]};
    mat_vec(R, u, out);                                     /* np.dot(base_rot, ...) :853 */
}

/* ------------------------------------------------------------------------- */
/* N4 -- ActionType.PID / VEL / ONE_D_RPM / ONE_D_PID: BaseSingleAgentAviary._preprocessAction  */
/* (BaseSingleAgentAviary.py:180-222) around DSLPIDControl.computeControl                       */
/* (Sol/PyBullet/DSLPIDControl.py:78-262).  Pinned by tests/golden/pid_control.npz (the          */
/* reference's own methods; p.getMatrixFromQuaternion / getEulerFromQuaternion are [3P-recall]). */
/* st = integral_pos_e[3], last_rpy[3], integral_rpy_e[3]; the reference never resets it after   */
/* construction (ctrl.reset() is only called by DSLPIDControl.__init__).                          */
/* ------------------------------------------------------------------------- */
static void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
void orc_pid_control(int32_t action_type, const double pos[3], const double quat[4], const double vel[3],
                     const float action[4], double st[9], double rpm[4])
{
    const double dt = ORC_DT;                                  /* CTRL_TIMESTEP = 1 / ctrl_freq */
    if (action_type == 4) {                                    /* ONE_D_RPM: np.repeat(HOVER_RPM * (1 + 0.05 * action), 4), :211-212 */
        float s = 0.05f * action[0];
        float u = 1.0f + s;
        double r = hover_rpm() * (double)u;
        rpm[0] = rpm[1] = rpm[2] = rpm[3] = r;
        return;
    }
    double target_pos[3], target_vel[3] = {0.0, 0.0, 0.0}, target_yaw = 0.0;
    double rpy[3];
    orc_euler_from_quat(quat, rpy);                            /* state[7:10] and cur_rpy */
    if (action_type == 2) {                                    /* PID: _calculateNextStep(pos, action, 1), BaseAviary.py:1255-1298 */
        double dir[3] = {(double)action[0] - pos[0], (double)action[1] - pos[1], (double)action[2] - pos[2]};
        double dist = norm3(dir);
        for (int k = 0; k < 3; ++k) target_pos[k] = dist <= 1.0 ? (double)action[k] : pos[k] + dir[k] / dist * 1.0;
    } else if (action_type == 3) {                             /* VEL, :195-210: float32 arithmetic on the action array */
        float n2 = action[0] * action[0];
        n2 = n2 + action[1] * action[1];
        n2 = n2 + action[2] * action[2];
        float n = sqrtf(n2);
        float lim = 0.25f * fabsf(action[3]);                  /* SPEED_LIMIT = 0.03 * MAX_SPEED_KMH * (1000/3600) = 0.25 */
        for (int k = 0; k < 3; ++k) {
            float u = n != 0.0f ? action[k] / n : 0.0f;
            target_vel[k] = (double)(lim * u);
            target_pos[k] = pos[k];
        }
        target_yaw = rpy[2];                                   /* target_rpy = (0, 0, state[9]) */
    } else {                                                   /* ONE_D_PID, :213-221: state[0:3] + 0.1 * np.array([0, 0, action[0]]) */
        target_pos[0] = pos[0] + 0.1 * 0.0; target_pos[1] = pos[1] + 0.1 * 0.0;
        target_pos[2] = pos[2] + 0.1 * (double)action[0];
    }
    /* _dslPIDPositionControl, DSLPIDControl.py:140-199 */
    double R[9];
    quat_to_mat(quat, R);                                      /* p.getMatrixFromQuaternion */
    const double Pf[3] = {.4, .4, 1.25}, If[3] = {.05, .05, .05}, Df[3] = {.2, .2, .5};
    double pos_e[3], vel_e[3], tt[3];
    for (int k = 0; k < 3; ++k) {
        pos_e[k] = target_pos[k] - pos[k];
        vel_e[k] = target_vel[k] - vel[k];
        st[k] = clipd(st[k] + pos_e[k] * dt, -2.0, 2.0);
    }
    st[2] = clipd(st[2], -0.15, 0.15);
    for (int k = 0; k < 3; ++k) tt[k] = Pf[k] * pos_e[k] + If[k] * st[k] + Df[k] * vel_e[k] + (k == 2 ? ORC_GRAVITY : 0.0);
    double dotz = tt[0] * R[2] + tt[1] * R[5] + tt[2] * R[8];
    double scalar_thrust = dotz > 0.0 ? dotz : 0.0;
    double thrust = (sqrt(scalar_thrust / (4 * ORC_KF)) - ORC_PWM2RPM_CONST) / ORC_PWM2RPM_SCALE;
    double nt = norm3(tt);
    double z_ax[3] = {tt[0] / nt, tt[1] / nt, tt[2] / nt};
    double x_c[3] = {cos(target_yaw), sin(target_yaw), 0.0};
    double y_ax[3], x_ax[3];
    cross3(z_ax, x_c, y_ax);
    double ny = norm3(y_ax);
    y_ax[0] /= ny; y_ax[1] /= ny; y_ax[2] /= ny;
    cross3(y_ax, z_ax, x_ax);
    /* target_rotation = [x_ax y_ax z_ax] (columns).  The reference sends it through scipy: as_euler('XYZ') -> from_euler ->
     * as_quat -> from_quat -> as_matrix (:196, :236-238), a round trip that returns the same rotation to rounding. */
    double Rt[9] = {x_ax[0], y_ax[0], z_ax[0], x_ax[1], y_ax[1], z_ax[1], x_ax[2], y_ax[2], z_ax[2]};
    /* _dslPIDAttitudeControl, :203-262: rot_matrix_e = Rt^T Rc - Rc^T Rt, rot_e = (e[2,1], e[0,2], e[1,0]) */
    double A[9];                                               /* A = Rt^T Rc; then e = A - A^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[3 * i + j] = Rt[i] * R[j] + Rt[3 + i] * R[3 + j] + Rt[6 + i] * R[6 + j];
    double rot_e[3] = {A[7] - A[5], A[2] - A[6], A[3] - A[1]};
    const double Pt[3] = {70000., 70000., 60000.}, It[3] = {.0, .0, 500.}, Dt[3] = {20000., 20000., 12000.};
    double tq[3];
    for (int k = 0; k < 3; ++k) {
        double rate_e = 0.0 - (rpy[k] - st[3 + k]) / dt;
        st[3 + k] = rpy[k];
        st[6 + k] = clipd(st[6 + k] - rot_e[k] * dt, -1500.0, 1500.0);
        if (k < 2) st[6 + k] = clipd(st[6 + k], -1.0, 1.0);
        tq[k] = clipd(-(Pt[k] * rot_e[k]) + Dt[k] * rate_e + It[k] * st[6 + k], -3200.0, 3200.0);
    }
    const double MIX[4][3] = {{-.5, -.5, -1}, {-.5, .5, 1}, {.5, .5, -1}, {.5, -.5, 1}};
    for (int i = 0; i < 4; ++i) {
        double pwm = thrust + (MIX[i][0] * tq[0] + MIX[i][1] * tq[1] + MIX[i][2] * tq[2]);
        pwm = clipd(pwm, ORC_MIN_PWM, ORC_MAX_PWM);
        rpm[i] = ORC_PWM2RPM_SCALE * pwm + ORC_PWM2RPM_CONST;
    }
}

/* ------------------------------------------------------------------------- */
/* N4 -- random spawn around a track line: PositionGenerator.generate_random_point_around_line        */
/* (Sol/Utilities/position_generator.py:121-152; geometry pinned by tests/golden/random_spawn.npz) fed   */
/* by the dormant block of PBDroneEnv.reset (PBDroneEnv.py:622-627: two distinct target points at        */
/* random, max_distance 0.1, bounds = aviary_dim, :168-169).  The reference draws from `random` /         */
/* `np.random`; here the draws are Philox words keyed by (seed; global env id, vector step, streams       */
/* 11 / 12) so that a sharded fleet spawns where the unsharded one does.                                   */
/* ------------------------------------------------------------------------- */
void orc_point_around_line(const double frm[3], const double to[3], double t, const double rv[3], double offset,
                           const double bounds[6], double out[3])
{
    double dir[3] = {to[0] - frm[0], to[1] - frm[1], to[2] - frm[2]};
    double pt[3] = {frm[0] + t * (to[0] - frm[0]), frm[1] + t * (to[1] - frm[1]), frm[2] + t * (to[2] - frm[2])};
    double perp[3];
    cross3(dir, rv, perp);
    double n = norm3(perp);
    for (int k = 0; k < 3; ++k) {
        pt[k] += offset * (perp[k] / n);
        double hi = pt[k] < bounds[3 + k] ? pt[k] : bounds[3 + k];          /* max(lo, min(hi, x)), :113-118 */
        out[k] = bounds[k] > hi ? bounds[k] : hi;
    }
}
void orc_random_spawn(const orc_config *c, uint64_t env_id, uint64_t step, double out[3])
{
    const int W = c->num_waypoints;
    if (W < 2) { out[0] = c->spawn[0]; out[1] = c->spawn[1]; out[2] = c->spawn[2]; return; }   /* no line to spawn around */
    uint32_t r[4];
    orc_philox4x32((uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)step, 11u | ((uint32_t)(step >> 32) << 8),
                   (uint32_t)c->seed, (uint32_t)(c->seed >> 32), r);
    int i = (int)(r[0] % (uint32_t)W), j = (int)(r[1] % (uint32_t)(W - 1));     /* np.random.choice(W, size=2, replace=False) */
    if (j >= i) j += 1;
    double t = ((double)r[2] + 0.5) * (1.0 / 4294967296.0);
    double u = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
    double offset = -0.1 + (0.1 - -0.1) * u;                                    /* random.uniform(-max_distance, max_distance) */
    float z[4];
    orc_noise4(c->seed, env_id, step, 12u, z);                                  /* np.random.randn(3) */
    double rv[3] = {z[0], z[1], z[2]};
    orc_point_around_line(&c->waypoints[3 * i], &c->waypoints[3 * j], t, rv, offset, c->dim, out);
}

static double max_target_dist(const orc_config *c)
{   /* PBDroneEnv.py:91 */
    double a = fabs(c->dim[0]) + c->dim[3], b = fabs(c->dim[1]) + c->dim[4], z = c->dim[5];
    double m = a > b ? a : b;
    return m > z ? m : z;
}

void orc_compute_obs(const orc_config *c, const orc_env *e, float obs[ORC_OBS_DIM])
{
    double o[ORC_OBS_DIM];
    o[0] = e->pos[0] / c->dim[3];                            /* :375 (no clipping of position) */
    o[1] = e->pos[1] / c->dim[4];
    o[2] = e->pos[2] / c->dim[5];                            /* :377 */
    o[3] = clipd(e->rpy[0], -ORC_PI, ORC_PI) / ORC_PI;       /* :361,:379 */
    o[4] = clipd(e->rpy[1], -ORC_PI, ORC_PI) / ORC_PI;
    o[5] = e->rpy[2] / ORC_PI;                               /* :380 */
    o[6] = clipd(e->vel[0], -3.0, 3.0) / 3.0;                /* :362,:381 */
    o[7] = clipd(e->vel[1], -3.0, 3.0) / 3.0;
    o[8] = clipd(e->vel[2], -1.0, 1.0) / 3.0;                /* :363,:382 divides by MAX_LIN_VEL_XY */
    double nw = norm3(e->ang_v);                             /* :383-384 */
    if (nw != 0.0) { o[9] = e->ang_v[0] / nw; o[10] = e->ang_v[1] / nw; o[11] = e->ang_v[2] / nw; }
    else { o[9] = e->ang_v[0]; o[10] = e->ang_v[1]; o[11] = e->ang_v[2]; }
    o[12] = e->d / max_target_dist(c);                       /* :306-307, stale distance (Q1) */
    int n = c->include_distance ? 13 : 12;
    for (int i = 0; i < n; ++i)                              /* :326-327 */
        obs[i] = (float)clipd(o[i], -(double)FLT_MAX, (double)FLT_MAX);
    for (int i = n; i < ORC_OBS_DIM; ++i) obs[i] = 0.0f;
}

/* ------------------------------------------------------------------------- */
/* A8 -- _has_collision_occurred (PBDroneEnv.py:678-707) + is_out_of_cylinder_bounds        */
/* (PBDroneEnv.py:718-786).                                                                 */
/* ------------------------------------------------------------------------- */
static int out_of_cylinder(const orc_config *c, const orc_env *e)
{
    const double *p = e->pos;
    if (c->circle) {                                         /* :723-741, centre (0,0,1), radius 1 (:84) */
        double cv[3] = {p[0] - 0.0, p[1] - 0.0, 0.0};        /* center_to_drone_vec, z zeroed (:729) */
        double n = norm3(cv);
        double nv[3] = {cv[0] / n * 1.0, cv[1] / n * 1.0, cv[2] / n * 1.0};   /* :733 (0/0 -> NaN) */
        double cl[3] = {0.0 + nv[0], 0.0 + nv[1], 1.0 + nv[2]};
        double df[3] = {p[0] - cl[0], p[1] - cl[1], p[2] - cl[2]};
        return norm3(df) > c->threshold;                     /* :741 (NaN compares false) */
    }
    const double *b1 = (e->idx == 0) ? c->spawn : &c->waypoints[3 * (e->idx - 1)];   /* :746-751 */
    const double *b2 = &c->waypoints[3 * e->idx];
    double lv[3] = {b2[0] - b1[0], b2[1] - b1[1], b2[2] - b1[2]};
    double ll = norm3(lv);
    if (ll == 0.0) {                                         /* :756-757 */
        double df[3] = {p[0] - b1[0], p[1] - b1[1], p[2] - b1[2]};
        return norm3(df) > c->threshold;
    }
    const double ext = 0.2;
    double u[3] = {lv[0] / ll, lv[1] / ll, lv[2] / ll};      /* :759 */
    double e1[3] = {b1[0] - ext * u[0], b1[1] - ext * u[1], b1[2] - ext * u[2]};   /* :773 */
    double e2[3] = {b2[0] + ext * u[0], b2[1] + ext * u[1], b2[2] + ext * u[2]};   /* :774 */
    double pd[3] = {p[0] - e1[0], p[1] - e1[1], p[2] - e1[2]};                      /* :776 */
    double proj = pd[0] * u[0] + pd[1] * u[1] + pd[2] * u[2];                      /* :778 */
    double ee[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
    proj = clipd(proj, 0.0, norm3(ee));                      /* :780 */
    double cl[3] = {e1[0] + proj * u[0], e1[1] + proj * u[1], e1[2] + proj * u[2]};   /* :782 */
    double df[3] = {p[0] - cl[0], p[1] - cl[1], p[2] - cl[2]};
    return norm3(df) > c->threshold + ext;                   /* :786 */
}

/* len(p.getContactPoints()) > 0 (PBDroneEnv.py:699): the only other body is plane.urdf at z=0
 * (BaseAviary.py:561).  APPROXIMATION [3P-recall]: lowest point of base_link's collision
 * cylinder (r=.06, h=.025, cf2x.urdf:34) within Bullet's 0.02 contact-breaking threshold of the
 * plane.  Unreachable in the BASELINE configs (the corridor test fires at z >= 0.3 first). */
static int ground_contact(const orc_env *e)
{
    double R[9];
    quat_to_mat(e->quat, R);
    double c = fabs(R[8]);
    double s2 = 1.0 - R[8] * R[8];
    double s = s2 > 0.0 ? sqrt(s2) : 0.0;
    double low = e->pos[2] - (0.5 * ORC_COLL_H * c + ORC_COLL_R * s);
    return low <= 0.02;
}

int32_t orc_has_collision(const orc_config *c, const orc_env *e)
{
    const double *s = e->pos;
    if (s[0] > c->dim[3] || s[0] < c->dim[0] || s[1] > c->dim[4] || s[1] < c->dim[1]) return 1;
    if (c->ground_contact && ground_contact(e)) return 1;
    if (s[2] > c->dim[5]) return 1;
    if (c->cylinder && out_of_cylinder(c, e)) return 1;
    return 0;
}

int32_t orc_compute_terminated(const orc_config *c, const orc_env *e)
{   /* PBDroneEnv.py:456-473; short-circuit keeps current_target() from being None */
    if (e->is_done) return 1;
    return orc_has_collision(c, e);
}

int32_t orc_compute_truncated(const orc_config *c, const orc_env *e)
{   /* PBDroneEnv.py:444-454 */
    return c->max_steps <= e->steps;
}

/* ------------------------------------------------------------------------- */
/* A7 -- _computeReward (PBDroneEnv.py:475-571), orientation_reward (:573-586),             */
/* get_forward_vector (:588-597), smoothness_reward (:599-607).                              */
/* ------------------------------------------------------------------------- */
static int orientation_reward(const orc_env *e, const double *target)
{
    const double thr = 10.0 * (ORC_PI / 180.0);              /* np.radians(10) */
    double cy = cos(e->rpy[2]), sy = sin(e->rpy[2]), cp = cos(e->rpy[1]), sp = sin(e->rpy[1]);
    double f[3] = {cy * cp, sy * cp, sp};
    double t[3] = {target[0] - e->pos[0], target[1] - e->pos[1], target[2] - e->pos[2]};
    double n = norm3(t);
    t[0] /= n; t[1] /= n; t[2] /= n;
    double dot = f[0] * t[0] + f[1] * t[1] + f[2] * t[2];
    double ang = acos(clipd(dot, -1.0, 1.0));
    return (ang > thr) ? -1 : 0;                             /* NaN -> 0 */
}

static double smoothness_reward(const orc_env *e)
{
    double dv[3] = {e->cur_vel[0] - e->prev_vel[0], e->cur_vel[1] - e->prev_vel[1], e->cur_vel[2] - e->prev_vel[2]};
    double dw[3] = {e->cur_ang_v[0] - e->prev_ang_v[0], e->cur_ang_v[1] - e->prev_ang_v[1], e->cur_ang_v[2] - e->prev_ang_v[2]};
    double la = norm3(dv), aa = norm3(dw);
    double lp = (la > 0.7) ? -fabs(la) : 0.0;
    double ap = (aa > 0.3) ? -fabs(aa) : 0.0;
    return lp + ap;
}

double orc_compute_reward(const orc_config *c, orc_env *e)
{
    if (orc_compute_terminated(c, e) && !e->is_done) return -10.0;          /* :489-490 */
    /* reward starts as np.float32(0.0) (:492): it stays float32 while only Python ints are
     * added (the two "found" branches) and widens to float64 once np.exp(...) is added. */
    if (e->d <= c->threshold) {                                              /* :539 */
        e->idx += 1;
        float r32 = 0.0f;
        if (e->idx == c->num_waypoints) {                                    /* :542-546 */
            r32 = r32 + 200.0f;
            e->is_done = 1;
        } else {                                                             /* :548-552 */
            r32 = r32 + 75.0f;
            r32 = r32 + (float)(orientation_reward(e, &c->waypoints[3 * e->idx]) * 5);
            e->just_found = 1;
        }
        e->d_prev = e->d;                                                    /* :568 */
        return (double)(r32 / 25.0f);                                        /* :571 */
    }
    double r = 0.0;
    r = r + exp(-2.0 * e->d) * 3.0;                                          /* :555 */
    r = r + (e->just_found ? 0.0 : (e->d_prev - e->d) * 3000.0);             /* :556 */
    r = r + (double)(orientation_reward(e, &c->waypoints[3 * e->idx]) * 3);  /* :557 */
    r = r + smoothness_reward(e);                                            /* :558 */
    e->just_found = 0;                                                       /* :566 */
    e->d_prev = e->d;
    return r / 25.0;
}

/* ------------------------------------------------------------------------- */
/* A9 -- _update_state_post_step (PBDroneEnv.py:201-223), reset (:609-665),                 */
/* BaseAviary.reset (BaseAviary.py:276-320), constructor bookkeeping (PBDroneEnv.py:122-145). */
/* ------------------------------------------------------------------------- */
void orc_post_step(const orc_config *c, orc_env *e)
{
    e->steps += 1;
    memcpy(e->cur_pos, e->pos, sizeof e->pos);
    memcpy(e->prev_vel, e->cur_vel, sizeof e->vel);
    memcpy(e->prev_ang_v, e->cur_ang_v, sizeof e->vel);
    memcpy(e->cur_vel, e->vel, sizeof e->vel);
    memcpy(e->cur_ang_v, e->ang_v, sizeof e->vel);
    const double *t = &c->waypoints[3 * e->idx];
    double df[3] = {t[0] - e->cur_pos[0], t[1] - e->cur_pos[1], t[2] - e->cur_pos[2]};
    e->d = norm3(df);
}

static void bullet_reset(const orc_config *c, orc_env *e)
{   /* p.resetSimulation + _housekeeping: body reloaded at INIT_XYZS / INIT_RPYS = 0, at rest */
    memcpy(e->pos, c->spawn, sizeof e->pos);
    if (c->random_spawn && e->spawn_ready) memcpy(e->pos, e->spawn_pt, sizeof e->pos);   /* INIT_XYZS[0] of this episode */
    e->quat[0] = e->quat[1] = e->quat[2] = 0.0; e->quat[3] = 1.0;
    memset(e->vel, 0, sizeof e->vel);
    memset(e->ang_v, 0, sizeof e->ang_v);
    orc_euler_from_quat(e->quat, e->rpy);
    memset(e->last_clipped_action, 0, sizeof e->last_clipped_action);        /* _housekeeping, BaseAviary.py:545 */
}

void orc_env_construct(const orc_config *c, orc_env *e)
{
    memset(e, 0, sizeof *e);
    bullet_reset(c, e);
    memcpy(e->cur_pos, c->spawn, sizeof e->pos);                             /* :122 */
    double df[3] = {e->cur_pos[0] - c->waypoints[0], e->cur_pos[1] - c->waypoints[1], e->cur_pos[2] - c->waypoints[2]};
    e->d = e->d_prev = norm3(df);                                            /* :137-138 */
    for (int i = 0; i < ORC_OBS_DIM; ++i) { e->rms_mean[i] = 0.0; e->rms_var[i] = 1.0; }
    e->rms_count = 1e-4;                                                     /* normalize.py:14-18 */
}

void orc_env_reset(const orc_config *c, orc_env *e, float obs[ORC_OBS_DIM])
{
    if (c->random_spawn) {           /* the dormant block of PBDroneEnv.reset (:622-627), built as what it evidently intends: the
                                      * episode's spawn point is drawn, the body is loaded there and _current_position follows */
        orc_random_spawn(c, e->gid, e->step_count, e->spawn_pt);
        e->spawn_ready = 1;
    }
    bullet_reset(c, e);
    if (c->random_spawn) memcpy(e->cur_pos, e->spawn_pt, sizeof e->cur_pos);
    orc_compute_obs(c, e, obs);              /* BaseAviary.py:318 -- BEFORE the bookkeeping below (Q2) */
    e->is_done = 0; e->idx = 0; e->steps = 0;                                /* :617-619 */
    double df[3] = {e->cur_pos[0] - c->waypoints[0], e->cur_pos[1] - c->waypoints[1], e->cur_pos[2] - c->waypoints[2]};
    e->d = norm3(df);                        /* :651, _current_position is NOT reset (Q3) */
    e->d_prev = e->d;                        /* :652 */
    memset(e->prev_vel, 0, sizeof e->vel); memset(e->prev_ang_v, 0, sizeof e->vel);   /* :653-654 */
    memset(e->cur_vel, 0, sizeof e->vel); memset(e->cur_ang_v, 0, sizeof e->vel);     /* :657 */
    e->just_found = 0;                                                       /* :658 */
}

/* PBDroneEnv.step (PBDroneEnv.py:171-199) around BaseAviary.step (BaseAviary.py:324-453). */
void orc_env_step(const orc_config *c, orc_env *e, const float action[4], orc_step_out *out)
{
    float cmd[4], rpm32[4], f32[4], zt32;
    double rpm[4], f[4], zt, body_force[3] = {0.0, 0.0, 0.0};
    if (c->normalize_actions) orc_rescale_action(action, cmd);               /* :173-176 */
    else memcpy(cmd, action, sizeof cmd);
    const int rpm_is_f32 = c->action_type == 0;
    if (c->action_type == 1) orc_rpm_action(cmd, rpm, f, &zt);               /* ActionType.RPM (N4) */
    else if (c->action_type >= 2) {                                          /* PID / VEL / ONE_D_RPM / ONE_D_PID (N4) */
        double tq[4];
        orc_pid_control(c->action_type, e->pos, e->quat, e->vel, cmd, e->pid, rpm);
        for (int i = 0; i < 4; ++i) {                                        /* BaseAviary._physics on a float64 rpm array, :776-780 */
            double sq = rpm[i] * rpm[i];
            f[i] = sq * ORC_KF;
            tq[i] = sq * ORC_KM;
        }
        zt = -tq[0];
        zt = zt + tq[1];
        zt = zt - tq[2];
        zt = zt + tq[3];
    } else {
        orc_preprocess_action(cmd, rpm32);                                   /* BaseAviary.py:408 */
        orc_rotor_forces(rpm32, f32, &zt32);                                 /* :420-421 */
        for (int i = 0; i < 4; ++i) { rpm[i] = rpm32[i]; f[i] = f32[i]; }
        zt = (double)zt32;
    }
    if (c->physics == 1 || c->physics == 4) {                                /* _groundEffect, :422-424,431-434 */
        double g[4];
        orc_ground_effect(e->pos, e->quat, e->rpy, rpm, rpm_is_f32, g);
        for (int i = 0; i < 4; ++i) f[i] += g[i];                            /* same link, forces add up in Bullet */
    }
    if (c->physics == 2 || c->physics == 4)                                  /* _drag(last_clipped_action), :425-427,435 */
        orc_drag(e->quat, e->vel, e->last_clipped_action, rpm_is_f32, body_force);
    orc_bullet_step_damp(e->pos, e->quat, e->vel, e->ang_v, f, zt, body_force, c->zero_damping ? 0.0 : ORC_LIN_DAMP);   /* :439-440 */
    memcpy(e->last_clipped_action, rpm, sizeof rpm);                         /* :442 */
    orc_euler_from_quat(e->quat, e->rpy);                                    /* :444 */
    orc_compute_obs(c, e, out->obs);                                         /* :446 */
    out->reward = orc_compute_reward(c, e);                                  /* :447 */
    out->terminated = orc_compute_terminated(c, e);                          /* :448 */
    out->truncated = orc_compute_truncated(c, e);                            /* :449 */
    out->found_targets = e->idx;                                             /* :450, PBDroneEnv.py:442 */
    if (!out->terminated) orc_post_step(c, e);                               /* PBDroneEnv.py:196-197 */
}

/* A10 -- normalize.NormalizeObservation.normalize (normalize.py:94-97) with a batch of one:
 * RunningMeanStd.update -> update_mean_var_count_from_moments (normalize.py:34-47). */
void orc_normalize_obs(orc_env *e, const float obs_in[ORC_OBS_DIM], double obs_out[ORC_OBS_DIM])
{
    double count = e->rms_count;
    double tot = count + 1.0;
    for (int i = 0; i < ORC_OBS_DIM; ++i) {
        double x = (double)obs_in[i];
        double delta = x - e->rms_mean[i];
        double new_mean = e->rms_mean[i] + delta * 1.0 / tot;
        double m_a = e->rms_var[i] * count;
        double m_b = 0.0 * 1.0;
        double M2 = m_a + m_b + delta * delta * count * 1.0 / tot;
        e->rms_mean[i] = new_mean;
        e->rms_var[i] = M2 / tot;
        obs_out[i] = (x - e->rms_mean[i]) / sqrt(e->rms_var[i] + 1e-8);
    }
    e->rms_count = tot;
}

/* ------------------------------------------------------------------------- */
/* Noise (BASELINE config 5).  The reference has no noise code; sigma = 0 is the reference.   */
/* Philox4x32-10, key = seed, counter = (env id lo, env id hi, step lo, stream | step hi << 8); */
/* Box-Muller in                                                                               */
/* float64, rounded to float32.                                                                */
/* ------------------------------------------------------------------------- */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_noise4(uint64_t seed, uint64_t env_id, uint64_t step, uint32_t stream, float out[4])
{
    uint32_t r[4];
    /* counter = (env id lo, env id hi, step lo, stream | step hi << 8): the whole 64-bit vector-step counter, streams < 256 */
    orc_philox4x32((uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)step, stream | ((uint32_t)(step >> 32) << 8),
                   (uint32_t)seed, (uint32_t)(seed >> 32), r);
    for (int h = 0; h < 2; ++h) {
        double u1 = ((double)r[2 * h] + 0.5) * (1.0 / 4294967296.0);
        double u2 = ((double)r[2 * h + 1] + 0.5) * (1.0 / 4294967296.0);
        double rad = sqrt(-2.0 * log(u1));
        double ang = 2.0 * ORC_PI * u2;
        out[2 * h] = (float)(rad * cos(ang));
        out[2 * h + 1] = (float)(rad * sin(ang));
    }
}

/* The same for n consecutive drones (tests: the device draws against this definition over millions of draws). */
void orc_noise4_many(uint64_t seed, uint64_t env_id0, int64_t n, uint64_t step, uint32_t stream, float *out)
{
    for (int64_t i = 0; i < n; ++i) orc_noise4(seed, env_id0 + (uint64_t)i, step, stream, out + 4 * i);
}

static void add_obs_noise(const orc_config *c, uint64_t env_id, uint64_t step, uint32_t stream0, float obs[ORC_OBS_DIM])
{
    if (!(c->obs_noise_sigma > 0.0f)) return;
    for (int b = 0; b < 4; ++b) {
        float z[4];
        orc_noise4(c->seed, env_id, step, stream0 + (uint32_t)b, z);
        for (int j = 0; j < 4 && 4 * b + j < ORC_OBS_DIM; ++j) {
            float s = c->obs_noise_sigma * z[j];
            obs[4 * b + j] = obs[4 * b + j] + s;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* A11 -- SB3 SubprocVecEnv worker + Monitor [3P-recall; not in the tree], as used at        */
/* PBDroneSimulator.py:196,653-666:                                                          */
/*   obs, r, terminated, truncated, info = env.step(a); done = terminated or truncated       */
/*   info["TimeLimit.truncated"] = truncated and not terminated                              */
/*   if done: info["terminal_observation"] = obs; obs, _ = env.reset()                       */
/*   Monitor: info["episode"] = {r: sum(rewards), l: len(rewards)} when done                 */
/* Wrapper order (PBDroneSimulator.py:181-196): Monitor(NormalizeObservation(PBDroneEnv)).   */
/* ------------------------------------------------------------------------- */
static void round_state_f32(orc_env *e)
{
#define RF(x) (x) = (double)(float)(x)
    for (int i = 0; i < 3; ++i) { RF(e->pos[i]); RF(e->vel[i]); RF(e->ang_v[i]); RF(e->cur_pos[i]);
                                  RF(e->cur_vel[i]); RF(e->cur_ang_v[i]); RF(e->prev_vel[i]); RF(e->prev_ang_v[i]); }
    for (int i = 0; i < 4; ++i) { RF(e->quat[i]); RF(e->last_clipped_action[i]); }
    RF(e->d); RF(e->d_prev);                    /* not ep_ret: Monitor's sum is a Python float, and the device carries it as a float32 pair */
#undef RF
}

/* make_env's optional reward wrappers (PBDroneSimulator.py:191-194), applied inside Monitor:                  */
/*   TransformReward(lambda r: np.clip(r, -10, 10))                        if --clip_rew                       */
/*   NormalizeReward(gamma=0.99, epsilon=1e-8), normalize.py:132-147       if --norm_rew                       */
/*     returns = returns*gamma + rew; return_rms.update(returns) (batch of one: batch_var = 0);                */
/*     rew = rew / sqrt(return_rms.var + epsilon); returns[dones] = 0                                          */
double orc_reward_wrappers(const orc_config *c, orc_env *e, double reward, int32_t done)
{
    double r = reward;
    if (c->clip_rew) r = r < -10.0 ? -10.0 : (r > 10.0 ? 10.0 : r);
    if (c->norm_rew) {
        e->rr_returns = e->rr_returns * 0.99 + r;                         /* :134 */
        /* update_mean_var_count_from_moments, normalize.py:34-47, batch_mean = returns, batch_var = 0, batch_count = 1 */
        double delta = e->rr_returns - e->rr_mean;
        double tot = e->rr_count + 1.0;
        double new_mean = e->rr_mean + delta * 1.0 / tot;
        double m_a = e->rr_var * e->rr_count;
        double m_b = 0.0 * 1.0;
        double M2 = m_a + m_b + delta * delta * e->rr_count * 1.0 / tot;
        e->rr_mean = new_mean;
        e->rr_var = M2 / tot;
        e->rr_count = tot;
        r = r / sqrt(e->rr_var + 1e-8);                                   /* :147 */
        if (done) e->rr_returns = 0.0;                                    /* :138 */
    }
    return r;
}

void orc_vec_create(const orc_config *c, orc_env *envs, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float obs[ORC_OBS_DIM];
        orc_env_construct(c, &envs[i]);
        envs[i].gid = (uint64_t)(c->env_id_offset + i);
        envs[i].rr_returns = 0.0; envs[i].rr_mean = 0.0; envs[i].rr_var = 1.0; envs[i].rr_count = 1e-4;   /* normalize.py:14-18, :127-128 */
        orc_env_reset(c, &envs[i], obs);         /* make_env: env.reset(seed=seed+rank) before wrapping, PBDroneSimulator.py:173 */
        if (c->f32_state) round_state_f32(&envs[i]);
    }
}

static void finish_obs(const orc_config *c, orc_env *e, uint64_t env_id, uint32_t stream0, float obs[ORC_OBS_DIM])
{
    add_obs_noise(c, env_id, e->step_count, stream0, obs);
    if (c->normalize_obs) {
        double o[ORC_OBS_DIM];
        orc_normalize_obs(e, obs, o);
        for (int i = 0; i < ORC_OBS_DIM; ++i) obs[i] = (float)o[i];
    }
}

void orc_vec_reset(const orc_config *c, orc_env *envs, int64_t n, float *obs, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        orc_env *e = &envs[i];
        orc_env_reset(c, e, &obs[i * ORC_OBS_DIM]);
        finish_obs(c, e, (uint64_t)(c->env_id_offset + i), 5u, &obs[i * ORC_OBS_DIM]);
        e->ep_ret = 0.0; e->ep_len = 0;          /* Monitor.reset */
        if (c->f32_state) round_state_f32(e);
    }
}

void orc_vec_step(const orc_config *c, orc_env *envs, int64_t n, const float *actions,
                  float *obs, float *reward, uint8_t *done, uint8_t *truncated, int32_t *found_targets,
                  float *terminal_obs, float *ep_ret, int32_t *ep_len, uint8_t *terminated, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        orc_env *e = &envs[i];
        uint64_t gid = (uint64_t)(c->env_id_offset + i);
        float a[4] = {actions[4 * i], actions[4 * i + 1], actions[4 * i + 2], actions[4 * i + 3]};
        if (c->act_noise_sigma > 0.0f) {
            float z[4];
            orc_noise4(c->seed, gid, e->step_count, 0u, z);
            for (int j = 0; j < 4; ++j) {
                float s = c->act_noise_sigma * z[j];
                a[j] = clipf(a[j] + s, -1.0f, 1.0f);
            }
        }
        orc_step_out so;
        orc_env_step(c, e, a, &so);
        float *o = &obs[i * ORC_OBS_DIM];
        memcpy(o, so.obs, sizeof so.obs);
        finish_obs(c, e, gid, 1u, o);
        int dn = so.terminated || so.truncated;
        so.reward = orc_reward_wrappers(c, e, so.reward, dn);
        e->ep_ret += so.reward;                   /* Monitor.step */
        e->ep_len += 1;
        reward[i] = (float)so.reward;
        done[i] = (uint8_t)dn;
        truncated[i] = (uint8_t)(so.truncated && !so.terminated);
        found_targets[i] = so.found_targets;
        if (terminated) terminated[i] = (uint8_t)so.terminated;
        if (dn) {
            if (terminal_obs) memcpy(&terminal_obs[i * ORC_OBS_DIM], o, sizeof so.obs);
            if (ep_ret) ep_ret[i] = (float)e->ep_ret;
            if (ep_len) ep_len[i] = e->ep_len;
            orc_env_reset(c, e, o);
            finish_obs(c, e, gid, 5u, o);
            e->ep_ret = 0.0; e->ep_len = 0;
        }
        e->step_count += 1;
        if (c->f32_state) round_state_f32(e);
    }
}

/* ------------------------------------------------------------------------- */
/* N1 -- GAE, Sol/Model/Algorithms/cleanRLPPO.py:234-248 (float32 tensors):                  */
/*   nextnonterminal = 1 - dones[t+1]; delta = r[t] + gamma*nextvalues*nextnonterminal - v[t] */
/*   adv[t] = lastgaelam = delta + gamma*lambda*nextnonterminal*lastgaelam; returns = adv + v */
/* Layout [n_steps, n_envs]; dones[t] is the done flag going INTO step t (cleanRL convention). */
/* ------------------------------------------------------------------------- */
void orc_gae(const float *rewards, const float *values, const uint8_t *dones,
             const float *last_values, const uint8_t *last_dones,
             int64_t T, int64_t N, double gamma_, double lam_, float *adv, float *ret)
{
    /* gamma and gamma*gae_lambda are Python floats multiplied into float32 tensors */
    const float gamma = (float)gamma_;
    const float gl = (float)(gamma_ * lam_);
    for (int64_t i = 0; i < N; ++i) {
        float last = 0.0f;
        for (int64_t t = T - 1; t >= 0; --t) {
            float nnt, nv;
            if (t == T - 1) { nnt = 1.0f - (float)last_dones[i]; nv = last_values[i]; }
            else { nnt = 1.0f - (float)dones[(t + 1) * N + i]; nv = values[(t + 1) * N + i]; }
            float gv = gamma * nv;
            float delta = rewards[t * N + i] + gv * nnt;
            delta = delta - values[t * N + i];
            float k = gl * nnt;
            last = delta + k * last;
            adv[t * N + i] = last;
            ret[t * N + i] = last + values[t * N + i];
        }
    }
}

/* test helper (teacher forcing): _updateAndStoreKinematicInformation's rpy cache from the (overwritten) quaternion */
void orc_vec_refresh_rpy(orc_env *envs, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) orc_euler_from_quat(envs[i].quat, envs[i].rpy);
}

int32_t orc_sizeof_env(void) { return (int32_t)sizeof(orc_env); }
int32_t orc_sizeof_config(void) { return (int32_t)sizeof(orc_config); }
int32_t orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
