// dn_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the drone-navigation environment step.
//
// One drone per lane, one 64-lane wave per workgroup.  The whole reference step
//   PBDroneEnv.step -> BaseAviary.step -> p.stepSimulation -> obs / reward / done -> post-step
//   -> SubprocVecEnv auto-reset + Monitor (+ optional per-drone NormalizeObservation)
// is one kernel: the float32 state is read once as six float4 groups (16 B per lane, coalesced),
// advanced in registers, and written back once; the waypoint/corridor table sits in LDS; the [64,13]
// observation tile is transposed through LDS so it leaves as full-width float4 stores in the
// row-major [N,13] layout the SB3 VecEnv boundary needs.  No MFMA: this is element-wise work bounded
// by HBM traffic (288 algorithmic bytes per drone step).
//
// Arithmetic: the action chain (rescale -> thrust -> PWM -> RPM -> forces) is float32 exactly as the
// reference's numpy float32 arrays; everything after it is computed in `R` = double (parity grade: the
// reference is float64 throughout, Bullet included) or float (fast mode).  The file is compiled with
// -ffp-contract=off so that no multiply-add is fused: numpy and Bullet round every operation.
//
// Reference citations are file:line under /root/reference:
//   PBDroneEnv.py = Sol/Model/Environments/PBDroneEnv.py, BaseAviary.py = Sol/PyBullet/BaseAviary.py,
//   env_utils.py = Sol/Model/env_utils.py, normalize.py = Sol/Model/Environments/normalize.py.
#include "dn_internal.h"

#include <float.h>

#define DN_DEV __device__ __forceinline__

namespace {

// ---- constants: Sol/resources/safegym/cf2x.urdf:5,11-12 via BaseAviary._parse_urdf_parameters
//      (BaseAviary.py:1123-1163); prop offsets from the URDF Bullet actually loads
//      (Sol/resources/cf2x.urdf:42,54,66,78, BaseAviary.py:562-570); G at BaseAviary.py:76.
template <typename R> struct K {
    static constexpr R M = R(0.027);
    static constexpr R IXX = R(1.4e-5), IYY = R(1.4e-5), IZZ = R(2.17e-5);
    static constexpr R G = R(9.8);
    static constexpr R DT = R(1.0 / 240.0);
    static constexpr R ARM = R(0.028);
    static constexpr R LIN_DAMP = R(0.04), ANG_DAMP = R(0.04);   // btMultiBody defaults [3P-recall]
    static constexpr R MAX_COORD_VEL = R(100.0);                 // btMultiBody::m_maxCoordinateVelocity
    static constexpr R PI = R(3.14159265358979323846);
    static constexpr R COLL_R = R(0.06), COLL_H = R(0.025);      // cf2x.urdf:34
    // Reciprocals of the constant divisors: x * (1/c) instead of x / c.  A float64 divide is a ~15-instruction
    // dependent chain on gfx950; the product differs from the quotient by at most one float64 ulp (1e-16),
    // eleven orders of magnitude inside the 1e-5 parity bar and invisible after the float32 store.
    static constexpr R INV_M = R(1.0 / 0.027);
    static constexpr R INV_IXX = R(1.0 / 1.4e-5), INV_IYY = R(1.0 / 1.4e-5), INV_IZZ = R(1.0 / 2.17e-5);
    static constexpr R INV_PI = R(1.0 / 3.14159265358979323846);
    static constexpr R THIRD = R(1.0 / 3.0);
    static constexpr R INV_25 = R(1.0 / 25.0);
    static constexpr R COS_10DEG = R(0.98480775301220805936674302458952);   // cos(np.radians(10))
};
constexpr float KF32 = (float)3.16e-10, KM32 = (float)7.94e-12;
constexpr float PWM2RPM_SCALE32 = (float)0.2685, PWM2RPM_CONST32 = (float)4070.3;
constexpr float MIN_PWM32 = 20000.0f, MAX_PWM32 = 65535.0f;
// a_low / a_high = float32(KF * (SCALE * PWM + CONST)**2), PBDroneEnv.py:113-116
constexpr float A_LOW32 = (float)(3.16e-10 * ((0.2685 * 20000.0 + 4070.3) * (0.2685 * 20000.0 + 4070.3)));
constexpr float A_HIGH32 = (float)(3.16e-10 * ((0.2685 * 65535.0 + 4070.3) * (0.2685 * 65535.0 + 4070.3)));

template <typename T> DN_DEV T clipv(T x, T lo, T hi)
{   // np.clip = minimum(maximum(x, lo), hi); NaN propagates (both compares false)
    return x < lo ? lo : (x > hi ? hi : x);
}
template <typename R> DN_DEV R norm3(R a, R b, R c) { return sqrt(a * a + b * b + c * c); }

// ---- A1-A3: float32 action chain ------------------------------------------------------------------
DN_DEV float rotor_force_from_action(float a, bool normalize_actions, float &torque)
{
    float cmd = a;
    if (normalize_actions) {                     // PBDroneEnv.rescale_action, PBDroneEnv.py:949-971
        float num = a - A_LOW32;
        float den = A_HIGH32 - A_LOW32;
        float q = num / den;
        float m = 2.0f * q;                      // (high - low) = 1 - (-1)
        float r = -1.0f + m;
        cmd = clipv(r, -1.0f, 1.0f);
    }
    float thrust = clipv(cmd, A_LOW32, A_HIGH32);    // PBDroneEnv._preprocessAction, PBDroneEnv.py:889
    if (thrust < 0.0f) thrust = 0.0f;                // cmd2pwm, env_utils.py:29
    float t = thrust / KF32;                         // env_utils.py:30 (n_motor = 1)
    float s = sqrtf(t);
    float pwm = (s - PWM2RPM_CONST32) / PWM2RPM_SCALE32;
    pwm = clipv(pwm, MIN_PWM32, MAX_PWM32);          // env_utils.py:39
    float r0 = PWM2RPM_SCALE32 * pwm;
    float rpm = r0 + PWM2RPM_CONST32;                // pwm2rpm, env_utils.py:58
    float sq = rpm * rpm;                            // BaseAviary._physics, BaseAviary.py:776-777
    torque = sq * KM32;
    return sq * KF32;
}

// ---- noise (BASELINE config 5; sigma = 0 is the reference): Philox4x32-10 + float64 Box-Muller ------
DN_DEV void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
DN_DEV void noise4(unsigned long long seed, unsigned long long gid, unsigned step, unsigned stream, float z[4])
{
    unsigned r[4];
    philox4x32((unsigned)gid, (unsigned)(gid >> 32), step, stream, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double u1 = ((double)r[2 * h] + 0.5) * (1.0 / 4294967296.0);
        double u2 = ((double)r[2 * h + 1] + 0.5) * (1.0 / 4294967296.0);
        double rad = sqrt(-2.0 * log(u1));
        double ang = 2.0 * 3.14159265358979323846 * u2;
        z[2 * h] = (float)(rad * cos(ang));
        z[2 * h + 1] = (float)(rad * sin(ang));
    }
}
DN_DEV void add_obs_noise(const DnParams &p, unsigned long long gid, unsigned step, unsigned stream0, float o[DN_OBS_DIM])
{
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float z[4];
        noise4(p.seed, gid, step, stream0 + (unsigned)b, z);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < DN_OBS_DIM) {
                float s = p.obs_noise_sigma * z[j];
                o[4 * b + j] = o[4 * b + j] + s;
            }
    }
}

// ---- A10: normalize.NormalizeObservation with a batch of one (normalize.py:34-47, :94-97) ---------
DN_DEV void normalize_obs(const DnParams &p, long long i, bool active, double &count, float o[DN_OBS_DIM])
{
    double tot = count + 1.0;
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) {
        double mean = p.st.rms_mean[(long long)k * p.n + i];
        double var = p.st.rms_var[(long long)k * p.n + i];
        double x = (double)o[k];
        double delta = x - mean;
        double new_mean = mean + delta * 1.0 / tot;
        double m_a = var * count;
        double M2 = m_a + 0.0 + delta * delta * count * 1.0 / tot;
        double new_var = M2 / tot;
        if (active) {
            p.st.rms_mean[(long long)k * p.n + i] = new_mean;
            p.st.rms_var[(long long)k * p.n + i] = new_var;
        }
        o[k] = (float)((x - new_mean) / sqrt(new_var + 1e-8));
    }
    count = tot;
}

// ---- waypoint/corridor table in LDS ------------------------------------------------------------
template <typename R> DN_DEV const R *table_ptr(const DnParams &p);
template <> DN_DEV const double *table_ptr<double>(const DnParams &p) { return p.tab64; }
template <> DN_DEV const float *table_ptr<float>(const DnParams &p) { return p.tab32; }
template <typename R> DN_DEV const DnConsts<R> &consts(const DnParams &p);
template <> DN_DEV const DnConsts<double> &consts<double>(const DnParams &p) { return p.c64; }
template <> DN_DEV const DnConsts<float> &consts<float>(const DnParams &p) { return p.c32; }

// ---- A8: _has_collision_occurred (PBDroneEnv.py:678-707) + is_out_of_cylinder_bounds (:718-786) ----
template <typename R>
DN_DEV bool has_collision(const DnParams &p, const DnConsts<R> &c, const R *tab, R px, R py, R pz, R r22, int idx)
{
    if (px > c.dim[3] || px < c.dim[0] || py > c.dim[4] || py < c.dim[1]) return true;
    if (p.ground_contact) {
        // len(p.getContactPoints()) > 0 against plane.urdf, APPROXIMATED [3P-recall]: lowest point of the
        // collision cylinder within Bullet's 0.02 contact-breaking threshold of z = 0.
        R cz = fabs(r22);
        R s2 = R(1.0) - r22 * r22;
        R s = s2 > R(0.0) ? sqrt(s2) : R(0.0);
        R low = pz - (R(0.5) * K<R>::COLL_H * cz + K<R>::COLL_R * s);
        if (low <= R(0.02)) return true;
    }
    if (pz > c.dim[5]) return true;
    if (!p.cylinder) return false;
    if (p.circle) {                                   // :723-741, centre (0,0,1), radius 1
        R cx = px - R(0.0), cy = py - R(0.0), cz = R(0.0);
        R rn = R(1.0) / norm3(cx, cy, cz);
        R nx = cx * rn, ny = cy * rn, nz = cz * rn;    // radius 1; 0 * inf -> NaN -> the compare below is false
        R qx = R(0.0) + nx, qy = R(0.0) + ny, qz = R(1.0) + nz;
        return norm3(px - qx, py - qy, pz - qz) > c.threshold;
    }
    const R *e = tab + idx * DN_T_STRIDE;
    if (e[DN_T_LL] == R(0.0))                          // :756-757
        return norm3(px - e[DN_T_B1], py - e[DN_T_B1 + 1], pz - e[DN_T_B1 + 2]) > c.threshold;
    R ux = e[DN_T_U], uy = e[DN_T_U + 1], uz = e[DN_T_U + 2];
    R ex = e[DN_T_E1], ey = e[DN_T_E1 + 1], ez = e[DN_T_E1 + 2];
    R dx = px - ex, dy = py - ey, dz = pz - ez;        // :776
    R proj = dx * ux + dy * uy + dz * uz;              // :778
    proj = clipv(proj, R(0.0), e[DN_T_LEXT]);          // :780
    R qx = ex + proj * ux, qy = ey + proj * uy, qz = ez + proj * uz;   // :782
    return norm3(px - qx, py - qy, pz - qz) > c.thr_ext;               // :786
}

// orientation_reward (PBDroneEnv.py:573-586) with get_forward_vector (:588-597).  The reference tests
// arccos(clip(f . t, -1, 1)) > radians(10); arccos is strictly decreasing, so that is f . t < cos(10 deg)
// (NaN compares false on both forms) and the arccos is never evaluated.
template <typename R>
DN_DEV int orientation_reward(R fx, R fy, R fz, R px, R py, R pz, const R *wp)
{
    R tx = wp[0] - px, ty = wp[1] - py, tz = wp[2] - pz;
    R rn = R(1.0) / norm3(tx, ty, tz);
    R dot = fx * (tx * rn) + fy * (ty * rn) + fz * (tz * rn);
    return (clipv(dot, R(-1.0), R(1.0)) < K<R>::COS_10DEG) ? -1 : 0;
}

// sin(h)/h and cos(h) for the quaternion half-angle h = |w| dt / 2 <= pi/8 (Bullet clamps |w| dt at pi/4):
// Taylor polynomials in h^2, truncation < 1e-18 on that interval, no range reduction needed.
template <typename R> DN_DEV void sinc_cos_small(R h2, R &sinc, R &c)
{
    sinc = R(1.0) + h2 * (R(-1.0 / 6.0) + h2 * (R(1.0 / 120.0) + h2 * (R(-1.0 / 5040.0) + h2 * (R(1.0 / 362880.0) +
           h2 * (R(-1.0 / 39916800.0) + h2 * (R(1.0 / 6227020800.0) + h2 * R(-1.0 / 1307674368000.0)))))));
    c = R(1.0) + h2 * (R(-0.5) + h2 * (R(1.0 / 24.0) + h2 * (R(-1.0 / 720.0) + h2 * (R(1.0 / 40320.0) +
        h2 * (R(-1.0 / 3628800.0) + h2 * (R(1.0 / 479001600.0) + h2 * (R(-1.0 / 87178291200.0) +
        h2 * R(1.0 / 20922789888000.0))))))));
}

struct Meta {
    int steps, idx, just_found;
};
DN_DEV Meta unpack_meta(float f)
{
    unsigned u = __float_as_uint(f);
    Meta m;
    m.steps = (int)(u & 0xFFFFFFu);
    m.idx = (int)((u >> 24) & 0x7Fu);
    m.just_found = (int)(u >> 31);
    return m;
}
DN_DEV float pack_meta(int steps, int idx, int just_found)
{
    return __uint_as_float(((unsigned)steps & 0xFFFFFFu) | (((unsigned)idx & 0x7Fu) << 24) | ((unsigned)just_found << 31));
}

// Stores the wave's [64,13] observation tile: lanes park their 13 floats in LDS (stride 13 dwords: odd, so
// conflict-free), then the wave streams the 3328 contiguous bytes out as float4 (ds_read_b128 +
// global_store_dwordx4), i.e. 4 store instructions instead of 13 strided dword stores per destination.
DN_DEV void store_obs_tile(float *s_tile, float *gdst, long long tile_base, long long n, int lane, bool active,
                           const float o[DN_OBS_DIM])
{
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) s_tile[lane * DN_OBS_DIM + k] = o[k];
    __syncthreads();
    (void)active;
    if (tile_base + DN_BLOCK <= n) {
        float4 *g4 = reinterpret_cast<float4 *>(gdst + tile_base * DN_OBS_DIM);
        const float4 *s4 = reinterpret_cast<const float4 *>(s_tile);
#pragma unroll
        for (int r = 0; r < 3; ++r) g4[r * 64 + lane] = s4[r * 64 + lane];
        if (lane < (DN_BLOCK * DN_OBS_DIM / 4 - 192)) g4[192 + lane] = s4[192 + lane];
    } else {
        long long rem = (n - tile_base) * DN_OBS_DIM;      // ragged last tile
        for (int e = lane; e < rem; e += DN_BLOCK) gdst[tile_base * DN_OBS_DIM + e] = s_tile[e];
    }
}

template <typename R>
DN_DEV void stage_table(const DnParams &p, R *s_tab)
{
    const R *g = table_ptr<R>(p);
    for (int j = threadIdx.x; j < p.num_waypoints * DN_T_STRIDE; j += DN_BLOCK) s_tab[j] = g[j];
    __syncthreads();
}

// Observation of a body that has just been (re)loaded at the spawn pose (BaseAviary.reset -> _computeObs,
// BaseAviary.py:318) with the not-yet-reset stale distance d_last (quirk Q2).
template <typename R>
DN_DEV void reset_obs(const DnParams &p, const DnConsts<R> &c, R d_last, float o[DN_OBS_DIM])
{
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = (float)c.reset_obs[k];
    o[12] = p.include_distance ? (float)clipv(d_last * c.inv_max_target_dist, -(R)FLT_MAX, (R)FLT_MAX) : 0.0f;
}

// =====================================================================================================
// The step kernel.
// =====================================================================================================
// NORM / NOISE compile the optional per-drone observation normaliser and the Philox noise streams in or out:
// the reference-default kernels <R, false, false> carry neither their registers nor their code.
//
// step_body advances ONE drone (this lane) by one control step.  The persistent state travels as the six
// float4 groups G0..G5 exactly as they sit in HBM (float32), so the single-step kernel (load, step, store)
// and the fused multi-step kernel (load, K x step, store) run the same arithmetic on the same roundings.
template <typename R, bool NORM, bool NOISE>
DN_DEV void step_body(const DnParams &p, const DnConsts<R> &c, const R *s_tab, float *s_tile, const DnStepIO &io,
                      const unsigned step_count, const long long i, const long long tile_base, const int lane,
                      const bool active, const float4 A, float4 &G0, float4 &G1, float4 &G2, float4 &G3, float4 &G4,
                      float4 &G5)
{
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    float a[4] = {A.x, A.y, A.z, A.w};
    if (NOISE && p.act_noise_sigma > 0.0f) {
        float z[4];
        noise4(p.seed, gid, step_count, 0u, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = p.act_noise_sigma * z[j];
            a[j] = clipv(a[j] + s, -1.0f, 1.0f);
        }
    }

    // ---- A1-A3 (float32) ---------------------------------------------------------------------------
    float tq[4], f32[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f32[j] = rotor_force_from_action(a[j], p.normalize_actions != 0, tq[j]);
    float zt32 = -tq[0];
    zt32 = zt32 + tq[1];
    zt32 = zt32 - tq[2];
    zt32 = zt32 + tq[3];                               // BaseAviary.py:780

    // ---- unpack the entry state -----------------------------------------------------------------------
    R px = G0.x, py = G0.y, pz = G0.z;
    const R d_e = G0.w;
    R qx = G1.x, qy = G1.y, qz = G1.z, qw = G1.w;
    R vx = G2.x, vy = G2.y, vz = G2.z;
    const R dprev_e = G2.w;
    R wx = G3.x, wy = G3.y, wz = G3.z;
    const Meta m_e = unpack_meta(G3.w);
    const R pvx = G4.x, pvy = G4.y, pvz = G4.z;
    const R epret_e = G4.w;
    const R pwx = G5.x, pwy = G5.y, pwz = G5.z;
    const int eplen_e = __float_as_int(G5.w);
    const R pos_ex = px, pos_ey = py, pos_ez = pz;     // entry position  (= _current_position while steps > 0)
    const R vel_ex = vx, vel_ey = vy, vel_ez = vz;     // entry velocity  (= current_vel, quirk Q4)
    const R ang_ex = wx, ang_ey = wy, ang_ez = wz;     //                 (= current_ang_v)

    // ---- A4: p.stepSimulation, one free rigid body [3P-recall of Bullet3 btMultiBody] -------------------
    R r00, r01, r02, r10, r11, r12, r20, r21, r22;
    {
        R dd = qx * qx + qy * qy + qz * qz + qw * qw;  // btMatrix3x3::setRotation
        R s = R(2.0) / dd;
        R xs = qx * s, ys = qy * s, zs = qz * s;
        R wxs = qw * xs, wys = qw * ys, wzs = qw * zs;
        R xx = qx * xs, xy = qx * ys, xz = qx * zs;
        R yy = qy * ys, yz = qy * zs, zz = qz * zs;
        r00 = R(1.0) - (yy + zz); r01 = xy - wzs;         r02 = xz + wys;
        r10 = xy + wzs;         r11 = R(1.0) - (xx + zz); r12 = yz - wxs;
        r20 = xz - wys;         r21 = yz + wxs;         r22 = R(1.0) - (xx + yy);
    }
    {
        const R dt = K<R>::DT;
        // world -> base
        R vbx = r00 * vx + r10 * vy + r20 * vz, vby = r01 * vx + r11 * vy + r21 * vz, vbz = r02 * vx + r12 * vy + r22 * vz;
        R wbx = r00 * wx + r10 * wy + r20 * wz, wby = r01 * wx + r11 * wy + r21 * wz, wbz = r02 * wx + r12 * wy + r22 * wz;
        R F0 = f32[0], F1 = f32[1], F2 = f32[2], F3 = f32[3];
        R fz = F0 + F1 + F2 + F3;
        // r x F with the prop offsets (+,-) (-,-) (-,+) (+,+) * 0.028 (cf2x.urdf:42,54,66,78)
        R tx = R(0.0), ty = R(0.0);
        tx += -K<R>::ARM * F0; ty -= K<R>::ARM * F0;
        tx += -K<R>::ARM * F1; ty -= -K<R>::ARM * F1;
        tx += K<R>::ARM * F2;  ty -= -K<R>::ARM * F2;
        tx += K<R>::ARM * F3;  ty -= K<R>::ARM * F3;
        const R gwz = -K<R>::G * K<R>::M;
        R gbx = r00 * R(0.0) + r10 * R(0.0) + r20 * gwz, gby = r01 * R(0.0) + r11 * R(0.0) + r21 * gwz,
          gbz = r02 * R(0.0) + r12 * R(0.0) + r22 * gwz;
        R nv = norm3(vbx, vby, vbz), nw = norm3(wbx, wby, wbz);
        R kl = K<R>::LIN_DAMP + K<R>::LIN_DAMP * nv;
        R ka = K<R>::ANG_DAMP + K<R>::ANG_DAMP * nw;
        R Fbx = gbx - K<R>::M * vbx * kl, Fby = gby - K<R>::M * vby * kl, Fbz = fz + gbz - K<R>::M * vbz * kl;
        R Iwx = K<R>::IXX * wbx, Iwy = K<R>::IYY * wby, Iwz = K<R>::IZZ * wbz;
        R gx = wby * Iwz - wbz * Iwy, gy = wbz * Iwx - wbx * Iwz, gz = wbx * Iwy - wby * Iwx;
        R Tbx = tx - gx - Iwx * ka, Tby = ty - gy - Iwy * ka, Tbz = (R)zt32 - gz - Iwz * ka;
        R abx = Fbx * K<R>::INV_M, aby = Fby * K<R>::INV_M, abz = Fbz * K<R>::INV_M;
        R dbx = Tbx * K<R>::INV_IXX, dby = Tby * K<R>::INV_IYY, dbz = Tbz * K<R>::INV_IZZ;
        // base -> world
        R awx = r00 * abx + r01 * aby + r02 * abz, awy = r10 * abx + r11 * aby + r12 * abz, awz = r20 * abx + r21 * aby + r22 * abz;
        R dwx = r00 * dbx + r01 * dby + r02 * dbz, dwy = r10 * dbx + r11 * dby + r12 * dbz, dwz = r20 * dbx + r21 * dby + r22 * dbz;
        const R mv = K<R>::MAX_COORD_VEL;
        wx = clipv(wx + dwx * dt, -mv, mv); vx = clipv(vx + awx * dt, -mv, mv);   // applyDeltaVeeMultiDof
        wy = clipv(wy + dwy * dt, -mv, mv); vy = clipv(vy + awy * dt, -mv, mv);
        wz = clipv(wz + dwz * dt, -mv, mv); vz = clipv(vz + awz * dt, -mv, mv);
        px += dt * vx; py += dt * vy; pz += dt * vz;                              // stepPositionsMultiDof
        R fAngle = norm3(wx, wy, wz);
        if (fAngle * dt > R(0.25) * K<R>::PI) fAngle = R(0.5) * (R(0.5) * K<R>::PI) / dt;
        // axis = w * sin(h)/|w| with h = |w| dt/2, i.e. w * (dt/2) * sinc(h): one formula covers Bullet's Taylor
        // branch (|w| < 1e-3, identical to 1e-24) and its clamped branch, and needs no divide.
        R hh = R(0.5) * fAngle * dt, sinc, aw;
        sinc_cos_small<R>(hh * hh, sinc, aw);
        R k = (R(0.5) * dt) * sinc;
        R ax = wx * k, ay = wy * k, az = wz * k;
        R nx = aw * qx + ax * qw + ay * qz - az * qy;
        R ny = aw * qy + ay * qw + az * qx - ax * qz;
        R nz = aw * qz + az * qw + ax * qy - ay * qx;
        R nw_ = aw * qw - ax * qx - ay * qy - az * qz;
        R inv = R(1.0) / sqrt(nx * nx + ny * ny + nz * nz + nw_ * nw_);
        qx = nx * inv; qy = ny * inv; qz = nz * inv; qw = nw_ * inv;
    }

    // ---- A5: p.getEulerFromQuaternion [3P-recall of pybullet.c] ------------------------------------------
    // get_forward_vector (PBDroneEnv.py:588-597) = (cos yaw cos pitch, sin yaw cos pitch, sin pitch) follows
    // algebraically from the same quaternion terms: sin pitch = sarg, cos pitch = sqrt(1 - sarg^2) (pitch is an
    // arcsine, so its cosine is non-negative), (cos yaw, sin yaw) = (yc, ys)/hypot(yc, ys) -- no sin/cos calls.
    R roll, pitch, yaw, fwx, fwy, fwz;
    {
        R sqx = qx * qx, sqy = qy * qy, sqz = qz * qz, squ = qw * qw;
        R sarg = R(-2.0) * (qx * qz - qw * qy);
        if (sarg <= R(-0.99999) || sarg >= R(0.99999)) {       // gimbal-lock branches: rare, keep them literal
            roll = R(0.0);
            if (sarg < R(0.0)) { pitch = R(-0.5) * K<R>::PI; yaw = R(2.0) * atan2(qx, -qy); }
            else { pitch = R(0.5) * K<R>::PI; yaw = R(2.0) * atan2(-qx, qy); }
            R cpit = cos(pitch);
            fwx = cos(yaw) * cpit; fwy = sin(yaw) * cpit; fwz = sin(pitch);
        } else {
            R ys = R(2.0) * (qx * qy + qw * qz), yc = squ + sqx - sqy - sqz;
            roll = atan2(R(2.0) * (qy * qz + qw * qx), squ - sqx - sqy + sqz);
            pitch = asin(sarg);
            yaw = atan2(ys, yc);
            R cpit = sqrt(R(1.0) - sarg * sarg);
            R hy = sqrt(ys * ys + yc * yc);
            R cyaw = R(1.0), syaw = R(0.0);
            if (hy > R(0.0)) { R rh = R(1.0) / hy; cyaw = yc * rh; syaw = ys * rh; }
            fwx = cyaw * cpit; fwy = syaw * cpit; fwz = sarg;
        }
    }
    // rotation entry R[2][2] of the NEW attitude (ground-contact approximation only)
    const R r22n = R(1.0) - (qx * (qx * (R(2.0) / (qx * qx + qy * qy + qz * qz + qw * qw))) +
                             qy * (qy * (R(2.0) / (qx * qx + qy * qy + qz * qz + qw * qw))));

    // ---- A6: _computeObs (PBDroneEnv.py:296-336, :338-398), stale distance d_e (quirk Q1) -----------------
    float o[DN_OBS_DIM];
    {
        const R fmax = (R)FLT_MAX;
        o[0] = (float)clipv(px * c.inv_dim[0], -fmax, fmax);
        o[1] = (float)clipv(py * c.inv_dim[1], -fmax, fmax);
        o[2] = (float)clipv(pz * c.inv_dim[2], -fmax, fmax);
        o[3] = (float)(clipv(roll, -K<R>::PI, K<R>::PI) * K<R>::INV_PI);
        o[4] = (float)(clipv(pitch, -K<R>::PI, K<R>::PI) * K<R>::INV_PI);
        o[5] = (float)clipv(yaw * K<R>::INV_PI, -fmax, fmax);
        o[6] = (float)(clipv(vx, R(-3.0), R(3.0)) * K<R>::THIRD);
        o[7] = (float)(clipv(vy, R(-3.0), R(3.0)) * K<R>::THIRD);
        o[8] = (float)(clipv(vz, R(-1.0), R(1.0)) * K<R>::THIRD);
        R nw = norm3(wx, wy, wz);
        if (nw != R(0.0)) { R rw = R(1.0) / nw; o[9] = (float)(wx * rw); o[10] = (float)(wy * rw); o[11] = (float)(wz * rw); }
        else { o[9] = (float)wx; o[10] = (float)wy; o[11] = (float)wz; }
        o[12] = p.include_distance ? (float)clipv(d_e * c.inv_max_target_dist, -fmax, fmax) : 0.0f;
    }

    // ---- A7 + A8: _computeReward (PBDroneEnv.py:475-571), _computeTerminated (:456-473) --------------------
    int idx = m_e.idx, just_found = m_e.just_found, is_done = 0;
    R d_prev = dprev_e;
    R reward;
    const bool coll1 = has_collision<R>(p, c, s_tab, px, py, pz, r22n, idx);
    bool terminated;
    if (coll1) {                                       // :489-490 (entry _is_done is always False here)
        reward = R(-10.0);
        terminated = true;
    } else {
        const R fx = fwx, fy = fwy, fz = fwz;
        if (d_e <= c.threshold) {                      // :539
            idx += 1;
            float r32 = 0.0f;
            if (idx == p.num_waypoints) { r32 = r32 + 200.0f; is_done = 1; }          // :542-546
            else {
                r32 = r32 + 75.0f;                                                    // :548-552
                r32 = r32 + (float)(orientation_reward<R>(fx, fy, fz, px, py, pz, s_tab + idx * DN_T_STRIDE) * 5);
                just_found = 1;
            }
            d_prev = d_e;
            reward = (R)(r32 / 25.0f);
            // second _computeTerminated (BaseAviary.py:448) sees the advanced index
            terminated = is_done ? true : has_collision<R>(p, c, s_tab, px, py, pz, r22n, idx);
        } else {
            R r = R(0.0);
            r = r + exp(R(-2.0) * d_e) * R(3.0);                                      // :555
            r = r + (just_found ? R(0.0) : (dprev_e - d_e) * R(3000.0));              // :556
            r = r + (R)(orientation_reward<R>(fx, fy, fz, px, py, pz, s_tab + idx * DN_T_STRIDE) * 3);   // :557
            // smoothness_reward (:599-607) on the stale post-step copies (quirk Q4)
            R la = norm3(vel_ex - pvx, vel_ey - pvy, vel_ez - pvz);
            R aa = norm3(ang_ex - pwx, ang_ey - pwy, ang_ez - pwz);
            R lp = (la > R(0.7)) ? -fabs(la) : R(0.0);
            R ap = (aa > R(0.3)) ? -fabs(aa) : R(0.0);
            r = r + (lp + ap);                                                        // :558
            just_found = 0;
            d_prev = d_e;
            reward = r * K<R>::INV_25;
            terminated = false;
        }
    }
    const bool truncated = p.max_steps <= m_e.steps;   // :444-454, evaluated on the un-incremented _steps
    const int found = idx;
    const bool done = terminated || truncated;

    // ---- A9: _update_state_post_step (:201-223), skipped on a terminated step (quirk Q5) -----------------
    int steps = m_e.steps;
    R d = d_e;
    R npvx = pvx, npvy = pvy, npvz = pvz, npwx = pwx, npwy = pwy, npwz = pwz;
    if (!terminated) {
        steps += 1;
        npvx = vel_ex; npvy = vel_ey; npvz = vel_ez;
        npwx = ang_ex; npwy = ang_ey; npwz = ang_ez;
        const R *wp = s_tab + idx * DN_T_STRIDE;
        d = norm3(wp[0] - px, wp[1] - py, wp[2] - pz);
    }

    // ---- A11: Monitor + SubprocVecEnv worker ---------------------------------------------------------------
    R ep_ret = epret_e + reward;
    int ep_len = eplen_e + 1;

    // sensor noise / per-drone normaliser act on the step observation (which is also terminal_observation)
    double rms_count = 0.0;
    if (NORM) rms_count = p.st.rms_count[i];
    if (NOISE && p.obs_noise_sigma > 0.0f) add_obs_noise(p, gid, step_count, 1u, o);
    if (NORM) normalize_obs(p, i, active, rms_count, o);

    const unsigned long long done_ballot = __ballot(done && active);
    if (done_ballot != 0ull) {                         // wave-uniform: most waves skip the whole reset path
        if (done) {
            if (active) {
                if (io.terminal_obs) {
#pragma unroll
                    for (int k = 0; k < DN_OBS_DIM; ++k) io.terminal_obs[i * DN_OBS_DIM + k] = o[k];
                }
                if (io.ep_return) io.ep_return[i] = (float)ep_ret;
                if (io.ep_length) io.ep_length[i] = ep_len;
            }
            // PBDroneEnv.reset (:609-665): _current_position is NOT reset (quirk Q3)
            R cpx, cpy, cpz;
            if (!terminated) { cpx = px; cpy = py; cpz = pz; }            // post-step ran: it is the new position
            else if (m_e.steps > 0) { cpx = pos_ex; cpy = pos_ey; cpz = pos_ez; }
            else { const float4 G6 = p.st.g6[i]; cpx = G6.x; cpy = G6.y; cpz = G6.z; }
            if (active && !(terminated && m_e.steps == 0)) p.st.g6[i] = make_float4((float)cpx, (float)cpy, (float)cpz, 0.0f);
            reset_obs<R>(p, c, d, o);                                     // BaseAviary.py:318 before :617-658 (Q2)
            if (NOISE && p.obs_noise_sigma > 0.0f) add_obs_noise(p, gid, step_count, 5u, o);
            if (NORM) normalize_obs(p, i, active, rms_count, o);
            px = c.spawn[0]; py = c.spawn[1]; pz = c.spawn[2];
            qx = R(0.0); qy = R(0.0); qz = R(0.0); qw = R(1.0);
            vx = vy = vz = wx = wy = wz = R(0.0);
            npvx = npvy = npvz = npwx = npwy = npwz = R(0.0);
            d = norm3(cpx - s_tab[0], cpy - s_tab[1], cpz - s_tab[2]);    // :651
            d_prev = d;                                                   // :652
            idx = 0; steps = 0; just_found = 0;
            ep_ret = R(0.0); ep_len = 0;
        }
        // wave-level reduction of the episode statistics -> this workgroup's slot (no atomics)
        long long s_ep = done && active ? 1 : 0, s_tr = (done && active && truncated && !terminated) ? 1 : 0;
        long long s_co = (done && active && is_done) ? 1 : 0;
        long long s_len = done && active ? (long long)(eplen_e + 1) : 0, s_fd = done && active ? (long long)found : 0;
        long long s_ret = done && active ? (long long)llrint((double)(epret_e + reward) * 1e6) : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s_ep += __shfl_xor(s_ep, off); s_tr += __shfl_xor(s_tr, off); s_co += __shfl_xor(s_co, off);
            s_len += __shfl_xor(s_len, off); s_fd += __shfl_xor(s_fd, off); s_ret += __shfl_xor(s_ret, off);
        }
        if (lane == 0) {
            DnStatSlot sl = p.st.stats[blockIdx.x];
            sl.episodes += s_ep; sl.truncated += s_tr; sl.completed += s_co;
            sl.sum_len += s_len; sl.sum_found += s_fd; sl.sum_ret_fix += s_ret;
            p.st.stats[blockIdx.x] = sl;
        }
    }
    if (NORM && active) p.st.rms_count[i] = rms_count;

    // ---- hand the state back as float32 groups; scalars and the observation tile go straight to HBM -----------
    G0 = make_float4((float)px, (float)py, (float)pz, (float)d);
    G1 = make_float4((float)qx, (float)qy, (float)qz, (float)qw);
    G2 = make_float4((float)vx, (float)vy, (float)vz, (float)d_prev);
    G3 = make_float4((float)wx, (float)wy, (float)wz, pack_meta(steps, idx, just_found));
    G4 = make_float4((float)npvx, (float)npvy, (float)npvz, (float)ep_ret);
    G5 = make_float4((float)npwx, (float)npwy, (float)npwz, __int_as_float(ep_len));
    if (active) {
        io.reward[i] = (float)reward;
        io.done[i] = (uint8_t)done;
        io.truncated[i] = (uint8_t)(truncated && !terminated);
        io.found_targets[i] = found;
    }
    if (io.done_mask && lane == 0) io.done_mask[blockIdx.x] = done_ballot;
    store_obs_tile(s_tile, io.obs, tile_base, p.n, lane, active, o);
}

// One control step per launch: what VecEnv.step() maps to when a policy sits between steps.
template <typename R, bool NORM, bool NOISE>
__global__ __launch_bounds__(DN_BLOCK) void dn_step_kernel(const DnParams p, const DnStepIO io)
{
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    const int lane = threadIdx.x;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const bool active = tile_base + lane < p.n;
    const long long i = active ? tile_base + lane : p.n - 1;   // inactive lanes shadow the last drone, never store
    // issue every load up front (6 x 16 B state + 16 B action per lane), then stage the table
    const float4 A = reinterpret_cast<const float4 *>(io.actions)[i];
    float4 G0 = p.st.g0[i], G1 = p.st.g1[i], G2 = p.st.g2[i], G3 = p.st.g3[i], G4 = p.st.g4[i], G5 = p.st.g5[i];
    stage_table<R>(p, s_tab);
    step_body<R, NORM, NOISE>(p, consts<R>(p), s_tab, s_tile, io, p.step_count, i, tile_base, lane, active, A,
                              G0, G1, G2, G3, G4, G5);
    if (active) {
        p.st.g0[i] = G0; p.st.g1[i] = G1; p.st.g2[i] = G2; p.st.g3[i] = G3; p.st.g4[i] = G4; p.st.g5[i] = G5;
    }
}

// K control steps per launch for open-loop action sequences (dn_step_many): the state is read once, stays in
// registers for K steps and is written once; per step only the action (16 B) comes in and the outputs (62 B) go
// out, and the K-1 kernel boundaries disappear.  Buffers are step-major [K, N, ...].
template <typename R, bool NORM, bool NOISE>
__global__ __launch_bounds__(DN_BLOCK) void dn_step_many_kernel(const DnParams p, const DnStepIO io0, const int k_steps)
{
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    const int lane = threadIdx.x;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const bool active = tile_base + lane < p.n;
    const long long i = active ? tile_base + lane : p.n - 1;
    const float4 *act = reinterpret_cast<const float4 *>(io0.actions);
    float4 A = act[i];
    float4 G0 = p.st.g0[i], G1 = p.st.g1[i], G2 = p.st.g2[i], G3 = p.st.g3[i], G4 = p.st.g4[i], G5 = p.st.g5[i];
    stage_table<R>(p, s_tab);
    const long long n = p.n, words = (p.n + 63) / 64;
#pragma clang loop unroll(disable)
    for (int t = 0; t < k_steps; ++t) {
        // prefetch the next step's action while this step computes
        const float4 A_next = act[(long long)(t + 1 < k_steps ? t + 1 : t) * n + i];
        DnStepIO io;
        io.actions = nullptr;
        io.obs = io0.obs + (long long)t * n * DN_OBS_DIM;
        io.reward = io0.reward + (long long)t * n;
        io.done = io0.done + (long long)t * n;
        io.truncated = io0.truncated + (long long)t * n;
        io.found_targets = io0.found_targets + (long long)t * n;
        io.terminal_obs = io0.terminal_obs ? io0.terminal_obs + (long long)t * n * DN_OBS_DIM : nullptr;
        io.ep_return = io0.ep_return ? io0.ep_return + (long long)t * n : nullptr;
        io.ep_length = io0.ep_length ? io0.ep_length + (long long)t * n : nullptr;
        io.done_mask = io0.done_mask ? io0.done_mask + (long long)t * words : nullptr;
        step_body<R, NORM, NOISE>(p, consts<R>(p), s_tab, s_tile, io, p.step_count + (unsigned)t, i, tile_base, lane,
                                  active, A, G0, G1, G2, G3, G4, G5);
        A = A_next;
    }
    if (active) {
        p.st.g0[i] = G0; p.st.g1[i] = G1; p.st.g2[i] = G2; p.st.g3[i] = G3; p.st.g4[i] = G4; p.st.g5[i] = G5;
    }
}

// =====================================================================================================
// VecEnv.reset(): every drone goes through Monitor.reset / NormalizeObservation.reset / PBDroneEnv.reset.
// =====================================================================================================
template <typename R>
__global__ __launch_bounds__(DN_BLOCK) void dn_reset_kernel(const DnParams p, float *obs)
{
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    const int lane = threadIdx.x;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const bool active = tile_base + lane < p.n;
    const long long i = active ? tile_base + lane : p.n - 1;
    const DnConsts<R> &c = consts<R>(p);
    const float4 G0 = p.st.g0[i], G3 = p.st.g3[i], G6 = p.st.g6[i];
    stage_table<R>(p, s_tab);
    const Meta m = unpack_meta(G3.w);
    R cpx, cpy, cpz;
    if (m.steps > 0) { cpx = G0.x; cpy = G0.y; cpz = G0.z; } else { cpx = G6.x; cpy = G6.y; cpz = G6.z; }
    float o[DN_OBS_DIM];
    reset_obs<R>(p, c, (R)G0.w, o);
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    if (p.obs_noise_sigma > 0.0f) add_obs_noise(p, gid, p.step_count, 5u, o);
    if (p.normalize_obs) {
        double cnt = p.st.rms_count[i];
        normalize_obs(p, i, active, cnt, o);
        if (active) p.st.rms_count[i] = cnt;
    }
    R d = norm3(cpx - s_tab[0], cpy - s_tab[1], cpz - s_tab[2]);
    if (active) {
        p.st.g0[i] = make_float4((float)c.spawn[0], (float)c.spawn[1], (float)c.spawn[2], (float)d);
        p.st.g1[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        p.st.g2[i] = make_float4(0.0f, 0.0f, 0.0f, (float)d);
        p.st.g3[i] = make_float4(0.0f, 0.0f, 0.0f, pack_meta(0, 0, 0));
        p.st.g4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        p.st.g5[i] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0));
        p.st.g6[i] = make_float4((float)cpx, (float)cpy, (float)cpz, 0.0f);
    }
    store_obs_tile(s_tile, obs, tile_base, p.n, lane, active, o);
}

// =====================================================================================================
// N1: GAE, one lane per drone, time-reversed scan (cleanRLPPO.py:234-248).  float32, unfused, in the
// reference's operation order so the result is bit-identical to the torch float32 loop.
// =====================================================================================================
__global__ __launch_bounds__(256) void dn_gae_kernel(const float *__restrict__ rewards, const float *__restrict__ values,
                                                     const uint8_t *__restrict__ dones, const float *__restrict__ last_values,
                                                     const uint8_t *__restrict__ last_dones, long long T, long long N,
                                                     float gamma, float gl, float *__restrict__ adv, float *__restrict__ ret)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float last = 0.0f;
    float nnt = 1.0f - (float)last_dones[i];
    float nv = last_values[i];
    for (long long t = T - 1; t >= 0; --t) {
        const float v = values[t * N + i];
        float gv = gamma * nv;
        float delta = rewards[t * N + i] + gv * nnt;
        delta = delta - v;
        float k = gl * nnt;
        last = delta + k * last;
        adv[t * N + i] = last;
        ret[t * N + i] = last + v;
        nnt = 1.0f - (float)dones[t * N + i];
        nv = v;
    }
}

// =====================================================================================================
// Episode-done compaction: ballot words -> ordered index list (popcount + block-wide exclusive scan).
// Single workgroup of 1024 lanes, each lane walks ceil(words/1024) consecutive words.
// =====================================================================================================
__global__ __launch_bounds__(1024) void dn_compact_kernel(const unsigned long long *__restrict__ mask, long long n,
                                                          int32_t *__restrict__ indices, int32_t *__restrict__ count)
{
    __shared__ int s_wave[16];
    const long long words = (n + 63) / 64;
    const long long per = (words + 1023) / 1024;
    const long long w0 = (long long)threadIdx.x * per;
    int mine = 0;
    for (long long w = w0; w < w0 + per && w < words; ++w) mine += __popcll(mask[w]);
    // exclusive scan: within the wave by shuffles, across the 16 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += s_wave[k];
    int pos = base + incl - mine;
    for (long long w = w0; w < w0 + per && w < words; ++w) {
        unsigned long long mword = mask[w];
        while (mword) {
            int b = __ffsll((long long)mword) - 1;
            indices[pos++] = (int32_t)(w * 64 + b);
            mword &= mword - 1;
        }
    }
    if (threadIdx.x == 1023) *count = base + incl;
}

__global__ __launch_bounds__(256) void dn_fill4_kernel(float4 *dst, float4 v, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ __launch_bounds__(256) void dn_filld_kernel(double *dst, double v, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}

}  // namespace

hipError_t dn_launch_fill4(float4 *dst, float4 v, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(dn_fill4_kernel, dim3(grid), dim3(256), 0, stream, dst, v, n);
    return hipGetLastError();
}

hipError_t dn_launch_filld(double *dst, double v, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(dn_filld_kernel, dim3(grid), dim3(256), 0, stream, dst, v, n);
    return hipGetLastError();
}

hipError_t dn_launch_step(const DnParams &p, const DnStepIO &io, bool f32, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    const bool norm = p.normalize_obs != 0;
    const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
#define DN_LAUNCH(R, NORM, NOISE) \
    hipLaunchKernelGGL((dn_step_kernel<R, NORM, NOISE>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io)
    if (f32) {
        if (norm) { if (noise) DN_LAUNCH(float, true, true); else DN_LAUNCH(float, true, false); }
        else { if (noise) DN_LAUNCH(float, false, true); else DN_LAUNCH(float, false, false); }
    } else {
        if (norm) { if (noise) DN_LAUNCH(double, true, true); else DN_LAUNCH(double, true, false); }
        else { if (noise) DN_LAUNCH(double, false, true); else DN_LAUNCH(double, false, false); }
    }
#undef DN_LAUNCH
    return hipGetLastError();
}

hipError_t dn_launch_step_many(const DnParams &p, const DnStepIO &io, int k, bool f32, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    const bool norm = p.normalize_obs != 0;
    const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
#define DN_LAUNCH(R, NORM, NOISE) \
    hipLaunchKernelGGL((dn_step_many_kernel<R, NORM, NOISE>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, k)
    if (f32) {
        if (norm) { if (noise) DN_LAUNCH(float, true, true); else DN_LAUNCH(float, true, false); }
        else { if (noise) DN_LAUNCH(float, false, true); else DN_LAUNCH(float, false, false); }
    } else {
        if (norm) { if (noise) DN_LAUNCH(double, true, true); else DN_LAUNCH(double, true, false); }
        else { if (noise) DN_LAUNCH(double, false, true); else DN_LAUNCH(double, false, false); }
    }
#undef DN_LAUNCH
    return hipGetLastError();
}

hipError_t dn_launch_reset(const DnParams &p, float *obs, bool f32, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    if (f32) hipLaunchKernelGGL(dn_reset_kernel<float>, dim3(grid), dim3(DN_BLOCK), 0, stream, p, obs);
    else hipLaunchKernelGGL(dn_reset_kernel<double>, dim3(grid), dim3(DN_BLOCK), 0, stream, p, obs);
    return hipGetLastError();
}

hipError_t dn_launch_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
                         const uint8_t *last_dones, long long T, long long N, float gamma, float gl, float *adv,
                         float *ret, hipStream_t stream)
{
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(dn_gae_kernel, dim3(grid), dim3(256), 0, stream, rewards, values, dones, last_values,
                       last_dones, T, N, gamma, gl, adv, ret);
    return hipGetLastError();
}

hipError_t dn_launch_compact(const unsigned long long *mask, long long n, int32_t *indices, int32_t *count,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(dn_compact_kernel, dim3(1), dim3(1024), 0, stream, mask, n, indices, count);
    return hipGetLastError();
}
