// dn_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the drone-navigation environment step.
//
// One drone per lane, 64 drones (one tile) per workgroup, one to eight 64-lane waves per tile (the step cut by data
// dependency over waves that exchange LDS mail).  The whole reference step
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
// -ffp-contract=off: the float32 action chain rounds every operation as numpy does, and in the float64 part every fused
// multiply-add is WRITTEN OUT (FM<R>::fma) in one fixed nesting, never left to the compiler's licence, so that every kernel
// shape -- hence every split of a fleet over ranks -- produces the same bits.
//
// Reference citations are file:line under /root/reference:
//   PBDroneEnv.py = Sol/Model/Environments/PBDroneEnv.py, BaseAviary.py = Sol/PyBullet/BaseAviary.py,
//   env_utils.py = Sol/Model/env_utils.py, normalize.py = Sol/Model/Environments/normalize.py.
#include "dn_internal.h"
#include "dn_action_sat.h"

#ifndef DN_TU
#define DN_TU 1      // see dn_launch_step_many: 1 = this file as such, 2 = through dn_kernels_mw.hip
#endif

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
// Bit-exact against numpy float32 (golden actions.npz).  IEEE divide and sqrt expand to 10 and 23 instructions on
// gfx950 (scaling, fix-up); every divisor here is a constant and every operand range is known, so the chain uses
//   a / c  =  q0 + fma(-q0, c, a) * RN(1/c),  q0 = a * RN(1/c)        (Markstein's correction, 3 instructions)
//   sqrt   =  v_sqrt_f32 (1 ulp) + a one-ulp-down / one-ulp-up test with fma residuals
// which tests/tools/check_action_chain_exact.c proves identical to the correctly rounded results for EVERY float32
// input that can reach them (2.1e9 actions, 2e7 thrusts, 1e7 roots; exhaustive).  Clips are v_med3_f32; np.clip's
// NaN propagation is restored by the final select.
constexpr float DEN32 = A_HIGH32 - A_LOW32;
constexpr float INV_DEN32 = 1.0f / DEN32, INV_KF32 = 1.0f / KF32, INV_SCALE32 = 1.0f / PWM2RPM_SCALE32;
DN_DEV float div_const32(float a, float c, float inv_c)
{
    const float q0 = a * inv_c;
    const float r = __builtin_fmaf(-q0, c, a);
    return __builtin_fmaf(r, inv_c, q0);
}
DN_DEV float sqrt_rn32(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float out = rm <= 0.0f ? sm : s;
    out = rp > 0.0f ? sp : out;
    return out;
}
DN_DEV float rescale_unclipped32(float a)
{   // PBDroneEnv.rescale_action, PBDroneEnv.py:949-971, before its final clip to [-1, 1]
    const float ac = __builtin_amdgcn_fmed3f(a, -2.0f, 2.0f);   // beyond +-2 the result is saturated anyway
    const float num = ac - A_LOW32;
    const float q = div_const32(num, DEN32, INV_DEN32);
    const float m = 2.0f * q;                // (high - low) = 1 - (-1)
    return -1.0f + m;
}
DN_DEV float rescale_action32(float a) { return __builtin_amdgcn_fmed3f(rescale_unclipped32(a), -1.0f, 1.0f); }
// cmd: the (rescaled) action.  rescale_action's own clip to [-1, 1] may be left out by the caller: the thrust clip below
// is to [a_low, a_high], a sub-interval, and clip(clip(x, -1, 1), lo, hi) = clip(x, lo, hi).  a: the raw action (NaN test).
DN_DEV float rotor_force_from_cmd(float cmd, float a, float &torque, float *rpm_out = nullptr, bool nan_check = true)
{
    // PBDroneEnv._preprocessAction, PBDroneEnv.py:889.  The clip makes thrust >= a_low > 0, so cmd2pwm's
    // maximum(thrust, 0) (env_utils.py:29) is the identity.
    const float thrust = __builtin_amdgcn_fmed3f(cmd, A_LOW32, A_HIGH32);
    const float t = div_const32(thrust, KF32, INV_KF32);             // env_utils.py:30 (n_motor = 1)
    const float s = sqrt_rn32(t);
    float pwm = div_const32(s - PWM2RPM_CONST32, PWM2RPM_SCALE32, INV_SCALE32);
    pwm = __builtin_amdgcn_fmed3f(pwm, MIN_PWM32, MAX_PWM32);        // env_utils.py:39
    const float r0 = PWM2RPM_SCALE32 * pwm;
    const float rpm = r0 + PWM2RPM_CONST32;          // pwm2rpm, env_utils.py:58
    const float sq = rpm * rpm;                      // BaseAviary._physics, BaseAviary.py:776-777
    const bool nan = nan_check && a != a;            // np.clip / sqrt propagate NaN
    if (rpm_out) *rpm_out = nan ? a : rpm;
    torque = nan ? a : sq * KM32;
    return nan ? a : sq * KF32;
}
DN_DEV float rotor_force_from_action(float a, bool normalize_actions, float &torque, float *rpm_out = nullptr, bool nan_check = true)
{
    return rotor_force_from_cmd(normalize_actions ? rescale_unclipped32(a) : a, a, torque, rpm_out, nan_check);
}
// The saturation fast path (round 6).  The thrust clip of rotor_force_from_cmd is the chain's ONLY read of the command, so every
// command <= a_low yields the force / torque of a_low and every command >= a_high those of a_high; rescale_action is monotone in
// the action, so in raw-action space that is two float32 thresholds (csrc/dn_action_sat.h; tests/tools/check_action_chain_exact.c
// proves thresholds and constants against the literal numpy-order chain for all 2^32 - 2^24 non-NaN actions).  99.6 % of U(-1,1)
// actions -- and of a sigma = 1 Gaussian policy's -- lie outside the 0.0072-wide band in between: per rotor the WAVE takes the chain
// only if one of its lanes is inside the band (or NaN: both compares false), 21 % of rotor evaluations at U(-1,1); then every
// lane takes it, and a saturated lane's chain returns the same two constants: bit-exact by construction either way.
constexpr float ACT_SAT_LO32 = __builtin_bit_cast(float, DN_ACT_SAT_LO_BITS), ACT_SAT_HI32 = __builtin_bit_cast(float, DN_ACT_SAT_HI_BITS);
constexpr float F_LO32 = __builtin_bit_cast(float, DN_F_LO_BITS), F_HI32 = __builtin_bit_cast(float, DN_F_HI_BITS);
constexpr float TQ_LO32 = __builtin_bit_cast(float, DN_TQ_LO_BITS), TQ_HI32 = __builtin_bit_cast(float, DN_TQ_HI_BITS);
static_assert(__builtin_bit_cast(unsigned, A_LOW32) == DN_A_LOW_BITS && __builtin_bit_cast(unsigned, A_HIGH32) == DN_A_HIGH_BITS,
              "dn_action_sat.h was derived for other action bounds");
DN_DEV float rotor_force_sat(const float a, const bool normalize_actions, float &torque)
{
    const float t_lo = normalize_actions ? ACT_SAT_LO32 : A_LOW32, t_hi = normalize_actions ? ACT_SAT_HI32 : A_HIGH32;   // wave-uniform
    const bool hi = a >= t_hi, lo = a <= t_lo;
    float f = hi ? F_HI32 : F_LO32;
    torque = hi ? TQ_HI32 : TQ_LO32;
    if (__ballot(!(hi || lo)) != 0ull)            // wave-uniform: a lane inside the band, or a NaN (np.clip / sqrt propagate it: the chain's own select)
        f = rotor_force_from_cmd(normalize_actions ? rescale_unclipped32(a) : a, a, torque, nullptr, true);
    return f;
}

DN_DEV float z_torque32(const float tq[4])
{   // z_torque = -t0 + t1 - t2 + t3, left to right in float32 (BaseAviary.py:780)
    float z = -tq[0];
    z = z + tq[1];
    z = z - tq[2];
    z = z + tq[3];
    return z;
}

// ---- N4: options that are present but unreachable in the reference (BaseAviary.step pins Physics.PYB,
// BaseAviary.py:411; PBDroneEnv overrides _preprocessAction).  Compiled only into the XOPT kernels.  These helpers
// stay outside every `fp contract(fast)` region: their float32 halves follow numpy's operation order unfused.
constexpr double GND_EFF_COEFF = 11.36859, PROP_RADIUS = 2.31348e-2, DRAG_XY = 9.1785e-7, DRAG_Z = 10.311e-7;   // cf2x.urdf:5
constexpr double HOVER_RPM = 14468.429183500699;         // sqrt(GRAVITY / (4 KF)), BaseAviary.py:164
constexpr double GND_EFF_H_CLIP = 0.03776371349209501;   // BaseAviary.py:175-176
struct Extras {
    int gnd, drag, rpm_f32;     // Physics.PYB_GND / PYB_DRAG terms on; the rpm array is float32 (ActionType.THRUST chain)
    double rpm[4];              // this step's clipped_action (rpm)
    float4 last;                // BaseAviary.last_clipped_action: the previous step's rpm, zeros after a reset
    double damp;                // btMultiBody m_linearDamping = m_angularDamping: 0.04, or 0 with dn_config.zero_damping
};
DN_DEV double gnd_effect_rotor(double rpm, bool rpm_f32)
{   // np.array(rpm**2) * KF * GND_EFF_COEFF, BaseAviary.py:822 -- float32 while the rpm array is
    if (rpm_f32) {
        const float r = (float)rpm;
        float t = r * r;
        t = t * KF32;
        t = t * (float)GND_EFF_COEFF;
        return (double)t;
    }
    double t = rpm * rpm;
    t = t * 3.16e-10;
    return t * GND_EFF_COEFF;
}
DN_DEV double drag_omega_sum(const float4 last, bool rpm_f32)
{   // np.sum(np.array(2*np.pi*rpm/60)), BaseAviary.py:852 (left-to-right reduce)
    const float r[4] = {last.x, last.y, last.z, last.w};
    if (rpm_f32) {
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = (float)(2.0 * 3.14159265358979323846) * r[j];
            w[j] = m / 60.0f;
        }
        float sum = w[0] + w[1];
        sum = sum + w[2];
        sum = sum + w[3];
        return (double)sum;
    }
    double w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (2.0 * 3.14159265358979323846) * (double)r[j] / 60.0;
    double sum = w[0] + w[1];
    sum = sum + w[2];
    return sum + w[3];
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
// The EXACT form (draws that feed the dynamics: action noise, the policies' sampling, random spawn): Box-Muller on one pair of Philox words, float64 throughout, every operation spelled out (no libm call: the device
// library's log / sin / cos carry argument-reduction paths for inputs that cannot occur here and cost ~300 instructions
// a pair; this is ~80).  u = (r + 0.5) / 2^32 lies strictly inside (0, 1).
//   ln u1:  u1 = m 2^e, m in [sqrt(1/2), sqrt(2)),  ln m = 2 atanh(s) = 2 s (1 + s^2/3 + ... + s^14/15),  s = (m-1)/(m+1),
//           |s| <= 0.1716: truncation 3e-14 relative, and no cancellation as u1 -> 1 (e = 0, m - 1 exact);
//   angle:  2 pi u2 = k pi/2 + theta, k = rint(4 u2), theta = 2 pi (u2 - k/4) in [-pi/4, pi/4] (the subtraction is
//           exact), sin / cos Taylor to theta^13 / theta^14 (2e-14), quadrant fix-up by k.
// Equal to the libm form (log, sqrt, cos, sin of the C library) to ~1e-13 before the float32 cast.
DN_DEV void box_muller_pair64(unsigned ra, unsigned rb, float &z0, float &z1)
{
    const double u1 = ((double)ra + 0.5) * (1.0 / 4294967296.0);
    const double u2 = ((double)rb + 0.5) * (1.0 / 4294967296.0);
    double m = __builtin_amdgcn_frexp_mant(u1);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(u1);
    if (m < 0.70710678118654752440) { m = m + m; e = e - 1; }
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    double q = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, q, r);
    q = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, q, r);
    const double sv = num * r, s2 = sv * sv;
    double pl = 1.0 / 15.0;
    pl = __builtin_fma(pl, s2, 1.0 / 13.0);
    pl = __builtin_fma(pl, s2, 1.0 / 11.0);
    pl = __builtin_fma(pl, s2, 1.0 / 9.0);
    pl = __builtin_fma(pl, s2, 1.0 / 7.0);
    pl = __builtin_fma(pl, s2, 1.0 / 5.0);
    pl = __builtin_fma(pl, s2, 1.0 / 3.0);
    pl = __builtin_fma(pl, s2, 1.0);
    const double ln_u1 = __builtin_fma((double)e, 0.69314718055994530942, (sv + sv) * pl);
    const double t = -2.0 * ln_u1;                         // > 0
    double y = __builtin_amdgcn_rsq(t);
    y = y * __builtin_fma(-0.5 * t * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * t * y, y, 1.5);
    const double rad = t * y;
    const double k4 = __builtin_rint(u2 * 4.0);
    const double th = (2.0 * 3.14159265358979323846) * __builtin_fma(k4, -0.25, u2);
    const double t2 = th * th;
    double ps = 1.0 / 6227020800.0;                        //  1/13!
    ps = __builtin_fma(ps, t2, -1.0 / 39916800.0);         // -1/11!
    ps = __builtin_fma(ps, t2, 1.0 / 362880.0);
    ps = __builtin_fma(ps, t2, -1.0 / 5040.0);
    ps = __builtin_fma(ps, t2, 1.0 / 120.0);
    ps = __builtin_fma(ps, t2, -1.0 / 6.0);
    const double sn = __builtin_fma(th * t2, ps, th);
    double pc = -1.0 / 87178291200.0;                      // -1/14!
    pc = __builtin_fma(pc, t2, 1.0 / 479001600.0);         //  1/12!
    pc = __builtin_fma(pc, t2, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, t2, 1.0 / 40320.0);
    pc = __builtin_fma(pc, t2, -1.0 / 720.0);
    pc = __builtin_fma(pc, t2, 1.0 / 24.0);
    pc = __builtin_fma(pc, t2, -0.5);
    const double cs = __builtin_fma(pc, t2, 1.0);
    const int k = (int)k4 & 3;
    const double c = (k & 1) ? sn : cs, d = (k & 1) ? cs : sn;          // odd quadrants swap the two ...
    const double cosv = (k == 1 || k == 2) ? -c : c;                    // ... and the signs follow cos(a + pi/2) = -sin a
    const double sinv = (k >= 2) ? -d : d;
    z0 = (float)(rad * cosv);
    z1 = (float)(rad * sinv);
}
// The FLOAT32 form (observation noise: 13 of the 17 draws of a config-5 step, 26 when an episode ends): the same Box-Muller on the
// hardware transcendentals, ~25 instructions a pair instead of ~80.  Observation noise is added to a float32 output and feeds nothing
// back inside the environment, so an error of 1e-6 in z is 1e-8 in the observation; the draws that DO feed the dynamics keep the exact
// form above, because the action chain rounds in float32 and the near-cancelling rotor torques of a hovering drone turn a last-bit
// difference in one thrust into 1e-4 in the unit angular-velocity columns (seen when the action noise was tried in this form).
// Care is taken where float32 would lose the draw:
//   ln u1:  for u1 < 1/2 the float32 value of u1 carries it to 2^-24 relative; for u1 >= 1/2 it does not (u1 -> 1 rounds to 1 and the
//           radius to 0), so the COMPLEMENT w = 1 - u1 = (~word + 0.5) / 2^32 is formed from the integer (exact to 2^-24 relative)
//           and -ln(1 - w) taken as w (1 + w/2 + w^2/3) below w = 2^-9 (next term: 2e-9 relative) and as -ln2 log2(1 - w) above;
//   angle:  v_sin_f32 / v_cos_f32 take their argument in revolutions, i.e. u2 itself: no 2 pi product to round.
// Against the float64 libm evaluation of the same formula the draws differ by at most 1.2e-6 absolute (mean 7e-8; measured over 3.3e7 draws:
// tests/test_gpu_parity.py::test_observation_noise_draws_match_their_definition); sigma z enters the state at sigma <= 1e-2.
DN_DEV void box_muller_pair32(unsigned ra, unsigned rb, float &z0, float &z1)
{
    const bool upper = ra >= 0x80000000u;
    const float u1 = __builtin_fmaf((float)ra, 2.3283064365386963e-10f, 1.1641532182693481e-10f);          // (ra + 0.5) 2^-32
    const float w = __builtin_fmaf((float)(~ra), 2.3283064365386963e-10f, 1.1641532182693481e-10f);        // 1 - u1, exact to 2^-24 relative
    const float arg = upper ? 1.0f - w : u1;
    float L = -0.69314718055994530942f * __builtin_amdgcn_logf(arg);                                          // -ln(arg); v_log_f32 = log2
    const float series = w * __builtin_fmaf(w, __builtin_fmaf(w, 0.33333333333333333f, 0.5f), 1.0f);
    L = (upper && w < 0.001953125f) ? series : L;
    const float rad = __builtin_amdgcn_sqrtf(L + L);
    const float u2 = __builtin_fmaf((float)rb, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    z0 = rad * __builtin_amdgcn_cosf(u2);                                                                    // cos(2 pi u2)
    z1 = rad * __builtin_amdgcn_sinf(u2);
}
// Philox counter = (drone id lo, drone id hi, vector step lo, stream | vector step hi << 8): the 64-bit vector-step counter
// enters whole, so the streams do not repeat when its low word wraps (2^32 vector steps = hours of fused stepping);
// stream ids are below 256.
template <bool EXACT = true>
DN_DEV void noise4(unsigned long long seed, unsigned long long gid, unsigned long long step, unsigned stream, float z[4])
{
    unsigned r[4];
    philox4x32((unsigned)gid, (unsigned)(gid >> 32), (unsigned)step, stream | ((unsigned)(step >> 32) << 8), (unsigned)seed, (unsigned)(seed >> 32), r);
    if (EXACT) { box_muller_pair64(r[0], r[1], z[0], z[1]); box_muller_pair64(r[2], r[3], z[2], z[3]); }
    else { box_muller_pair32(r[0], r[1], z[0], z[1]); box_muller_pair32(r[2], r[3], z[2], z[3]); }
}
// The observation noise's draws: the float32 form, or -- DN_EXACT_OBS_NOISE=1, for runs that must replay bit for bit on another GPU generation
// or against the CPU definition -- the exact form the dynamics-feeding draws always use (one wave-uniform branch; ~3x the instructions a pair).
DN_DEV void obs_pair(const DnParams &p, const unsigned ra, const unsigned rb, float &z0, float &z1)
{
    if (p.exact_obs_noise) box_muller_pair64(ra, rb, z0, z1);
    else box_muller_pair32(ra, rb, z0, z1);
}
DN_DEV void obs_noise4(const DnParams &p, unsigned long long gid, unsigned long long step, unsigned stream, float z[4])
{
    unsigned r[4];
    philox4x32((unsigned)gid, (unsigned)(gid >> 32), (unsigned)step, stream | ((unsigned)(step >> 32) << 8), (unsigned)p.seed, (unsigned)(p.seed >> 32), r);
    obs_pair(p, r[0], r[1], z[0], z[1]);
    obs_pair(p, r[2], r[3], z[2], z[3]);
}
DN_DEV void add_act_noise(const DnParams &p, unsigned long long gid, unsigned long long step, float a[4])
{   // float32, unfused (its own function: the fused-multiply-add licence of step_body must not reach it)
    float z[4];
    noise4(p.seed, gid, step, 0u, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = p.act_noise_sigma * z[j];
        a[j] = clipv(a[j] + s, -1.0f, 1.0f);
    }
}
// add_obs_noise with the draws handed in: z[0..12] = the first 13 values of streams stream0 .. stream0 + 3 (the same float32 operations)
DN_DEV void add_obs_noise_drawn(const DnParams &p, const float z[DN_OBS_DIM], float o[DN_OBS_DIM])
{
#pragma unroll
    for (int j = 0; j < DN_OBS_DIM; ++j) {
        float s = p.obs_noise_sigma * z[j];
        o[j] = o[j] + s;
    }
}
DN_DEV void add_obs_noise(const DnParams &p, unsigned long long gid, unsigned long long step, unsigned stream0, float o[DN_OBS_DIM])
{
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float z[4];
        obs_noise4(p, gid, step, stream0 + (unsigned)b, z);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < DN_OBS_DIM) {
                float s = p.obs_noise_sigma * z[j];
                o[4 * b + j] = o[4 * b + j] + s;
            }
    }
}

// The reset observation's noise (streams stream0 .. stream0 + 3 of the drones whose episode ended) drawn ACROSS the wave: an episode
// ends on a few of a tile's 64 drones per step, yet add_obs_noise inside the lane-divergent reset branch costs the wave its seven
// Box-Muller pairs serially (~2 us of dn_step_squashed's tile time, on most steps of a short-episode fleet).  Here lane L draws pair
// L % 7 of the (L / 7)-th finished drone -- nine drones per pass -- and the draws travel through `scratch` (float[15][64] of LDS owned
// by this wave: 14 columns + the list of finished lanes).  Same counter, same key, same Box-Muller: the same bits as add_obs_noise.
// Wave-uniform: every lane calls it; z[0..12] is meaningful on the lanes of `done_mask`.
DN_DEV void draw_obs_noise_across(const DnParams &p, const unsigned long long gid_base, const unsigned long long step, const unsigned stream0,
                                  const unsigned long long done_mask, const bool mine, const unsigned lane, float *scratch, float z[DN_OBS_DIM])
{
    int *dlist = reinterpret_cast<int *>(scratch + 14 * DN_BLOCK);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(done_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)done_mask, 0u));
    if (mine) dlist[rank] = (int)lane;
    const unsigned count = (unsigned)__builtin_popcountll(done_mask);
    const unsigned w = lane / 7u, pr = lane - 7u * w;
    for (unsigned base = 0; base < count; base += 9u) {
        const bool work = w < 9u && base + w < count;
        const unsigned d = work ? (unsigned)dlist[base + w] : 0u;
        const unsigned long long gid = gid_base + d;
        unsigned r[4];
        philox4x32((unsigned)gid, (unsigned)(gid >> 32), (unsigned)step, (stream0 + (pr >> 1)) | ((unsigned)(step >> 32) << 8), (unsigned)p.seed,
                   (unsigned)(p.seed >> 32), r);
        float z0, z1;
        obs_pair(p, (pr & 1u) ? r[2] : r[0], (pr & 1u) ? r[3] : r[1], z0, z1);
        if (work) {
            scratch[(2u * pr) * DN_BLOCK + d] = z0;
            scratch[(2u * pr + 1u) * DN_BLOCK + d] = z1;
        }
    }
#pragma unroll
    for (int j = 0; j < DN_OBS_DIM; ++j) z[j] = scratch[j * DN_BLOCK + lane];
}

// ---- A10: normalize.NormalizeObservation with a batch of one (normalize.py:34-47, :94-97) ---------
// The statistics of one drone (13 means, 13 second moments, the count) live in registers for the whole launch: loaded
// once, updated by every observation the drone emits (step observations and reset observations, in that order),
// stored once.  The reference's update, with batch_count = 1 and batch_var = 0,
//     tot = count + 1;  new_mean = mean + delta / tot;  M2 = var count + delta^2 count / tot;  new_var = M2 / tot
// carries the SECOND MOMENT M2 = var count from step to step here (round 6; HBM holds it as well: dn_get_state / dn_set_state
// convert, var = M2 / count), because the update then needs no division at all and shares a product with the output:
//     r = 1 / tot (v_rcp_f64 + two Newton steps, shared by the 13 columns);  cw = count r
//     e = delta cw  (= x - new_mean);  new_mean = mean + delta r;  M2' = M2 + delta e;  new_var + 1e-8 = M2' r + 1e-8
// -- five float64 operations and one fused multiply-add for the radicand per column (the var-carrying form: seven), equal to
// the literal form to 1e-16.
struct Rms {
    double mean[DN_OBS_DIM], m2[DN_OBS_DIM], count;
};
DN_DEV void load_rms(const DnParams &p, long long i, Rms &r)
{
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) {
        r.mean[k] = p.st.rms_mean[(long long)k * p.n + i];
        r.m2[k] = p.st.rms_m2[(long long)k * p.n + i];
    }
    r.count = p.st.rms_count[i];
}
DN_DEV void store_rms(const DnParams &p, long long i, const Rms &r)
{
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) {
        p.st.rms_mean[(long long)k * p.n + i] = r.mean[k];
        p.st.rms_m2[(long long)k * p.n + i] = r.m2[k];
    }
    p.st.rms_count[i] = r.count;
}
DN_DEV double rcp_f64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
// Columns [K0, K1) only: a kernel may give the columns of one drone to two waves (each then holds its columns' statistics and a
// copy of the count).  The shared quantities (tot, inv, cw) and every per-column expression are the same whatever the range.
//
// The OUTPUT stage (round 6).  The statistics are float64 and updated exactly as above whatever the build; the normalised value
// leaves as a float32, and by default it is formed there: float32(x - new_mean) * v_rsq_f32(float32(new_var + 1e-8)) -- two conversions,
// one float32 transcendental and one float32 product instead of v_rsq_f64 + a Newton step + a float64 product + a conversion (37.6 -> 21 ns
// of a SIMD per column with the second-moment form above; the normaliser is a third of the fused step's vector-ALU time).  The result is within 3 float32 ulp (3.6e-7 relative) of
// the correctly rounded float64 evaluation, against the 1e-5 bar of the parity contract; nothing is fed back (the statistics never
// read the output).  -DDN_NORM_EXACT=1 keeps the float64 output stage -- the float32 nearest to the float64 evaluation (1/2 ulp) --
// and is what libdronenav_exact.so is built with (build.py; DN_EXACT_NORM=1 selects that library, dn_get_exact_flags reports it).
// A compile-time switch: as a run-time one (tried first) both forms sit in every normaliser pass and the five-wave kernel spills 159
// registers instead of 48.
#ifndef DN_NORM_EXACT
#define DN_NORM_EXACT 0
#endif
template <int K0, int K1>
DN_DEV void normalize_obs_cols(Rms &r, float o[DN_OBS_DIM])
{   // explicit fused multiply-adds, no contraction licence (one arithmetic sequence for every kernel that inlines this)
    const double tot = r.count + 1.0;
    const double inv = rcp_f64(tot);
    const double cw = r.count * inv;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const double x = (double)o[k];
        const double delta = x - r.mean[k];
        const double e = delta * cw;                                       // x - new_mean
        const double new_mean = __builtin_fma(delta, inv, r.mean[k]);
        const double new_m2 = __builtin_fma(delta, e, r.m2[k]);
        r.mean[k] = new_mean;
        r.m2[k] = new_m2;
        const double s = __builtin_fma(new_m2, inv, 1e-8);                 // new_var + 1e-8
#if DN_NORM_EXACT
        double y = __builtin_amdgcn_rsq(s);
        y = __builtin_fma(y, __builtin_fma(-(0.5 * s * y), y, 0.5), y);
        o[k] = (float)(e * y);
#else
        o[k] = (float)e * __builtin_amdgcn_rsqf((float)s);
#endif
    }
    r.count = tot;
}
DN_DEV void normalize_obs(Rms &r, float o[DN_OBS_DIM]) { normalize_obs_cols<0, DN_OBS_DIM>(r, o); }
// the statistics of columns [K0, K1) of the tile's drones: uniform column bases (SGPR pairs) + the lane's 32-bit offset
// The same loads / stores with the column base WALKED in one scalar register pair (base += n per column, opaque to the optimiser): 27 uniform
// bases computed up front are 54 SGPRs the single-step kernel does not have -- it then forms 27 per-lane 64-bit addresses (54 VGPRs) and
// keeps them from the loads to the stores.  Walked, load and store each recompute their base from one pair and share the lane's 32-bit offset.
DN_DEV void load_rms_walk(const DnParams &p, const long long tile_base, const unsigned li, Rms &r)
{
    const double *pm = p.st.rms_mean + tile_base, *pv = p.st.rms_m2 + tile_base;
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) {
        r.mean[k] = pm[li];
        r.m2[k] = pv[li];
        pm += p.n; pv += p.n;
        asm volatile("" : "+s"(pm), "+s"(pv));
    }
    r.count = (p.st.rms_count + tile_base)[li];
}
DN_DEV void store_rms_walk(const DnParams &p, const long long tile_base, const unsigned li, const Rms &r)
{
    double *pm = p.st.rms_mean + tile_base, *pv = p.st.rms_m2 + tile_base;
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) {
        pm[li] = r.mean[k];
        pv[li] = r.m2[k];
        pm += p.n; pv += p.n;
        asm volatile("" : "+s"(pm), "+s"(pv));
    }
    (p.st.rms_count + tile_base)[li] = r.count;
}
template <int K0, int K1>
DN_DEV void load_rms_cols(const DnParams &p, const long long tile_base, const unsigned li, Rms &r)
{
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        r.mean[k] = (p.st.rms_mean + ((long long)k * p.n + tile_base))[li];
        r.m2[k] = (p.st.rms_m2 + ((long long)k * p.n + tile_base))[li];
    }
    r.count = (p.st.rms_count + tile_base)[li];
}
template <int K0, int K1, bool COUNT>
DN_DEV void store_rms_cols(const DnParams &p, const long long tile_base, const unsigned li, const Rms &r)
{
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        (p.st.rms_mean + ((long long)k * p.n + tile_base))[li] = r.mean[k];
        (p.st.rms_m2 + ((long long)k * p.n + tile_base))[li] = r.m2[k];
    }
    if (COUNT) (p.st.rms_count + tile_base)[li] = r.count;
}

// ---- waypoint/corridor table in LDS ------------------------------------------------------------
template <typename R> DN_DEV const R *table_ptr(const DnParams &p);
template <> DN_DEV const double *table_ptr<double>(const DnParams &p) { return p.tab64; }
template <> DN_DEV const float *table_ptr<float>(const DnParams &p) { return p.tab32; }
template <typename R> DN_DEV const DnConsts<R> &consts(const DnParams &p);
template <> DN_DEV const DnConsts<double> &consts<double>(const DnParams &p) { return p.c64; }
template <> DN_DEV const DnConsts<float> &consts<float>(const DnParams &p) { return p.c32; }

// ---- arithmetic helpers ---------------------------------------------------------------------------
// The float32 action chain above is evaluated operation by operation (file-wide -ffp-contract=off) because it
// is bit-exact against the reference's numpy float32.  Everything from here on is float64 in registers over a
// float32 state and is held to 1e-5 (flags exact); there a multiply-add may fuse (the difference is one float64
// rounding, 1e-16) and divide / sqrt are replaced by the hardware seed (v_rcp_f64 / v_rsq_f64, ~2^-23) plus two
// Newton steps: the operands are in benign ranges (no denormals, no overflow), so the scaling and fix-up code of
// the IEEE library routines (3x the instructions) buys nothing.  The kernel is instruction-issue bound at
// 32768 drones (one wave per SIMD), so the instruction count IS the step time.
template <typename R> struct FM;
template <> struct FM<double> {
    static DN_DEV double rcp(double x)
    {
        double r = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(r, e, r);
    }
    static DN_DEV double rsq(double x)         // x > 0; x == 0 gives NaN (callers compare / select afterwards)
    {
        double y = __builtin_amdgcn_rsq(x);
        const double hx = 0.5 * x;
        double e = __builtin_fma(-(hx * y), y, 0.5);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-(hx * y), y, 0.5);
        return __builtin_fma(y, e, y);
    }
    static DN_DEV double rsq_f32grade(double x)   // one Newton step: ~2^-45, for results that leave as float32
    {
        double y = __builtin_amdgcn_rsq(x);
        double e = __builtin_fma(-(0.5 * x * y), y, 0.5);
        return __builtin_fma(y, e, y);
    }
    static DN_DEV double rcp_f32grade(double x)   // one Newton step on the 2^-23 seed: ~2^-46, for results that leave as float32
    {
        double r = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(r, e, r);
    }
    // x >= 0, exact 0 allowed.  One coupled Newton step on the 2^-23 seed (g ~ sqrt x, h ~ 1 / (2 sqrt x): 2^-45) and the residual
    // correction g += (x - g^2) h, which squares that again: the double nearest to the root or its neighbour (round 6; a second
    // coupled step in front of the correction bought nothing but three instructions).
    static DN_DEV double sqrt0(double x)
    {
        double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = 0.5 * y;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        const double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        return x == 0.0 ? 0.0 : g;
    }
    static DN_DEV double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
};
template <> struct FM<float> {                  // speed option (compute_f32): hardware 1-ulp approximations
    static DN_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static DN_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
    static DN_DEV float rsq_f32grade(float x) { return __builtin_amdgcn_rsqf(x); }
    static DN_DEV float rcp_f32grade(float x) { return __builtin_amdgcn_rcpf(x); }
    static DN_DEV float sqrt0(float x) { return __builtin_amdgcn_sqrtf(x); }
    static DN_DEV float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};

// atan2 in float32 for the Euler angles (observation columns 3..5 only, bar 1e-5 on angle/pi): the Cephes atanf scheme
// -- two range reductions at tan(pi/8) and tan(3pi/8), a degree-4 polynomial in z^2 (|error| < 2e-7 rad) -- on the
// hardware reciprocal, ~25 instructions against ~52 for the library atan2f with its special-case handling.  Inputs
// here are finite and not both zero except at exactly zero attitude, which returns 0 like atan2(0, 1).
DN_DEV float atan_pos32(float t)                       // t >= 0 (may be +inf)
{
    const bool big = t > 2.414213562373095f, mid = t > 0.4142135623730950f;
    const float y0 = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
    const float num = big ? -1.0f : (mid ? t - 1.0f : t);
    const float den = big ? t : (mid ? t + 1.0f : 1.0f);
    const float x = num * __builtin_amdgcn_rcpf(den);
    const float z = x * x;
    float pz = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    pz = __builtin_fmaf(pz, z, 1.99777106478e-1f);
    pz = __builtin_fmaf(pz, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(pz * z, x, x);
}
DN_DEV float atan2_fast32(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float r = atan_pos32(ay * __builtin_amdgcn_rcpf(ax));       // ax = 0 -> +inf -> pi/2
    r = (ax == 0.0f && ay == 0.0f) ? 0.0f : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return __builtin_copysignf(r, y);
}

// ---- A8: _has_collision_occurred (PBDroneEnv.py:678-707) + is_out_of_cylinder_bounds (:718-786) ----
// The reference compares distances, norm(.) > radius; here the squared distance is compared with the squared
// radius (no sqrt).  The two predicates differ only when the squared distance is within an ulp of the squared
// radius (probability ~1e-16 per test); NaN compares false on both forms.
template <typename R>
DN_DEV bool collision_common(const DnParams &p, const DnConsts<R> &c, R px, R py, R pz, R r22)
{   // everything in _has_collision_occurred that does not depend on the waypoint index
    // (explicit fused multiply-adds, no contraction licence: see physics_phase)
    bool out = px > c.dim[3] || px < c.dim[0] || py > c.dim[4] || py < c.dim[1] || pz > c.dim[5];
    if (p.ground_contact) {
        // len(p.getContactPoints()) > 0 against plane.urdf, APPROXIMATED [3P-recall]: lowest point of the
        // collision cylinder within Bullet's 0.02 contact-breaking threshold of z = 0:
        //   pz - (H/2 |r22| + R sqrt(1 - r22^2)) <= 0.02   <=>   m <= 0  or  m^2 <= R^2 (1 - r22^2),
        //   m = pz - H/2 |r22| - 0.02
        R s2 = FM<R>::fma(-r22, r22, R(1.0));
        s2 = s2 > R(0.0) ? s2 : R(0.0);
        const R m = FM<R>::fma(R(-0.5) * K<R>::COLL_H, fabs(r22), pz) - R(0.02);
        out = out || m <= R(0.0) || m * m <= (K<R>::COLL_R * K<R>::COLL_R) * s2;
    }
    if (p.cylinder && p.circle) {                     // :723-741, centre (0,0,1), radius 1
        const R rn = FM<R>::rsq(FM<R>::fma(py, py, px * px));   // 0 -> NaN -> the compare below is false, as in the reference
        const R ex = FM<R>::fma(-px, rn, px), ey = FM<R>::fma(-py, rn, py), ez = pz - R(1.0);
        out = out || FM<R>::fma(ez, ez, FM<R>::fma(ey, ey, ex * ex)) > c.thr2;
    }
    return out;
}

// The table row of a drone's CURRENT waypoint, read at the top of a step (the index is part of the entry state) so
// that the LDS round trip hides behind the physics instead of stalling the rules that need it.
template <typename R> struct GateRow {
    R wp[3], u[3], e1[3], lext, ll;
};
template <typename R> DN_DEV GateRow<R> load_gate_row(const R *tab, int idx)
{
    const R *e = tab + idx * DN_T_STRIDE;
    GateRow<R> g;
#pragma unroll
    for (int j = 0; j < 3; ++j) { g.wp[j] = e[DN_T_WP + j]; g.u[j] = e[DN_T_U + j]; g.e1[j] = e[DN_T_E1 + j]; }
    g.lext = e[DN_T_LEXT]; g.ll = e[DN_T_LL];
    return g;
}
template <typename R>
DN_DEV bool outside_segment_corridor_row(const DnConsts<R> &c, const GateRow<R> &g, R px, R py, R pz)
{   // :746-786 on a table row in registers (explicit fused multiply-adds, no contraction licence: see physics_phase)
    const R ux = g.u[0], uy = g.u[1], uz = g.u[2];
    const R dx = px - g.e1[0], dy = py - g.e1[1], dz = pz - g.e1[2];                       // :776
    R proj = FM<R>::fma(dz, uz, FM<R>::fma(dy, uy, dx * ux));                              // :778
    proj = clipv(proj, R(0.0), g.lext);                                                     // :780
    // distance to the clamped projection (:782-786); a zero-length segment has u = 0, e1 = base1, lext = 0, so the
    // same expression is |pos - base1|, which the reference tests against the bare threshold (:756-757)
    const R qx = FM<R>::fma(-proj, ux, dx), qy = FM<R>::fma(-proj, uy, dy), qz = FM<R>::fma(-proj, uz, dz);
    const R lim = g.ll == R(0.0) ? c.thr2 : c.thr_ext2;
    return FM<R>::fma(qz, qz, FM<R>::fma(qy, qy, qx * qx)) > lim;
}

template <typename R>
DN_DEV bool outside_segment_corridor(const DnConsts<R> &c, const R *tab, R px, R py, R pz, int idx)
{   // :746-786, the corridor around the segment that ends at waypoint idx (non-circle tracks)
    return outside_segment_corridor_row<R>(c, load_gate_row<R>(tab, idx), px, py, pz);
}

// orientation_reward (PBDroneEnv.py:573-586) with get_forward_vector (:588-597).  The reference tests
// arccos(clip(f . t, -1, 1)) > radians(10); arccos is strictly decreasing, so that is f . t < cos(10 deg)
// (the clip cannot change the outcome because -1 < cos 10 deg < 1; NaN compares false on both forms).
template <typename R>
DN_DEV int orientation_reward(R fx, R fy, R fz, R px, R py, R pz, const R *wp)
{
    const R tx = wp[0] - px, ty = wp[1] - py, tz = wp[2] - pz;
    // f . t / |t| < cos(10 deg), without the root: cos(10 deg) > 0, so the test holds iff f . t < 0 or (f . t)^2 < cos^2 |t|^2.
    // (|t| = 0: the reference divides 0 by 0, NaN compares false -> 0; here 0 < 0 is false as well.)
    const R t2 = FM<R>::fma(tz, tz, FM<R>::fma(ty, ty, tx * tx));                  // explicit order, see attitude_phase
    const R dot = FM<R>::fma(fz, tz, FM<R>::fma(fy, ty, fx * tx));
    return (dot < R(0.0) || dot * dot < (K<R>::COS_10DEG * K<R>::COS_10DEG) * t2) ? -1 : 0;
}

// sin(h)/h and cos(h) for the quaternion half-angle h = |w| dt / 2 <= pi/8 (Bullet clamps |w| dt at pi/4):
// Taylor polynomials in h^2 (Horner, fused), no range reduction needed.  Truncation < 3e-15 at the clamp (h^2 = 0.154)
// and < 1e-21 below |w| = 60 rad/s: nine orders of magnitude inside the float32 the attitude is stored in.
template <typename R> DN_DEV void sinc_cos_small(R h2, R &sinc, R &c)
{
    R s = R(-1.0 / 39916800.0);
    s = FM<R>::fma(s, h2, R(1.0 / 362880.0));
    s = FM<R>::fma(s, h2, R(-1.0 / 5040.0));
    s = FM<R>::fma(s, h2, R(1.0 / 120.0));
    s = FM<R>::fma(s, h2, R(-1.0 / 6.0));
    sinc = FM<R>::fma(s, h2, R(1.0));
    R k = R(1.0 / 479001600.0);
    k = FM<R>::fma(k, h2, R(-1.0 / 3628800.0));
    k = FM<R>::fma(k, h2, R(1.0 / 40320.0));
    k = FM<R>::fma(k, h2, R(-1.0 / 720.0));
    k = FM<R>::fma(k, h2, R(1.0 / 24.0));
    k = FM<R>::fma(k, h2, R(-0.5));
    c = FM<R>::fma(k, h2, R(1.0));
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

// Block-relative output pointers of one control step.  Every pointer is uniform over the wave (an SGPR pair), so a
// lane's access is "scalar base + 32-bit lane offset" (the saddr form of global_load/store) and the 64-bit
// address arithmetic is done once per wave on the scalar unit instead of once per lane per access.
struct StepOut {
    float *obs;                    // [64][13] tile of this workgroup
    float *reward;
    uint8_t *done, *truncated;
    int32_t *found;
    float *terminal_obs;           // or nullptr
    float *ep_return;              // or nullptr
    int32_t *ep_length;            // or nullptr
    unsigned long long *done_word; // this workgroup's ballot word, or nullptr
};
DN_DEV StepOut block_out(const DnStepIO &io, long long tile_base, long long step_off, long long word_off)
{
    StepOut o;
    const long long e = step_off + tile_base;
    o.obs = io.obs + e * DN_OBS_DIM;
    o.reward = io.reward + e;
    o.done = io.done + e;
    o.truncated = io.truncated + e;
    o.found = io.found_targets + e;
    o.terminal_obs = io.terminal_obs ? io.terminal_obs + e * DN_OBS_DIM : nullptr;
    o.ep_return = io.ep_return ? io.ep_return + e : nullptr;
    o.ep_length = io.ep_length ? io.ep_length + e : nullptr;
    o.done_word = io.done_mask ? io.done_mask + word_off + tile_base / DN_BLOCK : nullptr;     // the tile's ballot word (not blockIdx.x: a fused launch steps several tiles per workgroup)
    return o;
}

// "Evaluate this here": the compiler is free to sink pure arithmetic below a workgroup barrier (nothing orders ALU work
// against s_barrier), which would move work a wave is meant to do WHILE it waits to after the wait.  An empty asm that
// reads and writes the value pins it to this point of the program.
DN_DEV void pin(double &x) { asm volatile("" : "+v"(x)); }
DN_DEV void pin(float &x) { asm volatile("" : "+v"(x)); }
DN_DEV void pin(bool &x) { int t = x; asm volatile("" : "+v"(t)); x = t != 0; }

// LDS written by some lanes of a wave and read by other lanes of the SAME wave: order the accesses without a
// workgroup barrier (the observation tile belongs to one wave).
DN_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0) only: LDS traffic, not the outstanding global stores
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Workgroup barrier that orders LDS only.  __syncthreads() would also drain the global stores in flight
// (vmcnt(0)), i.e. stall every step on the HBM round trip of the previous step's outputs.
DN_DEV void block_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The wave's [64,13] observation tile goes out through LDS: lanes park their 13 floats (stride 13 dwords: odd, so
// conflict-free), then the wave streams the 3328 contiguous bytes out as float4 (ds_read_b128 +
// global_store_dwordx4), i.e. 4 store instructions instead of 13 strided dword stores per destination.
// Split in two so that the fused two-wave kernel can park step t's tile at the end of report(t) and stream it out
// at the start of report(t+1), with the LDS round trip hidden behind a whole step of other work; LDS operations of
// one wave execute in order, so the later reads see the earlier writes without a wait in between.
DN_DEV void tile_park(float *s_tile, unsigned lane, const float o[DN_OBS_DIM])
{
    asm volatile("" ::: "memory");                        // compiler-level ordering only: no s_waitcnt
#pragma unroll
    for (int k = 0; k < DN_OBS_DIM; ++k) s_tile[lane * DN_OBS_DIM + k] = o[k];
    asm volatile("" ::: "memory");
}
struct TileRegs {
    float4 r[4];
};
DN_DEV TileRegs tile_fetch(const float *s_tile, unsigned lane)
{
    asm volatile("" ::: "memory");
    const float4 *s4 = reinterpret_cast<const float4 *>(s_tile);
    TileRegs t;
#pragma unroll
    for (int r = 0; r < 3; ++r) t.r[r] = s4[r * 64 + lane];
    t.r[3] = s4[192 + (lane < (DN_BLOCK * DN_OBS_DIM / 4 - 192) ? lane : 0)];
    asm volatile("" ::: "memory");
    return t;
}
DN_DEV void tile_stream(const TileRegs &t, const float *s_tile, float *gtile, unsigned rows, unsigned lane)
{
    if (rows == DN_BLOCK) {
        float4 *g4 = reinterpret_cast<float4 *>(gtile);
#pragma unroll
        for (int r = 0; r < 3; ++r) g4[r * 64 + lane] = t.r[r];
        if (lane < (DN_BLOCK * DN_OBS_DIM / 4 - 192)) g4[192 + lane] = t.r[3];
    } else {
        const unsigned rem = rows * DN_OBS_DIM;            // ragged last tile
        for (unsigned e = lane; e < rem; e += DN_BLOCK) gtile[e] = s_tile[e];
    }
}
DN_DEV void store_obs_tile(float *s_tile, float *gtile, unsigned rows, unsigned lane, const float o[DN_OBS_DIM])
{   // park + stream back to back (single-step launches, the one-wave kernels, reset)
    tile_park(s_tile, lane, o);
    const TileRegs t = tile_fetch(s_tile, lane);
    tile_stream(t, s_tile, gtile, rows, lane);
}

template <typename R>
DN_DEV void stage_table(const DnParams &p, R *s_tab)
{   // all threads of the workgroup cooperate; the caller's next block barrier publishes the table
    const R *g = table_ptr<R>(p);
    for (unsigned j = threadIdx.x; j < (unsigned)(p.num_waypoints * DN_T_STRIDE); j += blockDim.x) s_tab[j] = g[j];
}
template <typename R>
DN_DEV void stage_table_by(const DnParams &p, R *s_tab, const unsigned tid, const unsigned nthreads)
{   // the first `nthreads` threads of the workgroup stage it
    const R *g = table_ptr<R>(p);
    for (unsigned j = tid; j < (unsigned)(p.num_waypoints * DN_T_STRIDE); j += nthreads) s_tab[j] = g[j];
}

// Observation of a body that has just been (re)loaded at the spawn pose (BaseAviary.reset -> _computeObs,
// BaseAviary.py:318) with the not-yet-reset stale distance d_last (quirk Q2).
template <typename R>
DN_DEV void reset_obs(const DnParams &p, const DnConsts<R> &c, R d_last, float o[DN_OBS_DIM])
{
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = c.reset_obs32[k];
    o[12] = p.include_distance ? (float)(d_last * c.inv_max_target_dist) : 0.0f;
}

// =====================================================================================================
// The step, in four phases.
// =====================================================================================================
// One control step of one drone (one lane) is cut where its data dependencies allow two wavefronts to work on it
// at the same time:
//
//     flight wave (the recurrence)                               report wave (side outputs, one step behind)
//     ----------------------------                               -------------------------------------------
//     thrust_phase   A1-A3: action -> rotor forces
//     physics_phase  A4: Bullet step -> post-physics state  --Flight-->  observe_phase  A5-A7: Euler, observation,
//     rules_phase    A8-A9: collision, gate logic,                                      reward candidates
//                    truncation, post-step distance,        --Verdict->  report_phase   A7 select, A10-A11: Monitor,
//                    auto-reset of the body                              terminal / reset observation, statistics, outputs
//
// The flight wave owns the state the dynamics and the rules feed on (G0-G3, G6); the report wave owns what only
// the outputs feed on (G4 prev_vel/episode return, G5 prev_ang_v/episode length, normaliser statistics).  At
// 32 768 drones the chip holds one wave per SIMD on half its SIMDs and the step is instruction-issue bound, so
// putting the two halves of a step on two SIMDs shortens the step itself; the structs below are what crosses
// from one wave to the other (through LDS in the two-wave kernels, in registers in the one-wave kernels -- the
// values are float64 either way, so both kernel shapes produce identical bits).
//
// NORM / NOISE compile the optional per-drone observation normaliser and the Philox noise streams in or out.
//
// Algebra used to shorten Bullet's free-base step (A4) without changing what it computes beyond float64 rounding:
//   * linear part: Bullet forms F_b = f_thrust_b + R^T (0,0,-M G) - M v_b (c + c |v_b|) in the body frame and
//     rotates a_b = F_b / M back.  R R^T = 1 and |v_b| = |v|, so a = R[:,2] fz / M - (0,0,G) - v (c + c |v|).
//   * the exponential-map quaternion update needs |w| only through (|w| dt / 2)^2 (sinc/cos are even), so no sqrt.
//   * get_forward_vector (PBDroneEnv.py:588-597) = (cos yaw cos pitch, sin yaw cos pitch, sin pitch) is the first
//     column of the rotation matrix of the (just normalised) quaternion: (1 - 2(y^2+z^2), 2(xy+wz), -2(xz-wy)).
struct Thrust {
    float f[4];        // rotor forces along body z (newton)
    float zt;          // yaw torque
};
template <typename R> struct Flight {
    R px, py, pz, qx, qy, qz, qw;                            // post-physics pose (before any reset): feeds compares, stays R
    R fwx, fwy, fwz;                                         // get_forward_vector of that pose (attitude_phase): feeds a compare, stays R
    float roll_num32, roll_den32, pitch32, yaw32;            // p.getEulerFromQuaternion of that pose (observation columns only);
                                                             // the roll's atan2 is left to the report wave, which has the slack
    float vx, vy, vz, wx, wy, wz;                            // post-physics velocities as they go back to HBM (float32):
                                                             // the report wave only turns them into observation columns
    float vex, vey, vez, aex, aey, aez;                      // entry velocities = current_vel / current_ang_v (quirk Q4), float32 state
    float d_e, dprev_e;                                      // entry _distance_to_target / _prev_distance_to_target, float32 state
    int idx_e, just_found_e, truncated;                      // entry index / flag; _computeTruncated (entry _steps)
};
constexpr int DN_NMAIL64 = 7;                                // R-valued words that cross to the report wave: position, forward vector, Verdict.d_obs
template <typename R> struct Verdict {
    R d_obs;           // the distance the reset observation shows (quirk Q2)
    int coll1;         // _computeTerminated inside _computeReward (entry index)
    int terminated;    // _computeTerminated of BaseAviary.step (advanced index, _is_done)
};

// ---- N4: ActionType.PID / VEL / ONE_D_PID -- BaseSingleAgentAviary._preprocessAction (BaseSingleAgentAviary.py:180-222)
// around DSLPIDControl.computeControl (Sol/PyBullet/DSLPIDControl.py:78-262), float64 as numpy evaluates it, one controller per
// drone.  st = integral_pos_e[3], last_rpy[3], integral_rpy_e[3]: resident in registers for the launch, never reset (the
// reference calls ctrl.reset() only from DSLPIDControl.__init__).  The controller reads the ENTRY state of the step
// (_getDroneStateVector before p.stepSimulation), so these action types run on the one-wave kernels, where the thrust is
// computed in the step it belongs to.  Unreachable in the reference (PBDroneEnv overrides _preprocessAction); the libm
// calls are left literal.
struct PidCtx {
    float4 G0, G1, G2;     // entry pos, quat, vel
    double *st;            // 9 doubles, in registers
};
DN_DEV void euler_from_quat64(const double q[4], double rpy[3])
{   // p.getEulerFromQuaternion [3P-recall of pybullet.c]
    const double sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], squ = q[3] * q[3];
    const double sarg = -2.0 * (q[0] * q[2] - q[3] * q[1]);
    if (sarg <= -0.99999) { rpy[0] = 0.0; rpy[1] = -0.5 * 3.14159265358979323846; rpy[2] = 2.0 * atan2(q[0], -q[1]); }
    else if (sarg >= 0.99999) { rpy[0] = 0.0; rpy[1] = 0.5 * 3.14159265358979323846; rpy[2] = 2.0 * atan2(-q[0], q[1]); }
    else {
        rpy[0] = atan2(2.0 * (q[1] * q[2] + q[3] * q[0]), squ - sqx - sqy + sqz);
        rpy[1] = asin(sarg);
        rpy[2] = atan2(2.0 * (q[0] * q[1] + q[3] * q[2]), squ + sqx - sqy - sqz);
    }
}
DN_DEV void pid_control64(const int mode, const PidCtx &cx, const float cmd[4], double rpm[4])
{
    const double dt = 1.0 / 240.0;                             // CTRL_TIMESTEP
    const double pos[3] = {cx.G0.x, cx.G0.y, cx.G0.z}, quat[4] = {cx.G1.x, cx.G1.y, cx.G1.z, cx.G1.w}, vel[3] = {cx.G2.x, cx.G2.y, cx.G2.z};
    double *st = cx.st;
    double target_pos[3], target_vel[3] = {0.0, 0.0, 0.0}, target_yaw = 0.0, rpy[3];
    euler_from_quat64(quat, rpy);
    if (mode == 2) {                                           // PID: _calculateNextStep(pos, action, 1), BaseAviary.py:1255-1298
        const double dir[3] = {(double)cmd[0] - pos[0], (double)cmd[1] - pos[1], (double)cmd[2] - pos[2]};
        const double dist = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) target_pos[k] = dist <= 1.0 ? (double)cmd[k] : pos[k] + dir[k] / dist * 1.0;
    } else if (mode == 3) {                                    // VEL (:195-210): float32 arithmetic on the action array
        float n2 = cmd[0] * cmd[0];
        n2 = n2 + cmd[1] * cmd[1];
        n2 = n2 + cmd[2] * cmd[2];
        const float n = sqrtf(n2);
        const float lim = 0.25f * fabsf(cmd[3]);               // SPEED_LIMIT = 0.03 * MAX_SPEED_KMH * (1000 / 3600) = 0.25
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float u = n != 0.0f ? cmd[k] / n : 0.0f;
            target_vel[k] = (double)(lim * u);
            target_pos[k] = pos[k];
        }
        target_yaw = rpy[2];                                   // target_rpy = (0, 0, state[9])
    } else {                                                   // ONE_D_PID (:213-221)
        target_pos[0] = pos[0] + 0.1 * 0.0; target_pos[1] = pos[1] + 0.1 * 0.0;
        target_pos[2] = pos[2] + 0.1 * (double)cmd[0];
    }
    // p.getMatrixFromQuaternion (btMatrix3x3::setRotation)
    double R[9];
    {
        const double x = quat[0], y = quat[1], z = quat[2], w = quat[3];
        const double d = x * x + y * y + z * z + w * w, sc = 2.0 / d;
        const double xs = x * sc, ys = y * sc, zs = z * sc, wx = w * xs, wy = w * ys, wz = w * zs;
        const double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
        R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
        R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
        R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
    }
    // _dslPIDPositionControl, DSLPIDControl.py:140-199
    const double Pf[3] = {.4, .4, 1.25}, If[3] = {.05, .05, .05}, Df[3] = {.2, .2, .5};
    double pos_e[3], vel_e[3], tt[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pos_e[k] = target_pos[k] - pos[k];
        vel_e[k] = target_vel[k] - vel[k];
        st[k] = clipv(st[k] + pos_e[k] * dt, -2.0, 2.0);
    }
    st[2] = clipv(st[2], -0.15, 0.15);
#pragma unroll
    for (int k = 0; k < 3; ++k) tt[k] = Pf[k] * pos_e[k] + If[k] * st[k] + Df[k] * vel_e[k] + (k == 2 ? 9.8 * 0.027 : 0.0);
    const double dotz = tt[0] * R[2] + tt[1] * R[5] + tt[2] * R[8];
    const double scalar_thrust = dotz > 0.0 ? dotz : 0.0;
    const double thrust = (sqrt(scalar_thrust / (4 * 3.16e-10)) - 4070.3) / 0.2685;
    const double nt = sqrt(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]);
    const double z_ax[3] = {tt[0] / nt, tt[1] / nt, tt[2] / nt};
    const double x_c[3] = {cos(target_yaw), sin(target_yaw), 0.0};
    double y_ax[3] = {z_ax[1] * x_c[2] - z_ax[2] * x_c[1], z_ax[2] * x_c[0] - z_ax[0] * x_c[2], z_ax[0] * x_c[1] - z_ax[1] * x_c[0]};
    const double ny = sqrt(y_ax[0] * y_ax[0] + y_ax[1] * y_ax[1] + y_ax[2] * y_ax[2]);
    y_ax[0] /= ny; y_ax[1] /= ny; y_ax[2] /= ny;
    const double x_ax[3] = {y_ax[1] * z_ax[2] - y_ax[2] * z_ax[1], y_ax[2] * z_ax[0] - y_ax[0] * z_ax[2], y_ax[0] * z_ax[1] - y_ax[1] * z_ax[0]};
    // target_rotation = [x_ax y_ax z_ax] (columns); the reference's scipy round trip (as_euler 'XYZ' -> from_euler -> as_quat ->
    // from_quat -> as_matrix, :196, :236-238) returns the same rotation to rounding
    const double Rt[9] = {x_ax[0], y_ax[0], z_ax[0], x_ax[1], y_ax[1], z_ax[1], x_ax[2], y_ax[2], z_ax[2]};
    // _dslPIDAttitudeControl, :203-262: rot_matrix_e = Rt^T Rc - Rc^T Rt, rot_e = (e[2,1], e[0,2], e[1,0])
    double A[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[3 * i + j] = Rt[i] * R[j] + Rt[3 + i] * R[3 + j] + Rt[6 + i] * R[6 + j];
    const double rot_e[3] = {A[7] - A[5], A[2] - A[6], A[3] - A[1]};
    const double Pt[3] = {70000., 70000., 60000.}, It[3] = {.0, .0, 500.}, Dt[3] = {20000., 20000., 12000.};
    double tq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double rate_e = 0.0 - (rpy[k] - st[3 + k]) / dt;
        st[3 + k] = rpy[k];
        st[6 + k] = clipv(st[6 + k] - rot_e[k] * dt, -1500.0, 1500.0);
        if (k < 2) st[6 + k] = clipv(st[6 + k], -1.0, 1.0);
        tq[k] = clipv(-(Pt[k] * rot_e[k]) + Dt[k] * rate_e + It[k] * st[6 + k], -3200.0, 3200.0);
    }
    const double MIX[4][3] = {{-.5, -.5, -1}, {-.5, .5, 1}, {.5, .5, -1}, {.5, -.5, 1}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double pwm = thrust + (MIX[i][0] * tq[0] + MIX[i][1] * tq[1] + MIX[i][2] * tq[2]);
        pwm = clipv(pwm, 20000.0, 65535.0);
        rpm[i] = 0.2685 * pwm + 4070.3;
    }
}

struct ThrustX {       // XOPT kernels: float64 carriers (ActionType.RPM works in float64) + the rpm for the extra terms
    double f[4];
    double zt;
};
template <bool NOISE>
DN_DEV ThrustX thrust_phase_x(const DnParams &p, unsigned long long gid, unsigned long long step_count, const float4 A, Extras &x,
                              const PidCtx *pid = nullptr)
{
    float a[4] = {A.x, A.y, A.z, A.w};
    if (NOISE && p.act_noise_sigma > 0.0f) add_act_noise(p, gid, step_count, a);
    ThrustX t;
    x.gnd = p.gnd; x.drag = p.drag; x.rpm_f32 = !p.rpm_actions && !p.pid_mode;
    x.damp = p.zero_damping ? 0.0 : 0.04;
    if (pid && p.pid_mode) {   // ActionType.PID / VEL / ONE_D_PID: the DSLPIDControl loop on the entry state
        float cmd[4];
        double rpm[4], tq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cmd[j] = p.normalize_actions ? rescale_action32(a[j]) : a[j];   // PBDroneEnv.step, :173-176
        pid_control64(p.pid_mode, *pid, cmd, rpm);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double sq = rpm[j] * rpm[j];              // BaseAviary._physics on a float64 rpm array, BaseAviary.py:776-780
            x.rpm[j] = rpm[j];
            t.f[j] = sq * 3.16e-10;
            tq[j] = sq * 7.94e-12;
        }
        double z = -tq[0];
        z = z + tq[1];
        z = z - tq[2];
        t.zt = z + tq[3];
    } else if (p.rpm_actions) {   // BaseSingleAgentAviary._preprocessAction, ActionType.RPM / ONE_D_RPM (BaseSingleAgentAviary.py:176-179, :211-212)
        double tq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float aj = p.rpm_actions == 2 ? a[0] : a[j];                       // ONE_D_RPM: np.repeat(.., 4)
            const float cmd = p.normalize_actions ? rescale_action32(aj) : aj;       // PBDroneEnv.step, :173-176
            const float s = 0.05f * cmd;
            const float u = 1.0f + s;
            const double rpm = HOVER_RPM * (double)u;       // np.float64 scalar (x) float32 array
            const double sq = rpm * rpm;                    // BaseAviary._physics, BaseAviary.py:776-777
            x.rpm[j] = rpm;
            t.f[j] = sq * 3.16e-10;
            tq[j] = sq * 7.94e-12;
        }
        double z = -tq[0];
        z = z + tq[1];
        z = z - tq[2];
        t.zt = z + tq[3];
    } else {
        float tq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float rpm;
            t.f[j] = (double)rotor_force_from_action(a[j], p.normalize_actions != 0, tq[j], &rpm);
            x.rpm[j] = (double)rpm;
        }
        t.zt = (double)z_torque32(tq);
    }
    return t;
}
template <bool NOISE>
DN_DEV Thrust thrust_phase(const DnParams &p, unsigned long long gid, unsigned long long step_count, const float4 A)
{   // float32, unfused: bit-exact numpy
    float a[4] = {A.x, A.y, A.z, A.w};
    if (NOISE && p.act_noise_sigma > 0.0f) add_act_noise(p, gid, step_count, a);
    Thrust t;
    float tq[4];
    const bool norm_act = p.normalize_actions != 0;
#ifdef DN_NO_SAT_FASTPATH                              // ablation switch (profiles/r06_notes.md): the chain for every rotor of every lane
#pragma unroll
    for (int j = 0; j < 4; ++j) t.f[j] = rotor_force_from_action(a[j], norm_act, tq[j], nullptr, true);
#else
    // Per rotor: two selects, and the chain only if a lane of the wave is inside the band.  The four chains sit in four branches and
    // cannot interleave, so where EVERY rotor of the wave needs its chain (a policy in the hover band: always; U(-1,1): 0.2 % of
    // wave-steps) they are evaluated side by side as before -- four independent dependency chains for the scheduler to overlap.  That
    // test is made only once rotor 0 is known to need its chain (U(-1,1): one wave-step in five), so the saturated regime does not pay
    // for it on the role every barrier waits for (measured: the four ballots up front cost the U(-1,1) launch 2 %).
    const float t_lo = norm_act ? ACT_SAT_LO32 : A_LOW32, t_hi = norm_act ? ACT_SAT_HI32 : A_HIGH32;      // wave-uniform
    bool side_by_side = false;
    if (__ballot(!(a[0] >= t_hi || a[0] <= t_lo)) != 0ull) {
        side_by_side = true;
#pragma unroll
        for (int j = 1; j < 4; ++j) side_by_side = side_by_side && __ballot(!(a[j] >= t_hi || a[j] <= t_lo)) != 0ull;
    }
    if (side_by_side) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t.f[j] = rotor_force_from_action(a[j], norm_act, tq[j], nullptr, true);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) t.f[j] = rotor_force_sat(a[j], norm_act, tq[j]);
    }
#endif
    t.zt = z_torque32(tq);                             // BaseAviary.py:780
    return t;
}

// Body-frame resultant of the four rotor thrusts (A3), in the expressions physics_phase uses on a Thrust: the
// multi-wave kernels form it on the wave that computes the thrust and mail these four values instead of five floats.
template <typename R> struct Resultant {
    R fz, tx, ty, zt;
};
template <typename T> struct IsResultant { static constexpr bool value = false; };
template <typename R> struct IsResultant<Resultant<R>> { static constexpr bool value = true; };
template <typename R>
DN_DEV Resultant<R> rotor_resultant(const Thrust &th)
{
    const R F0 = th.f[0], F1 = th.f[1], F2 = th.f[2], F3 = th.f[3];
    Resultant<R> r;
    r.fz = (F0 + F1) + (F2 + F3);
    r.tx = K<R>::ARM * ((F2 + F3) - (F0 + F1));
    r.ty = K<R>::ARM * ((F1 + F2) - (F0 + F3));
    r.zt = (R)th.zt;
    return r;
}

// What a step reads of the drone's ENTRY state besides the rigid body itself: the stale distance pair (quirk Q1), the
// gate index / just_found flag, _computeTruncated on the un-incremented _steps (PBDroneEnv.py:444-454) and the entry
// velocities = current_vel / current_ang_v (quirk Q4).
template <typename R>
DN_DEV void flight_entry(Flight<R> &fl, const float4 G0, const float4 G2, const float4 G3, const int max_steps)
{
    const Meta m_e = unpack_meta(G3.w);
    fl.d_e = G0.w; fl.dprev_e = G2.w;
    fl.idx_e = m_e.idx; fl.just_found_e = m_e.just_found;
    fl.truncated = max_steps <= m_e.steps;             // PBDroneEnv.py:444-454, evaluated on the un-incremented _steps
    fl.vex = G2.x; fl.vey = G2.y; fl.vez = G2.z; fl.aex = G3.x; fl.aey = G3.y; fl.aez = G3.z;
}

// ---- A4: p.stepSimulation, one free rigid body [3P-recall of Bullet3 btMultiBody] -------------------------
// No `fp contract` licence in the recurrence: every fused multiply-add is spelled out (F), in one order.  Left to the
// compiler, WHICH product of a sum of products gets fused is picked per instantiation, and the picks differed between
// the kernel shapes (found by a soak run: one float32 ulp in a component of 1e-6 rad/s, which then diverges) -- the
// shapes, and with them every split of a fleet over ranks, are bit-identical only if the arithmetic is one sequence.
//
// The step is written as two halves that share nothing but their inputs, so that a kernel may run them on two waves:
//   physics_linear   v += a dt (a = R[:,2] fz/M - (0,0,G) - v (c + c|v|)), clamp, x += v dt       needs quat, vel, pos, fz
//   physics_angular  w += R I^-1 (tau - w_b x I w_b - I w_b (c + c|w_b|)) dt, clamp, q <- dq(w dt) q   needs quat, w, torques
// Both start from the same btMatrix3x3::setRotation terms (written once, below): the same expressions give the same bits
// on whichever wave evaluates them.  btMultiBody clamps every velocity coordinate at m_maxCoordinateVelocity; no reachable
// state gets there (thrust/weight = 5.5, damping), so each half tests its three once per wave and clamps only then -- the
// clamp is element-wise, so two tests over three values select the same results as one test over six.  NaN stays NaN.
#define F(a, b, c) FM<R>::fma((a), (b), (c))
template <typename R> struct QuatTerms {
    R xs, ys, zs, wxs, wys, wzs, yy, zz;
};
template <typename R> DN_DEV QuatTerms<R> quat_terms(const R qx, const R qy, const R qz, const R qw)
{   // btMatrix3x3::setRotation: s = 2 / |q|^2
    QuatTerms<R> t;
    // (one Newton step on the reciprocal: 2^-46 relative in a matrix whose products leave as float32 state words)
    const R s = R(2.0) * FM<R>::rcp_f32grade(F(qw, qw, F(qz, qz, F(qy, qy, qx * qx))));
    t.xs = qx * s; t.ys = qy * s; t.zs = qz * s;
    t.wxs = qw * t.xs; t.wys = qw * t.ys; t.wzs = qw * t.zs;
    t.yy = qy * t.ys; t.zz = qz * t.zs;
    return t;
}
template <typename R> struct Lin {
    R px, py, pz;          // new position
    R vx, vy, vz;          // new velocity
};
// The thrust direction = third column of the rotation matrix of the entry attitude: all physics_linear reads of the quaternion.  Its
// own function so that the wave that OWNS the attitude can form it (same expressions, same bits) and mail three values.
template <typename R> struct AttCol {
    R r02, r12, r22;
};
template <typename R>
DN_DEV AttCol<R> attitude_column_terms(const float4 G1, const QuatTerms<R> &t)
{
    const R qx = G1.x, qy = G1.y;
    AttCol<R> c;
    c.r02 = F(qx, t.zs, t.wys); c.r12 = F(qy, t.zs, -t.wxs); c.r22 = R(1.0) - F(qx, t.xs, t.yy);
    return c;
}
template <typename R>
DN_DEV AttCol<R> attitude_column(const float4 G1)
{
    const R qx = G1.x, qy = G1.y, qz = G1.z, qw = G1.w;
    return attitude_column_terms<R>(G1, quat_terms<R>(qx, qy, qz, qw));
}
// quat_terms of the identity attitude (a freshly reset body): s = 2, every product with x, y, z vanishes
template <typename R> DN_DEV QuatTerms<R> quat_terms_identity()
{
    QuatTerms<R> t;
    t.xs = t.ys = t.zs = t.wxs = t.wys = t.wzs = t.yy = t.zz = R(0.0);
    return t;
}
// physics_linear in two parts, like physics_angular: the damping products of the entry velocity (no thrust needed) | the rest.
template <typename R> struct LinPre {
    R vkx, vky, kl;        // vx kl, vy kl, kl = c + c |v|
};
template <typename R>
DN_DEV LinPre<R> physics_linear_pre(const float4 G2, const R damp = K<R>::LIN_DAMP)
{
    LinPre<R> l;
    const R vx = G2.x, vy = G2.y, vz = G2.z;
    // the damping norms enter the velocity update at dt * 0.04: a float32 root moves it by 1e-12 relative
    l.kl = F(damp, (R)__builtin_amdgcn_sqrtf((float)F(vz, vz, F(vy, vy, vx * vx))), damp);
    l.vkx = vx * l.kl; l.vky = vy * l.kl;
    return l;
}
template <typename R>
DN_DEV Lin<R> physics_linear_post(const float4 G0, const float4 G2, const AttCol<R> col, const LinPre<R> &l, const R fz, const R dax, const R day,
                                  const R daz, const bool extra)
{
    Lin<R> o;
    R px = G0.x, py = G0.y, pz = G0.z;
    R vx = G2.x, vy = G2.y, vz = G2.z;
    const R dt = K<R>::DT;
    const R r02 = col.r02, r12 = col.r12, r22 = col.r22;
    // linear: a = R[:,2] fz/M - (0,0,G) - v (c + c|v|)
    const R fm = fz * K<R>::INV_M;
    R awx = F(r02, fm, -l.vkx), awy = F(r12, fm, -l.vky), awz = F(-vz, l.kl, F(r22, fm, -K<R>::G));
    if (extra) { awx += dax; awy += day; awz += daz; }
    vx = F(awx, dt, vx); vy = F(awy, dt, vy); vz = F(awz, dt, vz);       // applyDeltaVeeMultiDof
    const R mv = K<R>::MAX_COORD_VEL;
    if (__builtin_expect(fmax(fmax(fabs(vx), fabs(vy)), fabs(vz)) > mv, 0)) {
        vx = clipv(vx, -mv, mv); vy = clipv(vy, -mv, mv); vz = clipv(vz, -mv, mv);
    }
    o.px = F(dt, vx, px); o.py = F(dt, vy, py); o.pz = F(dt, vz, pz);    // stepPositionsMultiDof
    o.vx = vx; o.vy = vy; o.vz = vz;
    return o;
}
template <typename R>
DN_DEV Lin<R> physics_linear_col(const float4 G0, const float4 G2, const AttCol<R> col, const R fz, const R dax, const R day, const R daz,
                                 const bool extra, const R damp = K<R>::LIN_DAMP)
{
    return physics_linear_post<R>(G0, G2, col, physics_linear_pre<R>(G2, damp), fz, dax, day, daz, extra);
}
template <typename R>
DN_DEV Lin<R> physics_linear(const float4 G0, const float4 G1, const float4 G2, const R fz, const R dax, const R day, const R daz,
                             const bool extra, const R damp = K<R>::LIN_DAMP)
{
    return physics_linear_col<R>(G0, G2, attitude_column<R>(G1), fz, dax, day, daz, extra, damp);
}
template <typename R> struct Ang {
    R qx, qy, qz, qw;      // new attitude (unit quaternion)
    R wx, wy, wz;          // new angular velocity
};
// physics_angular in two parts: everything that reads only the entry state (rotation matrix, body rates, gyroscopic and damping
// terms) and everything from the torques on.  A single-step kernel evaluates the first part while the thrust is still being computed
// on another wave; physics_angular = the two back to back, the same expressions in the same order.
template <typename R> struct AngPre {
    R qx, qy, qz, qw, wx, wy, wz;
    R r00, r01, r02, r10, r11, r12, r20, r21, r22;
    R ka, Iwx, Iwy, Iwz, gx, gy, gz;
};
// ... on quaternion terms the caller already holds: a wave that owns the attitude across the steps of a fused launch forms quat_terms of
// the NEW attitude once, at the end of a step (for the thrust direction it mails to the linear half), and starts the next step from them.
template <typename R>
DN_DEV AngPre<R> physics_angular_pre_terms(const float4 G1, const float4 G3, const QuatTerms<R> &t, const R damp = K<R>::ANG_DAMP);
template <typename R>
DN_DEV AngPre<R> physics_angular_pre(const float4 G1, const float4 G3, const R damp = K<R>::ANG_DAMP)
{
    const R qx = G1.x, qy = G1.y, qz = G1.z, qw = G1.w;
    return physics_angular_pre_terms<R>(G1, G3, quat_terms<R>(qx, qy, qz, qw), damp);
}
template <typename R>
DN_DEV AngPre<R> physics_angular_pre_terms(const float4 G1, const float4 G3, const QuatTerms<R> &t, const R damp)
{
    AngPre<R> a;
    const R qx = G1.x, qy = G1.y, qz = G1.z, qw = G1.w;
    const R wx = G3.x, wy = G3.y, wz = G3.z;
    a.r00 = R(1.0) - F(qy, t.ys, t.zz); a.r01 = F(qx, t.ys, -t.wzs); a.r02 = F(qx, t.zs, t.wys);
    a.r10 = F(qx, t.ys, t.wzs); a.r11 = R(1.0) - F(qx, t.xs, t.zz); a.r12 = F(qy, t.zs, -t.wxs);
    a.r20 = F(qx, t.zs, -t.wys); a.r21 = F(qy, t.zs, t.wxs); a.r22 = R(1.0) - F(qx, t.xs, t.yy);
    // angular, in the body frame: I dw = tau - w x (I w) - I w (c + c|w|)
    const R wbx = F(a.r20, wz, F(a.r10, wy, a.r00 * wx)), wby = F(a.r21, wz, F(a.r11, wy, a.r01 * wx)), wbz = F(a.r22, wz, F(a.r12, wy, a.r02 * wx));
    a.ka = F(damp, (R)__builtin_amdgcn_sqrtf((float)F(wz, wz, F(wy, wy, wx * wx))), damp);
    a.Iwx = K<R>::IXX * wbx; a.Iwy = K<R>::IYY * wby; a.Iwz = K<R>::IZZ * wbz;
    a.gx = F(wby, a.Iwz, -(wbz * a.Iwy)); a.gy = F(wbz, a.Iwx, -(wbx * a.Iwz)); a.gz = F(wbx, a.Iwy, -(wby * a.Iwx));
    a.qx = qx; a.qy = qy; a.qz = qz; a.qw = qw; a.wx = wx; a.wy = wy; a.wz = wz;
    return a;
}
template <typename R>
DN_DEV Ang<R> physics_angular_post(const AngPre<R> &a, const R tx, const R ty, const R ztq)
{
    Ang<R> o;
    const R qx = a.qx, qy = a.qy, qz = a.qz, qw = a.qw;
    R wx = a.wx, wy = a.wy, wz = a.wz;
    const R dt = K<R>::DT;
    const R dbx = F(-a.Iwx, a.ka, tx - a.gx) * K<R>::INV_IXX, dby = F(-a.Iwy, a.ka, ty - a.gy) * K<R>::INV_IYY,
            dbz = F(-a.Iwz, a.ka, ztq - a.gz) * K<R>::INV_IZZ;
    const R dwx = F(a.r02, dbz, F(a.r01, dby, a.r00 * dbx)), dwy = F(a.r12, dbz, F(a.r11, dby, a.r10 * dbx)), dwz = F(a.r22, dbz, F(a.r21, dby, a.r20 * dbx));
    wx = F(dwx, dt, wx); wy = F(dwy, dt, wy); wz = F(dwz, dt, wz);       // applyDeltaVeeMultiDof
    const R mv = K<R>::MAX_COORD_VEL;
    if (__builtin_expect(fmax(fmax(fabs(wx), fabs(wy)), fabs(wz)) > mv, 0)) {
        wx = clipv(wx, -mv, mv); wy = clipv(wy, -mv, mv); wz = clipv(wz, -mv, mv);
    }
    // exponential map: Bullet clamps the angle rate, fAngle = min(|w|, (pi/4)/dt), and takes
    // axis = w sin(h)/fAngle with h = fAngle dt/2, i.e. w (dt/2) sinc(h), and cos(h): both are even in h, so only
    // h^2 = min(|w|^2 dt^2/4, (pi/8)^2) is needed and the sqrt of |w|^2 never is.  (Bullet's |w| < 1e-3 Taylor
    // branch is the same function to 1e-24; and with |w_i| <= 100 the clamp itself cannot bind at dt = 1/240.)
    R h2 = (R(0.25) * dt * dt) * F(wz, wz, F(wy, wy, wx * wx));
    const R h2max = R(0.015625) * K<R>::PI * K<R>::PI;     // (pi/8)^2
    h2 = h2 > h2max ? h2max : h2;
    R sinc, aw;
    sinc_cos_small<R>(h2, sinc, aw);
    const R k = (R(0.5) * dt) * sinc;
    const R ax = wx * k, ay = wy * k, az = wz * k;
    const R nx = F(-az, qy, F(ay, qz, F(ax, qw, aw * qx)));
    const R ny = F(-ax, qz, F(az, qx, F(ay, qw, aw * qy)));
    const R nz = F(-ay, qx, F(ax, qy, F(az, qw, aw * qz)));
    const R nw_ = F(-az, qz, F(-ay, qy, F(-ax, qx, aw * qw)));
    const R inv = FM<R>::rsq_f32grade(F(nw_, nw_, F(nz, nz, F(ny, ny, nx * nx))));     // the unit quaternion leaves as four float32 words
    o.qx = nx * inv; o.qy = ny * inv; o.qz = nz * inv; o.qw = nw_ * inv;
    o.wx = wx; o.wy = wy; o.wz = wz;
    return o;
}
template <typename R>
DN_DEV Ang<R> physics_angular(const float4 G1, const float4 G3, const R tx, const R ty, const R ztq, const R damp = K<R>::ANG_DAMP)
{
    return physics_angular_post<R>(physics_angular_pre<R>(G1, G3, damp), tx, ty, ztq);
}

template <typename R, typename TH = Thrust, bool XOPT = false>
DN_DEV Flight<R> physics_phase(const TH &th, const float4 G0, const float4 G1, const float4 G2, const float4 G3,
                               const int max_steps, const Extras *x = nullptr)
{
    Flight<R> fl;
    flight_entry<R>(fl, G0, G2, G3, max_steps);
    // rotor thrusts along body z at the prop offsets (+,-) (-,-) (-,+) (+,+) * 0.028 (cf2x.urdf:42,54,66,78)
    R fz, tx, ty, ztq;
    R dax = R(0.0), day = R(0.0), daz = R(0.0);
    if constexpr (IsResultant<TH>::value) {               // the multi-wave kernels: summed one step ahead by another wave
        fz = th.fz; tx = th.tx; ty = th.ty; ztq = th.zt;
    } else {
    R F0 = th.f[0], F1 = th.f[1], F2 = th.f[2], F3 = th.f[3];
    if (XOPT) {
        const R px = G0.x, py = G0.y, pz = G0.z;
        const R qx = G1.x, qy = G1.y, qz = G1.z, qw = G1.w;
        const R vx = G2.x, vy = G2.y, vz = G2.z;
        (void)px; (void)py;
        const QuatTerms<R> t = quat_terms<R>(qx, qy, qz, qw);
        const R r00 = R(1.0) - F(qy, t.ys, t.zz), r01 = F(qx, t.ys, -t.wzs), r02 = F(qx, t.zs, t.wys);
        const R r10 = F(qx, t.ys, t.wzs), r11 = R(1.0) - F(qx, t.xs, t.zz), r12 = F(qy, t.zs, -t.wxs);
        const R r20 = F(qx, t.zs, -t.wys), r21 = F(qy, t.zs, t.wxs), r22 = R(1.0) - F(qx, t.xs, t.yy);
        if (x->gnd) {   // BaseAviary._groundEffect (BaseAviary.py:800-832): a second +z force on each prop link
            // |roll| < pi/2 and |pitch| < pi/2 on the cached rpy (getEulerFromQuaternion [3P-recall]): roll =
            // atan2(., w^2-x^2-y^2+z^2) is inside (-pi/2, pi/2) iff its second argument = r22 |q|^2 is positive;
            // pitch = asin(s), s = -r20, is +-pi/2 exactly in the gimbal branch |s| >= 0.99999
            const bool ok = fabs(r20) < R(0.99999) && r22 > R(0.0);
            const R X[4] = {K<R>::ARM, -K<R>::ARM, -K<R>::ARM, K<R>::ARM}, Y[4] = {-K<R>::ARM, -K<R>::ARM, K<R>::ARM, K<R>::ARM};
            R g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R h = pz + (r20 * X[j] + r21 * Y[j]);          // world z of prop link j (p.getLinkStates)
                h = h < R(GND_EFF_H_CLIP) ? R(GND_EFF_H_CLIP) : h;
                const R q = R(PROP_RADIUS) / (R(4.0) * h);
                g[j] = ok ? (R)gnd_effect_rotor(x->rpm[j], x->rpm_f32 != 0) * (q * q) : R(0.0);
            }
            F0 += g[0]; F1 += g[1]; F2 += g[2]; F3 += g[3];
        }
        if (x->drag) {  // BaseAviary._drag (:836-862): base_rot . (-DRAG_COEFF sum(omega) * vel) as a LINK_FRAME force on link 4
            const R sum = (R)drag_omega_sum(x->last, x->rpm_f32 != 0);
            const R kxy = R(-DRAG_XY) * sum, kz = R(-DRAG_Z) * sum;
            const R ux = kxy * vx, uy = kxy * vy, uz = kz * vz;
            const R bx = r00 * ux + r01 * uy + r02 * uz, by = r10 * ux + r11 * uy + r12 * uz, bz = r20 * ux + r21 * uy + r22 * uz;
            dax = (r00 * bx + r01 * by + r02 * bz) * K<R>::INV_M;
            day = (r10 * bx + r11 * by + r12 * bz) * K<R>::INV_M;
            daz = (r20 * bx + r21 * by + r22 * bz) * K<R>::INV_M;
        }
    }
    fz = (F0 + F1) + (F2 + F3);
    tx = K<R>::ARM * ((F2 + F3) - (F0 + F1));
    ty = K<R>::ARM * ((F1 + F2) - (F0 + F3));
    ztq = (R)th.zt;
    }
    const R damp = XOPT ? (R)x->damp : K<R>::LIN_DAMP;        // the option kernels take it at run time (dn_config.zero_damping)
    const Lin<R> lin = physics_linear<R>(G0, G1, G2, fz, dax, day, daz, XOPT, damp);
    const Ang<R> ang = physics_angular<R>(G1, G3, tx, ty, ztq, damp);
    fl.px = lin.px; fl.py = lin.py; fl.pz = lin.pz;
    fl.qx = ang.qx; fl.qy = ang.qy; fl.qz = ang.qz; fl.qw = ang.qw;
    fl.vx = (float)lin.vx; fl.vy = (float)lin.vy; fl.vz = (float)lin.vz;
    fl.wx = (float)ang.wx; fl.wy = (float)ang.wy; fl.wz = (float)ang.wz;
    return fl;
}
#undef F

// ---- A8 + A9 on the flight wave: _computeTerminated (PBDroneEnv.py:456-473) as evaluated inside _computeReward
// (:489) and again by BaseAviary.step (BaseAviary.py:448), the gate bookkeeping of _computeReward (:539-552),
// _update_state_post_step (:201-223, skipped on a terminated step: quirk Q5) and the body/bookkeeping half of the
// SubprocVecEnv auto-reset -> PBDroneEnv.reset (:609-665; _current_position is NOT reset: quirk Q3).
// Only the segment corridor depends on the waypoint index, so the common part of the collision test runs once
// and the segment test once per index that is actually needed.
// ---- N4: random spawn (PBDroneEnv(random_spawn=True), dormant in the reference: PBDroneEnv.py:622-627) --------------
// PositionGenerator.generate_random_point_around_line (position_generator.py:121-152, max_distance 0.1, bounds = aviary_dim)
// between two distinct target points drawn at random.  The reference draws from `random` / `np.random`; here Philox words
// keyed by (seed; global drone id, vector step, streams 11 / 12).  float64 whatever R is (a rare path).
DN_DEV void spawn_point(const DnParams &p, const unsigned long long gid, const unsigned long long step, double out[3])
{
    const int W = p.num_waypoints;
    if (W < 2) { out[0] = p.c64.spawn[0]; out[1] = p.c64.spawn[1]; out[2] = p.c64.spawn[2]; return; }
    unsigned r[4];
    philox4x32((unsigned)gid, (unsigned)(gid >> 32), (unsigned)step, 11u | ((unsigned)(step >> 32) << 8), (unsigned)p.seed,
               (unsigned)(p.seed >> 32), r);
    const int i = (int)(r[0] % (unsigned)W);
    int j = (int)(r[1] % (unsigned)(W - 1));                   // np.random.choice(W, size=2, replace=False)
    if (j >= i) j += 1;
    const double t = ((double)r[2] + 0.5) * (1.0 / 4294967296.0);
    const double u = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
    const double offset = -0.1 + (0.1 - -0.1) * u;             // random.uniform(-max_distance, max_distance)
    float z[4];
    noise4(p.seed, gid, step, 12u, z);                         // np.random.randn(3)
    const double *f = p.tab64 + i * DN_T_STRIDE + DN_T_WP, *g = p.tab64 + j * DN_T_STRIDE + DN_T_WP;     // global table: rare path
    const double dir[3] = {g[0] - f[0], g[1] - f[1], g[2] - f[2]};
    const double rv[3] = {(double)z[0], (double)z[1], (double)z[2]};
    const double perp[3] = {dir[1] * rv[2] - dir[2] * rv[1], dir[2] * rv[0] - dir[0] * rv[2], dir[0] * rv[1] - dir[1] * rv[0]};
    const double n = sqrt(perp[0] * perp[0] + perp[1] * perp[1] + perp[2] * perp[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = f[k] + t * (g[k] - f[k]);
        x += offset * (perp[k] / n);
        const double hi = x < p.c64.dim[3 + k] ? x : p.c64.dim[3 + k];      // max(lo, min(hi, x)), :113-118
        out[k] = p.c64.dim[k] > hi ? p.c64.dim[k] : hi;
    }
}

// rules_verdict: everything the OTHER phases of the step wait for (collision / gate / termination flags and the distance
// the reset observation shows); rules_commit: the state words that go back to HBM, with the auto-reset of the body.
// rules_phase = the two back to back.
template <typename R> struct RulesMid {
    int idx, just_found, steps;
    R d, d_prev;
    bool terminated, done;
};
template <typename R>
DN_DEV Verdict<R> rules_verdict(const DnParams &p, const DnConsts<R> &c, const R *s_tab, const GateRow<R> &row_e,
                                const Flight<R> &fl, const float4 G3e, RulesMid<R> &m)
{
// (explicit fused multiply-adds, no contraction licence: see physics_linear)
    const Meta m_e = unpack_meta(G3e.w);
    const R px = fl.px, py = fl.py, pz = fl.pz;
    // rotation entry R[2][2] of the new (unit) attitude, for the ground-contact approximation only
    const R r22n = p.ground_contact ? FM<R>::fma(R(-2.0), FM<R>::fma(fl.qy, fl.qy, fl.qx * fl.qx), R(1.0)) : R(1.0);
    int idx = m_e.idx, just_found = m_e.just_found;
    const bool seg_track = p.cylinder && !p.circle;
    const bool coll1 = collision_common<R>(p, c, px, py, pz, r22n) ||
                       (seg_track && outside_segment_corridor_row<R>(c, row_e, px, py, pz));
    const bool found_now = (R)fl.d_e <= c.threshold;   // :539
    const bool last_gate = idx + 1 == p.num_waypoints;
    R d_prev = (R)fl.dprev_e;
    bool terminated;
    if (coll1) terminated = true;                      // :489-490 (entry _is_done is always False here)
    else if (found_now) {
        idx += 1;
        if (last_gate) terminated = true;              // :542-546, _is_done
        else {
            just_found = 1;                            // :548-552
            // second _computeTerminated: the common part is already known to be false
            terminated = seg_track && outside_segment_corridor<R>(c, s_tab, px, py, pz, idx);
        }
        d_prev = (R)fl.d_e;
    } else { just_found = 0; d_prev = (R)fl.d_e; terminated = false; }
    int steps = m_e.steps;
    R d = (R)fl.d_e;
    if (!terminated) {                                 // _update_state_post_step
        steps += 1;
        R wx = row_e.wp[0], wy = row_e.wp[1], wz = row_e.wp[2];
        if (idx != m_e.idx) {                          // a gate was passed this step: the next waypoint (rare)
            const R *wp = s_tab + idx * DN_T_STRIDE;
            wx = wp[0]; wy = wp[1]; wz = wp[2];
        }
        const R ex = wx - px, ey = wy - py, ez = wz - pz;
        d = FM<R>::sqrt0(FM<R>::fma(ez, ez, FM<R>::fma(ey, ey, ex * ex)));
    }
    Verdict<R> v;
    v.d_obs = d; v.coll1 = coll1; v.terminated = terminated;
    m.idx = idx; m.just_found = just_found; m.steps = steps; m.d = d; m.d_prev = d_prev;
    m.terminated = terminated; m.done = terminated || fl.truncated != 0;
    return v;
}
template <typename R, bool SPAWN = false>
DN_DEV void rules_commit(const DnConsts<R> &c, const R (&wp0)[3], const Flight<R> &fl, const RulesMid<R> &m, const float4 G0e,
                         const float4 G3e, float4 *g6_blk, const unsigned li, const bool active,
                         float4 &G0, float4 &G1, float4 &G2, float4 &G3, const DnParams *sp = nullptr,
                         const unsigned long long gid = 0ull, const unsigned long long step = 0ull)
{
    const Meta m_e = unpack_meta(G3e.w);
    const R px = fl.px, py = fl.py, pz = fl.pz;
    const bool terminated = m.terminated, done = m.done;
    int idx = m.idx, just_found = m.just_found, steps = m.steps;
    R d = m.d, d_prev = m.d_prev;
    // the advanced body state as the float32 words that go back to HBM
    float4 S0 = make_float4((float)px, (float)py, (float)pz, 0.0f);
    float4 S1 = make_float4((float)fl.qx, (float)fl.qy, (float)fl.qz, (float)fl.qw);
    float4 S2 = make_float4(fl.vx, fl.vy, fl.vz, 0.0f);
    float4 S3 = make_float4(fl.wx, fl.wy, fl.wz, 0.0f);
    if (__ballot(done) != 0ull) {                      // wave-uniform: most wave-steps of a long flight skip this
        if (done) {
            R cpx, cpy, cpz;
            if (!terminated) { cpx = px; cpy = py; cpz = pz; }            // post-step ran: it is the new position
            else if (__builtin_expect(m_e.steps > 0, 1)) { cpx = G0e.x; cpy = G0e.y; cpz = G0e.z; }
            else { const float4 G6 = g6_blk[li]; cpx = G6.x; cpy = G6.y; cpz = G6.z; }
            if (active && !(terminated && m_e.steps == 0)) {       // .w is unused
                float *g6f = reinterpret_cast<float *>(g6_blk + li);
                g6f[0] = (float)cpx; g6f[1] = (float)cpy; g6f[2] = (float)cpz;
            }
            // freshly loaded body at the spawn pose, at rest
            S0 = make_float4((float)c.spawn[0], (float)c.spawn[1], (float)c.spawn[2], 0.0f);
            if (SPAWN && sp->random_spawn) {           // this episode's INIT_XYZS[0]; _current_position follows it (PBDroneEnv.py:624-626)
                double q[3];
                spawn_point(*sp, gid, step, q);
                S0 = make_float4((float)q[0], (float)q[1], (float)q[2], 0.0f);
                cpx = (R)S0.x; cpy = (R)S0.y; cpz = (R)S0.z;
                if (active) { float *g6f = reinterpret_cast<float *>(g6_blk + li); g6f[0] = S0.x; g6f[1] = S0.y; g6f[2] = S0.z; }
            }
            S1 = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            S2 = S3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const R ex = cpx - wp0[0], ey = cpy - wp0[1], ez = cpz - wp0[2];
            d = FM<R>::sqrt0(FM<R>::fma(ez, ez, FM<R>::fma(ey, ey, ex * ex)));    // :651
            d_prev = d;                                                   // :652
            idx = 0; steps = 0; just_found = 0;
        }
    }
    S0.w = (float)d; S2.w = (float)d_prev; S3.w = pack_meta(steps, idx, just_found);
    G0 = S0; G1 = S1; G2 = S2; G3 = S3;
}
template <typename R, bool SPAWN = false>
DN_DEV Verdict<R> rules_phase(const DnParams &p, const DnConsts<R> &c, const R *s_tab, const GateRow<R> &row_e, const R (&wp0)[3],
                              const Flight<R> &fl, const float4 G0e, const float4 G3e, float4 *g6_blk, const unsigned li,
                              const bool active, float4 &G0, float4 &G1, float4 &G2, float4 &G3,
                              const unsigned long long gid = 0ull, const unsigned long long step = 0ull)
{
    RulesMid<R> m;
    const Verdict<R> v = rules_verdict<R>(p, c, s_tab, row_e, fl, G3e, m);
    rules_commit<R, SPAWN>(c, wp0, fl, m, G0e, G3e, g6_blk, li, active, G0, G1, G2, G3, &p, gid, step);
    return v;
}

// ---- A5 on the flight wave: the attitude of the post-physics pose as the observation and the reward read it -----
template <typename R, bool EULER = true>      // EULER = false: the forward vector only (a wave that does not pack the observation)
DN_DEV void attitude_phase(Flight<R> &fl)
{
    const R qx = fl.qx, qy = fl.qy, qz = fl.qz, qw = fl.qw;
    // p.getEulerFromQuaternion [3P-recall of pybullet.c].  The three angles only feed observation columns 3..5
    // (float32, bar 1e-5): the quaternion products are formed in R, the inverse trigonometry runs in float32
    // (atan2f ~1e-7 rad).  The forward vector feeds a compare and stays in R.
    float roll_num32, roll_den32, pitch32, yaw32;
    R fwx, fwy, fwz;
    {
        // Sums of several products are written as explicit fused multiply-adds in ONE order: left to `fp contract`, the
        // compiler picks which product of a sum to fuse per instantiation, and the pick can differ between the kernel
        // shapes (a value that arrives through LDS ranks differently from one computed in place) -- one float32 ulp in
        // one observation value per ~1e9, found by a soak run.
        const R sarg = R(-2.0) * FM<R>::fma(qx, qz, -(qw * qy));
        const R ys = R(2.0) * FM<R>::fma(qx, qy, qw * qz);
        const R yc = FM<R>::fma(qw, qw, FM<R>::fma(qx, qx, -FM<R>::fma(qy, qy, qz * qz)));     // w^2 + x^2 - y^2 - z^2
        if (__builtin_expect(sarg <= R(-0.99999) || sarg >= R(0.99999), 0)) {   // gimbal-lock branches: rare, keep them literal (and out of line)
            R pitch, yaw;
            roll_num32 = 0.0f; roll_den32 = 1.0f;             // roll = 0 = atan2(0, 1)
            if (sarg < R(0.0)) { pitch = R(-0.5) * K<R>::PI; yaw = R(2.0) * atan2(qx, -qy); }
            else { pitch = R(0.5) * K<R>::PI; yaw = R(2.0) * atan2(-qx, qy); }
            const R cpit = cos(pitch);
            fwx = cos(yaw) * cpit; fwy = sin(yaw) * cpit; fwz = sin(pitch);
            pitch32 = (float)pitch; yaw32 = (float)yaw;
        } else {
            if (EULER) {
            roll_num32 = (float)(R(2.0) * FM<R>::fma(qy, qz, qw * qx));
            roll_den32 = (float)FM<R>::fma(qw, qw, FM<R>::fma(qz, qz, -FM<R>::fma(qx, qx, qy * qy)));   // w^2 - x^2 - y^2 + z^2
            yaw32 = atan2_fast32((float)ys, (float)yc);
            // asin(s) is ill-conditioned towards +-1; cos(pitch) = |(r00, r10)| = sqrt(yc^2 + ys^2) for the unit quaternion
            // of Flight, so pitch = atan2(s, cos pitch) is well conditioned everywhere (tumbling drones sit beyond
            // 72 deg often enough that a float64 asin tail would run on most wave-steps of a bang-bang workload)
            pitch32 = atan2_fast32((float)sarg, __builtin_amdgcn_sqrtf((float)FM<R>::fma(yc, yc, ys * ys)));
            } else { roll_num32 = 0.0f; roll_den32 = 1.0f; yaw32 = pitch32 = 0.0f; }
            fwx = yc; fwy = ys; fwz = sarg;
        }
    }
    fl.roll_num32 = roll_num32; fl.roll_den32 = roll_den32; fl.pitch32 = pitch32; fl.yaw32 = yaw32;
    fl.fwx = fwx; fl.fwy = fwy; fl.fwz = fwz;
}

// What the report wave carries from observe_phase to report_phase.
template <typename R> struct Observed {
    float o[DN_OBS_DIM];   // step observation, after sensor noise and the normaliser (also terminal_observation)
    R r_normal;            // _computeReward's ordinary branch, before /25
    float r_found32;       // gate-pass branch, float32 as the reference accumulates it
};

// ---- A5 + A6 + the value side of A7 on the report wave ---------------------------------------------------
// observe_columns: the observation row; reward_candidates: both value branches of _computeReward.  The two read the
// same Flight and share no intermediate, so a kernel may run them on two waves; observe_phase = both on one.
// The row in two halves that share nothing: the columns that read the linear state (position, velocity, distance: 0 1 2 6 7 8 12)
// and those that read the attitude and the angular velocity (3 4 5 9 10 11) -- a kernel may fill them on two waves.
template <typename R>
DN_DEV void observe_columns_lin(const DnParams &p, const DnConsts<R> &c, const Flight<R> &fl, float o[DN_OBS_DIM])
{
    // _computeObs (PBDroneEnv.py:296-336, :338-398), stale distance d_e (quirk Q1).  The reference clips position /
    // yaw / distance columns to the float32 range before the cast (:326); positions are bounded by the aviary box
    // plus one step at the velocity cap, so those clips can never bind and are not evaluated.  clip(v, -3, 3)/3 is
    // monotone, so it equals clip(float32(v/3), -1, 1) exactly.
    o[0] = (float)(fl.px * c.inv_dim[0]);
    o[1] = (float)(fl.py * c.inv_dim[1]);
    o[2] = (float)(fl.pz * c.inv_dim[2]);
    const float third32 = (float)K<R>::THIRD;
    const float v6 = (float)((R)fl.vx * K<R>::THIRD), v7 = (float)((R)fl.vy * K<R>::THIRD), v8 = (float)((R)fl.vz * K<R>::THIRD);
    o[6] = __builtin_amdgcn_fmed3f(v6, -1.0f, 1.0f);
    o[7] = __builtin_amdgcn_fmed3f(v7, -1.0f, 1.0f);
    o[8] = __builtin_amdgcn_fmed3f(v8, -third32, third32);
    // np.clip propagates NaN, v_med3 does not: one test per wave, the selects only where it fires (a NaN action poisons the state)
    if (__builtin_expect(__ballot(__builtin_isunordered(v6, v7) || v8 != v8) != 0ull, 0)) {
        o[6] = v6 != v6 ? v6 : o[6]; o[7] = v7 != v7 ? v7 : o[7]; o[8] = v8 != v8 ? v8 : o[8];
    }
    o[12] = p.include_distance ? (float)((R)fl.d_e * c.inv_max_target_dist) : 0.0f;
}
template <typename R>
DN_DEV void observe_columns_att(const Flight<R> &fl, float o[DN_OBS_DIM])
{
    const float roll32 = atan2_fast32(fl.roll_num32, fl.roll_den32), pitch32 = fl.pitch32, yaw32 = fl.yaw32;    // attitude_phase
    const float inv_pi32 = (float)K<R>::INV_PI;
    o[3] = roll32 * inv_pi32;
    o[4] = pitch32 * inv_pi32;
    o[5] = yaw32 * inv_pi32;
    // ang_v / |ang_v|, zero stays zero (:383-384).  The three words are float32 state and the columns leave as float32: the norm is taken
    // there (v_rsq_f32, 1 ulp: the columns within 2 float32 ulp of the float64 quotient, 2.4e-7 on a unit vector) wherever w^2 is a normal
    // float32 with room to spare; a drone turning slower than 1e-15 rad/s (w^2 would underflow) takes the float64 form -- a per-lane
    // choice (the block is skipped when no lane of the wave needs it), so a drone's columns do not depend on its neighbours.
    const float w2f = __builtin_fmaf(fl.wz, fl.wz, __builtin_fmaf(fl.wy, fl.wy, fl.wx * fl.wx));
    const float rw32 = w2f != 0.0f ? __builtin_amdgcn_rsqf(w2f) : 0.0f;
    o[9] = fl.wx * rw32; o[10] = fl.wy * rw32; o[11] = fl.wz * rw32;
    if (__builtin_expect(w2f < 1e-30f && (fl.wx != 0.0f || fl.wy != 0.0f || fl.wz != 0.0f), 0)) {
        const R w2 = FM<R>::fma((R)fl.wz, (R)fl.wz, FM<R>::fma((R)fl.wy, (R)fl.wy, (R)fl.wx * (R)fl.wx));   // explicit order, see attitude_phase
        const R rw = FM<R>::rsq_f32grade(w2);
        o[9] = (float)((R)fl.wx * rw); o[10] = (float)((R)fl.wy * rw); o[11] = (float)((R)fl.wz * rw);
    }
}
template <typename R>
DN_DEV void observe_columns(const DnParams &p, const DnConsts<R> &c, const Flight<R> &fl, float o[DN_OBS_DIM])
{
    observe_columns_lin<R>(p, c, fl, o);
    observe_columns_att<R>(fl, o);
}
// _computeReward (PBDroneEnv.py:475-571), both value branches; report_scalars selects once the verdict is in.
// reward_entry: the terms that read the ENTRY state only (distance gain, 3 e^{-2d}, smoothness penalties on the stale
// velocity copies); reward_pose: the orientation term on the new pose and the assembly, in the reference's order of
// additions.  reward_candidates = the two back to back (a kernel may evaluate the first while the pose is being computed).
template <typename R> struct RewardPre {
    R r0;                  // 3 e^{-2d} + 3000 (d_prev - d)                                   :555-556
    R s_lin, s_ang;        // |dv|, |dw| where the smoothness penalty applies                  :599-607
    bool pen_lin, pen_ang, found_now, last_gate;
};
// smoothness_reward (PBDroneEnv.py:599-607), one of its two terms: -|d| if |d| > limit, d = the entry rate minus its stale post-step
// copy (quirk Q4).  Its own function so that the wave that OWNS the rate (the balanced four-wave kernel: L the linear one, A the angular
// one) can form it -- the same expressions, the same bits -- and mail a float and a flag.  The penalty enters the reward at 1/25: a
// float32 root (1e-7 relative) is far inside the reward's 1e-5 bar.
template <typename R> struct Smooth {
    R s;                   // |d| where the penalty applies, else 0 (a float32 value)
    bool pen;
};
template <typename R>
DN_DEV Smooth<R> smooth_term(const float ex, const float ey, const float ez, const float4 P, const R lim2)
{
    const R lx = (R)ex - (R)P.x, ly = (R)ey - (R)P.y, lz = (R)ez - (R)P.z;
    const R a2 = FM<R>::fma(lz, lz, FM<R>::fma(ly, ly, lx * lx));
    Smooth<R> o;
    o.pen = a2 > lim2;
    o.s = R(0.0);
    if (o.pen) o.s = (R)__builtin_amdgcn_sqrtf((float)a2);
    return o;
}
// the terms of reward_entry that read the distance pair and the gate index only
template <typename R>
DN_DEV void reward_entry_core(const DnParams &p, const DnConsts<R> &c, const Flight<R> &fl, RewardPre<R> &q)
{
    q.found_now = (R)fl.d_e <= c.threshold;
    q.last_gate = fl.idx_e + 1 == p.num_waypoints;
    // :555 3 e^{-2d} (v_exp_f32: 1e-7 rel, 1e-8 in the reward) + :556
    const R gain = fl.just_found_e ? R(0.0) : ((R)fl.dprev_e - (R)fl.d_e) * R(3000.0);
    q.r0 = FM<R>::fma(R(3.0), (R)__builtin_amdgcn_exp2f((float)(R(-2.0 * 1.4426950408889634) * (R)fl.d_e)), gain);
}
template <typename R>
DN_DEV RewardPre<R> reward_entry(const DnParams &p, const DnConsts<R> &c, const Flight<R> &fl, const float4 G4, const float4 G5)
{
    RewardPre<R> q;
    reward_entry_core<R>(p, c, fl, q);
    // smoothness_reward (:599-607) on the stale post-step copies (quirk Q4): -|dv| if |dv| > 0.7 (needs > 160 m/s^2: rare), -|dw| if > 0.3
    const Smooth<R> sl = smooth_term<R>(fl.vex, fl.vey, fl.vez, G4, R(0.7) * R(0.7));
    const Smooth<R> sa = smooth_term<R>(fl.aex, fl.aey, fl.aez, G5, R(0.3) * R(0.3));
    q.pen_lin = sl.pen; q.pen_ang = sa.pen;
    q.s_lin = sl.s; q.s_ang = sa.s;
    return q;
}
// reward_pose in two halves: the orientation term against the waypoint the taken branch refers to (reads the pose), and the assembly
// of both value branches in the reference's order of additions (reads the entry-state terms) -- a kernel may run them on two waves.
template <typename R>
DN_DEV int reward_orientation(const DnParams &p, const R *s_tab, const Flight<R> &fl, const bool found_now, const bool last_gate)
{
    const int idx_ori = (found_now && !last_gate) ? fl.idx_e + 1 : fl.idx_e;
    return orientation_reward<R>(fl.fwx, fl.fwy, fl.fwz, fl.px, fl.py, fl.pz, s_tab + idx_ori * DN_T_STRIDE);
}
template <typename R>
DN_DEV void reward_assemble(const RewardPre<R> &q, const int ori, R &r_normal, float &r_found32)
{
    float r32 = 0.0f;
    if (q.last_gate) r32 = r32 + 200.0f;                                      // :542-546
    else { r32 = r32 + 75.0f; r32 = r32 + (float)(ori * 5); }                 // :548-552
    r_found32 = r32;
    R r = q.r0 + (R)(ori * 3);                                                // :557
    if (q.pen_lin) r = r - q.s_lin;
    if (q.pen_ang) r = r - q.s_ang;
    r_normal = r;
}
template <typename R>
DN_DEV void reward_pose(const DnParams &p, const R *s_tab, const Flight<R> &fl, const RewardPre<R> &q, R &r_normal, float &r_found32)
{
    // The orientation term is evaluated once, against the waypoint the taken branch refers to.
    reward_assemble<R>(q, reward_orientation<R>(p, s_tab, fl, q.found_now, q.last_gate), r_normal, r_found32);
}
template <typename R>
DN_DEV void reward_candidates(const DnParams &p, const DnConsts<R> &c, const R *s_tab, const Flight<R> &fl, const float4 G4,
                              const float4 G5, R &r_normal, float &r_found32)
{
    const RewardPre<R> q = reward_entry<R>(p, c, fl, G4, G5);
    reward_pose<R>(p, s_tab, fl, q, r_normal, r_found32);
}
template <typename R, bool NORM, bool NOISE>
DN_DEV Observed<R> observe_phase(const DnParams &p, const DnConsts<R> &c, const R *s_tab, const Flight<R> &fl,
                                 const float4 G4, const float4 G5, const unsigned long long gid, const unsigned long long step_count,
                                 Rms &rms)
{
    Observed<R> ob;
    observe_columns<R>(p, c, fl, ob.o);
    reward_candidates<R>(p, c, s_tab, fl, G4, G5, ob.r_normal, ob.r_found32);
    // sensor noise / per-drone normaliser act on the step observation (which is also terminal_observation)
    if (NOISE && p.obs_noise_sigma > 0.0f) add_obs_noise(p, gid, step_count, 1u, ob.o);
    if (NORM) normalize_obs(rms, ob.o);
    return ob;
}

// make_env's optional reward wrappers, inside Monitor (PBDroneSimulator.py:191-194): TransformReward(clip(-10, 10))
// then NormalizeReward (normalize.py:100-147; batch of one, so batch_var = 0): the discounted return and its running
// mean / variance / count are four float64 per drone, resident in registers for the launch like the observation
// statistics.  Same reciprocal / rsq treatment as normalize_obs (equal to the literal form to 1e-16).
struct RewNorm {
    double returns, mean, var, count;
};
DN_DEV void load_rewnorm(const DnParams &p, long long i, RewNorm &r)
{
    r.returns = p.st.rr[i]; r.mean = p.st.rr[p.n + i]; r.var = p.st.rr[2 * p.n + i]; r.count = p.st.rr[3 * p.n + i];
}
DN_DEV void store_rewnorm(const DnParams &p, long long i, const RewNorm &r)
{
    p.st.rr[i] = r.returns; p.st.rr[p.n + i] = r.mean; p.st.rr[2 * p.n + i] = r.var; p.st.rr[3 * p.n + i] = r.count;
}
DN_DEV double reward_wrappers(const DnParams &p, RewNorm &rn, double r, bool done)
{
    if (p.clip_rew) r = clipv(r, -10.0, 10.0);
    if (p.norm_rew) {
        rn.returns = __builtin_fma(rn.returns, 0.99, r);
        const double tot = rn.count + 1.0;
        const double inv = rcp_f64(tot);
        const double delta = rn.returns - rn.mean;
        rn.mean = __builtin_fma(delta, inv, rn.mean);
        rn.var = __builtin_fma(delta * delta, inv, rn.var) * (rn.count * inv);
        rn.count = tot;
        const double s = rn.var + 1e-8;
        double y = __builtin_amdgcn_rsq(s);
        y = __builtin_fma(y, __builtin_fma(-(0.5 * s * y), y, 0.5), y);
        y = __builtin_fma(y, __builtin_fma(-(0.5 * s * y), y, 0.5), y);
        r = r * y;
        if (done) rn.returns = 0.0;
    }
    return r;
}

// llrint(x * 1e6) for |x| < 2^51 * 1e-6 (any episode return), round to nearest even: adding 1.5 * 2^52 leaves the integer
// in the low mantissa bits -- four instructions instead of the fifteen of a float64 -> int64 conversion.
DN_DEV long long fixed6(double x)
{
    const double m = x * 1e6 + 6755399441055744.0;
    return __double_as_longlong(m) - 0x4338000000000000ll;
}

// Episode statistics of one tile, accumulated in (wave-uniform) registers over all the steps of a launch and added
// to the tile's slot in HBM once, at the end: the slot read-modify-write would otherwise put an HBM/L2 round trip
// on the report wave's critical path in every step that finishes an episode.
struct StatAcc {
    long long episodes = 0, truncated = 0, completed = 0, sum_len = 0, sum_found = 0, sum_ret_fix = 0;
};
DN_DEV void flush_stats(const DnParams &p, const StatAcc &a, unsigned long long steps_after, unsigned lane)
{
    if (lane == 0) {
        DnStatSlot sl = p.st.stats[blockIdx.x];
        sl.episodes += a.episodes; sl.truncated += a.truncated; sl.completed += a.completed;
        sl.sum_len += a.sum_len; sl.sum_found += a.sum_found; sl.sum_ret_fix += a.sum_ret_fix;
        sl.step_count = steps_after;
        p.st.stats[blockIdx.x] = sl;
    }
}
// The same with the slot read at the TOP of the kernel (the single-step launches: a read-modify-write at the end would put
// a whole memory round trip, ~0.5 us, on the tail of a 5 us kernel; the slot belongs to this workgroup alone).
DN_DEV void flush_stats_preloaded(const DnParams &p, DnStatSlot sl, const StatAcc &a, unsigned long long steps_after, unsigned lane,
                                  const long long tile)
{
    if (lane == 0) {
        sl.episodes += a.episodes; sl.truncated += a.truncated; sl.completed += a.completed;
        sl.sum_len += a.sum_len; sl.sum_found += a.sum_found; sl.sum_ret_fix += a.sum_ret_fix;
        sl.step_count = steps_after;
        p.st.stats[tile] = sl;
    }
}

// ---- A7 select + A10/A11 on the report wave: Monitor, SubprocVecEnv worker, outputs ---------------------------
// TILE: how the observation rows leave -- 0 through the LDS tile, written at once (one-wave kernels); 1 parked in the LDS tile
// and streamed out by the caller one step later (two-wave kernels); 2 straight from the lane's registers, 52 contiguous
// bytes per lane as three 16-byte stores and one 4-byte store at 4-byte alignment (three-wave kernel: the report wave
// has no phase to hide the tile's LDS round trip behind, and 17 fewer instructions per step is what counts there; the
// one-wave kernels keep the tile: at 2 M drones, where they are bandwidth bound, the scattered 16-byte stores cost 13 %).
struct __attribute__((packed, aligned(4))) ObsQuad { float x, y, z, w; };
DN_DEV void store_obs_direct(float *row, const float o[DN_OBS_DIM])
{
    *reinterpret_cast<ObsQuad *>(row) = ObsQuad{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<ObsQuad *>(row + 4) = ObsQuad{o[4], o[5], o[6], o[7]};
    *reinterpret_cast<ObsQuad *>(row + 8) = ObsQuad{o[8], o[9], o[10], o[11]};
    row[12] = o[12];
}
// The low byte of Monitor's running return (see report_scalars): k * 2^(e - 31) for hi = 1.m * 2^e, zero below 2^-95.
DN_DEV float ret_lo(const float hi, const int lenword)
{
    const int eb = (__float_as_int(hi) >> 23) & 0xFF;
    return (float)(lenword >> 24) * __int_as_float((eb > 31 ? eb - 31 : 0) << 23);
}
DN_DEV int ret_lo_byte(const double ep_ret, const float hi)
{
    const int eb = (__float_as_int(hi) >> 23) & 0xFF;
    const bool ok = eb > 31 && eb < 255;
    const float inv = __int_as_float((ok ? 285 - eb : 127) << 23);
    const double q = __builtin_rint((ep_ret - (double)hi) * (double)inv);
    const int k = (int)fmin(fmax(q, -128.0), 127.0);
    return ok ? (k & 0xFF) : 0;
}
// report_scalars: A7 select + Monitor + episode statistics + the scalar outputs; report_obs: the observation row(s) of
// the step -- terminal_observation and the reset observation of a finished drone (quirk Q2), sensor noise, normaliser.
// The two share only the verdict, so a kernel may run them on two waves; report_phase = both on one.
template <typename R, bool REW>
DN_DEV void report_scalars(const DnParams &p, const DnConsts<R> &c, const StepOut &out, const Flight<R> &fl, const Verdict<R> &v,
                           const R r_normal, const float r_found32, const unsigned li, const unsigned lane, const bool active,
                           float4 &G4, float4 &G5, StatAcc &acc, RewNorm &rn)
{
    const bool coll1 = v.coll1 != 0, terminated = v.terminated != 0, truncated = fl.truncated != 0;
    const bool found_now = !coll1 && (R)fl.d_e <= c.threshold;
    const bool is_done = found_now && fl.idx_e + 1 == p.num_waypoints;
    const int found = fl.idx_e + (found_now ? 1 : 0);
    const bool done = terminated || truncated;
    R reward;
    if (coll1) reward = R(-10.0);                                                 // :489-490
    else if (found_now) reward = (R)(r_found32 / 25.0f);                          // :568-571
    else reward = r_normal * K<R>::INV_25;
    if (REW) reward = (R)reward_wrappers(p, rn, (double)reward, done);       // --clip_rew / --norm_rew (compiled out otherwise)
    // Monitor sums the episode's rewards in float64 (a Python float); the state keeps the running return as the float32
    // hi (g4.w) plus a signed byte k in the top of the length word (g5.w), return = hi + k * ulp(hi) / 256: the part of
    // the sum that hi's rounding dropped, so that re-rounding it every step does not drift over a 4096-step episode
    // (32 mantissa bits; a fifth 16-byte group for a full low word cost 11 % of the bandwidth-bound 2 M-drone step)
    const int lenword = __float_as_int(G5.w);
    const int eplen_e = lenword & 0xFFFFFF;
    R ep_ret = ((R)G4.w + (R)ret_lo(G4.w, lenword)) + reward;
    int ep_len = eplen_e + 1;
    // prev_vel / prev_ang_v shift in _update_state_post_step (skipped on a terminated step, quirk Q5)
    float4 S4 = G4, S5 = G5;
    if (!terminated) {
        S4 = make_float4(fl.vex, fl.vey, fl.vez, 0.0f);
        S5 = make_float4(fl.aex, fl.aey, fl.aez, 0.0f);
    }
    const unsigned long long done_ballot = __ballot(done && active);
    if (done_ballot != 0ull) {                         // wave-uniform: waves without a finished drone skip all of this
        const long long fix = fixed6((double)ep_ret);                             // Monitor 'r' in 1e-6 fixed point
        if (done) {
            if (active) {
                if (out.ep_return) out.ep_return[li] = (float)ep_ret;
                if (out.ep_length) out.ep_length[li] = ep_len;
            }
            S4 = S5 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            ep_ret = R(0.0); ep_len = 0;
        }
        // Episode statistics of this wave -> the launch's accumulator (no atomics, deterministic).  Counts are ballots
        // + popcount on the scalar unit; the sums walk the set bits of the done ballot (typically one to three
        // finished drones per wave-step) with v_readlane instead of a 6-stage cross-lane reduction of 64-bit values.
        const long long n_trunc = __popcll(__ballot(done && active && truncated && !terminated));
        const long long n_compl = __popcll(__ballot(done && active && is_done));
        long long s_len = 0, s_fd = 0, s_ret = 0;
        const int fix_lo = (int)(unsigned)(fix & 0xFFFFFFFFll), fix_hi = (int)(fix >> 32);
        for (unsigned long long m = done_ballot; m != 0ull; m &= m - 1ull) {
            const int l = __builtin_ctzll(m);
            s_len += (long long)__builtin_amdgcn_readlane(eplen_e, l) + 1;
            s_fd += (long long)__builtin_amdgcn_readlane(found, l);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(fix_lo, l);
            const int hi = __builtin_amdgcn_readlane(fix_hi, l);
            s_ret += (long long)(((unsigned long long)(unsigned)hi << 32) | lo);
        }
        acc.episodes += __popcll(done_ballot); acc.truncated += n_trunc; acc.completed += n_compl;
        acc.sum_len += s_len; acc.sum_found += s_fd; acc.sum_ret_fix += s_ret;
    }
    S4.w = (float)ep_ret; S5.w = __int_as_float(ep_len | (ret_lo_byte((double)ep_ret, S4.w) << 24));
    G4 = S4; G5 = S5;
    if (active) {
        out.reward[li] = (float)reward;
        out.done[li] = (uint8_t)done;
        out.truncated[li] = (uint8_t)(truncated && !terminated);
        out.found[li] = found;
    }
    if (out.done_word && lane == 0) *out.done_word = done_ballot;
}
template <typename R, bool NORM, bool NOISE, int TILE, bool SPAWN = false>
DN_DEV void report_obs(const DnParams &p, const DnConsts<R> &c, float *s_tile, const StepOut &out, const bool truncated,
                       const Verdict<R> &v, float *o, const unsigned long long gid, const unsigned long long step_count,
                       const unsigned li, const unsigned lane, const unsigned rows, const bool active, Rms &rms)
{
    const bool done = v.terminated != 0 || truncated;
    const unsigned long long done_mask = __ballot(done && active);
    if (done_mask != 0ull) {
        const bool across = NOISE && TILE == 2 && s_tile != nullptr && p.obs_noise_sigma > 0.0f;      // pqx_step: the draws across the wave
        float zr[DN_OBS_DIM];
        if (across) draw_obs_noise_across(p, gid - li, step_count, 5u, done_mask, done && active, lane, s_tile, zr);
        if (done) {
            if (active && out.terminal_obs) {
#pragma unroll
                for (int k = 0; k < DN_OBS_DIM; ++k) out.terminal_obs[li * DN_OBS_DIM + k] = o[k];
            }
            reset_obs<R>(p, c, v.d_obs, o);                               // BaseAviary.py:318 before :617-658 (Q2)
            if (SPAWN && p.random_spawn) {                                // the body was loaded at this episode's spawn point
                double q[3];
                spawn_point(p, gid, step_count, q);
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k] = (float)((R)(float)q[k] * c.inv_dim[k]);
            }
            if (across) add_obs_noise_drawn(p, zr, o);
            else if (NOISE && p.obs_noise_sigma > 0.0f) add_obs_noise(p, gid, step_count, 5u, o);
            if (NORM) normalize_obs(rms, o);
        }
    }
    if (TILE == 2) { if (active) store_obs_direct(out.obs + li * DN_OBS_DIM, o); }
    else if (TILE == 1) tile_park(s_tile, lane, o);        // streamed out by the caller one step later
    else store_obs_tile(s_tile, out.obs, rows, lane, o);
}
template <typename R, bool NORM, bool NOISE, bool REW, int TILE = 0, bool SPAWN = false>
DN_DEV void report_phase(const DnParams &p, const DnConsts<R> &c, float *s_tile, const StepOut &out, const Flight<R> &fl,
                         const Verdict<R> &v, Observed<R> &ob, const unsigned long long gid, const unsigned long long step_count,
                         const unsigned li, const unsigned lane, const unsigned rows, const bool active,
                         float4 &G4, float4 &G5, StatAcc &acc, Rms &rms, RewNorm &rn)
{
    report_scalars<R, REW>(p, c, out, fl, v, ob.r_normal, ob.r_found32, li, lane, active, G4, G5, acc, rn);
    report_obs<R, NORM, NOISE, TILE, SPAWN>(p, c, s_tile, out, fl.truncated != 0, v, ob.o, gid, step_count, li, lane, rows, active, rms);
}

struct BlockState {
    float4 *g0, *g1, *g2, *g3, *g4, *g5, *g6, *g7;
};
DN_DEV BlockState block_state(const DnState &st, long long tile_base)
{   // uniform block bases (SGPR pairs); a lane adds its 32-bit offset
    BlockState b;
    b.g0 = st.g0 + tile_base; b.g1 = st.g1 + tile_base; b.g2 = st.g2 + tile_base; b.g3 = st.g3 + tile_base;
    b.g4 = st.g4 + tile_base; b.g5 = st.g5 + tile_base; b.g6 = st.g6 + tile_base;
    b.g7 = st.g7 + tile_base;      // allocated with Physics.PYB_DRAG only; never touched otherwise
    return b;
}

// thrust + physics of one step; the XOPT kernels take the float64 carriers and the optional force terms (N4)
template <typename R, bool NOISE, bool XOPT>
DN_DEV Flight<R> fly(const DnParams &p, unsigned long long gid, unsigned long long sc, const float4 A, const float4 G0, const float4 G1,
                     const float4 G2, const float4 G3, const float4 G7, float4 &rpm_now, double *pid_st = nullptr)
{
    if (XOPT) {
        Extras x;
        x.last = G7;
        PidCtx cx;
        cx.G0 = G0; cx.G1 = G1; cx.G2 = G2; cx.st = pid_st;
        const ThrustX th = thrust_phase_x<NOISE>(p, gid, sc, A, x, pid_st ? &cx : nullptr);
        rpm_now = make_float4((float)x.rpm[0], (float)x.rpm[1], (float)x.rpm[2], (float)x.rpm[3]);
        return physics_phase<R, ThrustX, true>(th, G0, G1, G2, G3, p.max_steps, &x);
    }
    const Thrust th = thrust_phase<NOISE>(p, gid, sc, A);
    return physics_phase<R>(th, G0, G1, G2, G3, p.max_steps);
}

// dn_step_sampled: SB3 DiagGaussianDistribution.sample / log_prob and the np.clip of collect_rollouts [3P-recall], drawn
// where the action is consumed: action = mean + exp(log_std) z, z ~ N(0,1) from the environment's Philox stream (seed,
// global drone id, the tile's vector-step counter, stream 9); the unclipped action and its log-probability go to the
// rollout buffer, the clipped one into the step.  Same expressions as dn_policy_sample_kernel (same bits).
// tanh(x) = 1 - 2 / (1 + e^{2x}): v_exp_f32 + v_rcp_f32 (absolute error ~1e-7; +-1 exactly once e^{2x} over- or underflows)
DN_DEV float tanh_squash(const float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);        // 2 log2(e)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// One drone's squashed-Gaussian draw (SB3 SAC Actor [3P-recall]): shared by dn_squashed_sample_kernel and the step kernels
// (dn_step_squashed), so that the two produce the same bits.  The log-probability costs four logf: only on request.
DN_DEV void squashed_draw(const float4 m, const float4 l, const float z[4], const bool want_lp, float a[4], float &lp)
{
    const float mu[4] = {m.x, m.y, m.z, m.w};
    const float ls[4] = {clipv(l.x, -20.0f, 2.0f), clipv(l.y, -20.0f, 2.0f), clipv(l.z, -20.0f, 2.0f), clipv(l.w, -20.0f, 2.0f)};
    lp = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = tanh_squash(mu[j] + expf(ls[j]) * z[j]);
    if (want_lp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) lp += (-0.5f * z[j] * z[j] - ls[j] - 0.91893853320467274178f) - logf(1.0f - a[j] * a[j] + 1e-6f);
    }
}
// m = the policy's mean row (PPO) or its mu row with l = the log_std row (SAC, io.sample_squash); taken by value so that a kernel
// that has just COMPUTED them (the fused policy + step launch, dn_fused.hip) hands them over without a trip through memory
DN_DEV float4 sample_action_from(const DnStepIO &io, const float4 m, const float4 l, const unsigned long long gid, const unsigned long long step,
                                 const long long i, const bool active)
{
    float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!io.sample_deterministic) noise4(io.sample_seed, gid, step, 9u, z);
    float a[4], lp = 0.0f;
    if (io.sample_squash) {
        // dn_step_squashed: SAC's Actor on the (mu | log_std) rows of dn_mlp_forward(arch = SAC) -- the expressions of
        // dn_squashed_sample_kernel (same bits); the squashed action is already inside the action box
        squashed_draw(m, l, z, io.logp_out != nullptr, a, lp);
    } else {
        const float mu[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = mu[j] + expf(io.log_std[j]) * z[j];
            lp += -0.5f * z[j] * z[j] - io.log_std[j] - 0.91893853320467274178f;
        }
    }
    if (active) {
        reinterpret_cast<float4 *>(io.act_out)[i] = make_float4(a[0], a[1], a[2], a[3]);
        if (io.logp_out) io.logp_out[i] = lp;
    }
    return make_float4(clipv(a[0], -1.0f, 1.0f), clipv(a[1], -1.0f, 1.0f), clipv(a[2], -1.0f, 1.0f), clipv(a[3], -1.0f, 1.0f));
}
DN_DEV float4 sample_action(const DnStepIO &io, const unsigned long long gid, const unsigned long long step, const long long i, const bool active)
{
    const float4 *rows = reinterpret_cast<const float4 *>(io.mean);
    if (io.sample_squash) return sample_action_from(io, rows[2 * i], rows[2 * i + 1], gid, step, i, active);
    return sample_action_from(io, rows[i], make_float4(0.0f, 0.0f, 0.0f, 0.0f), gid, step, i, active);
}

// -----------------------------------------------------------------------------------------------------
// One-wave kernels: all four phases on one wavefront, messages in registers.  Used where there are enough
// drones to fill the chip with whole steps (see dn_launch_step) and as the cross-check of the two-wave kernels.
// -----------------------------------------------------------------------------------------------------
// ONE = true is the single-step launch (dn_step): k_steps is the constant 1, and the kernel gets its own name in
// profiles (dn_step_many_*_kernel<..., true> = one control step per launch, <..., false> = k_arg steps per launch).
template <typename R, bool NORM, bool NOISE, bool ONE, bool XOPT, bool SAMPLE = false>
__global__ __launch_bounds__(DN_BLOCK) void dn_step_many_1w_kernel(const DnParams p, const DnStepIO io0, const int k_arg)
{
    const int k_steps = ONE ? 1 : k_arg;
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    const unsigned lane = threadIdx.x;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;       // inactive lanes shadow the last drone, never store
    __builtin_assume(li < DN_BLOCK);                    // lets the lane offset stay a 32-bit VGPR (saddr addressing)
    const long long i = tile_base + li;
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
    // issue every load up front (6 x 16 B state + 16 B action per lane), then stage the table
    constexpr bool sampled = SAMPLE;                      // dn_step_sampled: the action comes from the policy's mean (own instantiations:
                                                          // the sampler's Box-Muller code costs the plain single step 9 % if it is merely linked in)
    float4 A = sampled ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : act[li];
    float4 G0 = b.g0[li], G1 = b.g1[li], G2 = b.g2[li], G3 = b.g3[li], G4 = b.g4[li], G5 = b.g5[li];
    float4 G7 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (XOPT && p.drag) G7 = b.g7[li];
    stage_table<R>(p, s_tab);
    block_lds_barrier();
    const long long n = p.n, words = (p.n + 63) / 64;
    DnStatSlot slot0;
    if (ONE) slot0 = p.st.stats[blockIdx.x];                               // single-step launch: the whole slot now (see flush_stats_preloaded)
    const unsigned long long sc0 = ONE ? slot0.step_count : p.st.stats[blockIdx.x].step_count;      // this tile's vector-step counter
    if (sampled) A = sample_action(io0, gid, sc0, i, active);
    const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};   // waypoint 0: every reset measures against it
    StatAcc acc;
    Rms rms;
    if (NORM) load_rms_walk(p, tile_base, li, rms);
    RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
    if (XOPT && p.norm_rew) load_rewnorm(p, i, rn);
    double pid_st[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (XOPT && p.pid_mode) {
#pragma unroll
        for (int k = 0; k < 9; ++k) pid_st[k] = p.st.pid[(long long)k * p.n + i];
    }
#pragma clang loop unroll(disable)
    for (int t = 0; t < k_steps; ++t) {
        // prefetch the next step's action while this step computes
        const float4 A_next = sampled ? A : (act + (long long)(t + 1 < k_steps ? t + 1 : t) * n)[li];
        const StepOut out = block_out(io0, tile_base, (long long)t * n, (long long)t * words);
        const unsigned long long sc = sc0 + (unsigned long long)t;
        float4 rpm_now;
        const GateRow<R> row_e = load_gate_row<R>(s_tab, unpack_meta(G3.w).idx);
        Flight<R> fl = fly<R, NOISE, XOPT>(p, gid, sc, A, G0, G1, G2, G3, G7, rpm_now, XOPT ? pid_st : nullptr);
        const float4 G0e = G0, G3e = G3;
        const Verdict<R> v = rules_phase<R, XOPT>(p, c, s_tab, row_e, wp0, fl, G0e, G3e, b.g6, li, active, G0, G1, G2, G3, gid, sc);
        if (XOPT && p.drag) G7 = (v.terminated || fl.truncated) ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : rpm_now;   // BaseAviary.py:442,545
        attitude_phase<R>(fl);
        Observed<R> ob = observe_phase<R, NORM, NOISE>(p, c, s_tab, fl, G4, G5, gid, sc, rms);
        report_phase<R, NORM, NOISE, XOPT, 0, XOPT>(p, c, s_tile, out, fl, v, ob, gid, sc, li, lane, rows, active, G4, G5, acc, rms, rn);
        A = A_next;
    }
    if (ONE) flush_stats_preloaded(p, slot0, acc, sc0 + 1ull, lane, (long long)blockIdx.x);
    else flush_stats(p, acc, sc0 + (unsigned long long)k_steps, lane);
    if (NORM && active) store_rms_walk(p, tile_base, li, rms);
    if (XOPT && p.norm_rew && active) store_rewnorm(p, i, rn);
    if (XOPT && p.pid_mode && active) {
#pragma unroll
        for (int k = 0; k < 9; ++k) p.st.pid[(long long)k * p.n + i] = pid_st[k];
    }
    if (active) {
        b.g0[li] = G0; b.g1[li] = G1; b.g2[li] = G2; b.g3[li] = G3; b.g4[li] = G4; b.g5[li] = G5;
        if (XOPT && p.drag) b.g7[li] = G7;
    }
}

// -----------------------------------------------------------------------------------------------------
// dn_eval_kinematics: rows A5-A9 (+ A10/A11) of a control step on their own.  The rigid-body transition (A4,
// p.stepSimulation) is GIVEN -- pos, quat, vel, ang_v after the physics step, float64 as the reference holds them --
// and everything downstream of it runs through the SAME device functions as dn_step: rules_phase, attitude_phase,
// observe_phase, report_phase, with the same entry bookkeeping and the same auto-reset.  It exists so that the
// reference-generated fixtures that script a kinematic sequence (tests/golden/obs_pack.npz, script_*.npz: gimbal-lock
// attitudes, clip edges, gate passes, the last gate, corridor exits, truncation, the reset quirks) reach the HIP path
// directly (the CPU checker is not in between).
// -----------------------------------------------------------------------------------------------------
template <typename R, bool NORM>
__global__ __launch_bounds__(DN_BLOCK) void dn_eval_kinematics_kernel(const DnParams p, const DnStepIO io0, const double *__restrict__ kin)
{
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    const unsigned lane = threadIdx.x;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const long long i = tile_base + li;
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    float4 G0 = b.g0[li], G1 = b.g1[li], G2 = b.g2[li], G3 = b.g3[li], G4 = b.g4[li], G5 = b.g5[li];
    const double *kr = kin + i * 13;
    double kv[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) kv[k] = kr[k];
    stage_table<R>(p, s_tab);
    block_lds_barrier();
    const unsigned long long sc0 = p.st.stats[blockIdx.x].step_count;
    const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};
    StatAcc acc;
    Rms rms;
    if (NORM) load_rms(p, i, rms);
    RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
    const StepOut out = block_out(io0, tile_base, 0, 0);
    const GateRow<R> row_e = load_gate_row<R>(s_tab, unpack_meta(G3.w).idx);
    Flight<R> fl;
    flight_entry<R>(fl, G0, G2, G3, p.max_steps);
    fl.px = (R)kv[0]; fl.py = (R)kv[1]; fl.pz = (R)kv[2];
    fl.qx = (R)kv[3]; fl.qy = (R)kv[4]; fl.qz = (R)kv[5]; fl.qw = (R)kv[6];
    fl.vx = (float)kv[7]; fl.vy = (float)kv[8]; fl.vz = (float)kv[9];          // velocities leave physics_phase as float32 words
    fl.wx = (float)kv[10]; fl.wy = (float)kv[11]; fl.wz = (float)kv[12];
    const float4 G0e = G0, G3e = G3;
    const Verdict<R> v = rules_phase<R>(p, c, s_tab, row_e, wp0, fl, G0e, G3e, b.g6, li, active, G0, G1, G2, G3);
    attitude_phase<R>(fl);
    Observed<R> ob = observe_phase<R, NORM, false>(p, c, s_tab, fl, G4, G5, gid, sc0, rms);
    report_phase<R, NORM, false, false>(p, c, s_tile, out, fl, v, ob, gid, sc0, li, lane, rows, active, G4, G5, acc, rms, rn);
    flush_stats(p, acc, sc0 + 1ull, lane);
    if (NORM && active) store_rms(p, i, rms);
    if (active) {
        b.g0[li] = G0; b.g1[li] = G1; b.g2[li] = G2; b.g3[li] = G3; b.g4[li] = G4; b.g5[li] = G5;
    }
}

// -----------------------------------------------------------------------------------------------------
// Two-wave kernels: 128 threads = flight wave (threads 0..63) + report wave (64..127) over the same 64 drones,
// skewed by one step, ONE LDS-only barrier per step:
//
//   flight, iteration t:  thrust(t) | physics(t) | rules(t)    -> Flight(t), Verdict(t) into mail[t & 1]   == barrier t ==
//   report, iteration t:  mail[(t-1) & 1] -> observe(t-1) | report(t-1)                                     == barrier t ==
//   report, after the loop: observe(K-1) | report(K-1)
//
// The flight wave is the recurrence (state(t+1) needs state(t)); everything the report wave does is a side output
// of a step that is already decided, so it can trail by a step and the two waves never wait on each other inside
// a step.  Measured with s_memtime per phase (MI355X, 32768 drones): thrust 1180, physics 1860, rules 1510 cycles on
// the flight wave, observe 2290 + report 2140 on the report wave -- 4550 vs 4430 cycles per step, against 8980 on
// one wave.  The mail is double-buffered: iteration t+1 overwrites the buffer read in iteration t only after
// barrier t.  (With the thrust on the report wave and two barriers per step -- the first shape tried -- the report
// wave was busy 5610 cycles per step and the flight wave idle for 3040 of its 6410.)
// -----------------------------------------------------------------------------------------------------
template <typename R> struct Mail {       // LDS, field-major so that consecutive lanes hit consecutive banks
    R f64[DN_NMAIL64][DN_BLOCK];          // position (3), forward vector (3), Verdict.d_obs
    float4 f32[5][DN_BLOCK];              // 17 float32 fields + the two flag words: five 16-byte stores per lane
};
template <typename R> DN_DEV void post_mail(Mail<R> &m, unsigned lane, const Flight<R> &f, const Verdict<R> &v)
{
    const R x[DN_NMAIL64] = {f.px, f.py, f.pz, f.fwx, f.fwy, f.fwz, v.d_obs};
#pragma unroll
    for (int k = 0; k < DN_NMAIL64; ++k) m.f64[k][lane] = x[k];
    const int fb = f.idx_e | (f.just_found_e << 8) | (f.truncated << 9), vb = v.coll1 | (v.terminated << 1);
    m.f32[0][lane] = make_float4(f.vx, f.vy, f.vz, f.wx);
    m.f32[1][lane] = make_float4(f.wy, f.wz, f.vex, f.vey);
    m.f32[2][lane] = make_float4(f.vez, f.aex, f.aey, f.aez);
    m.f32[3][lane] = make_float4(f.d_e, f.dprev_e, __int_as_float(fb), __int_as_float(vb));
    m.f32[4][lane] = make_float4(f.roll_num32, f.roll_den32, f.pitch32, f.yaw32);
}
template <typename R> DN_DEV void take_mail(const Mail<R> &m, unsigned lane, Flight<R> &f, Verdict<R> &v)
{
    R x[DN_NMAIL64];
#pragma unroll
    for (int k = 0; k < DN_NMAIL64; ++k) x[k] = m.f64[k][lane];
    f.px = x[0]; f.py = x[1]; f.pz = x[2]; f.fwx = x[3]; f.fwy = x[4]; f.fwz = x[5];
    v.d_obs = x[6];
    f.qx = f.qy = f.qz = R(0.0); f.qw = R(1.0);      // the attitude itself stays on the flight wave
    const float4 a = m.f32[0][lane], b = m.f32[1][lane], c = m.f32[2][lane], d = m.f32[3][lane], e = m.f32[4][lane];
    f.roll_num32 = e.x; f.roll_den32 = e.y; f.pitch32 = e.z; f.yaw32 = e.w;
    f.vx = a.x; f.vy = a.y; f.vz = a.z; f.wx = a.w; f.wy = b.x; f.wz = b.y; f.vex = b.z; f.vey = b.w;
    f.vez = c.x; f.aex = c.y; f.aey = c.z; f.aez = c.w; f.d_e = d.x; f.dprev_e = d.y;
    const int bits = __float_as_int(d.z), vb = __float_as_int(d.w);
    f.idx_e = bits & 0xFF; f.just_found_e = (bits >> 8) & 1; f.truncated = (bits >> 9) & 1;
    v.coll1 = vb & 1; v.terminated = (vb >> 1) & 1;
}

// The thrust of a step depends on the action alone, not on the state: the report wave computes it one step AHEAD
// (thrust(t+1) during iteration t) and hands it over the same way, which moves a quarter of the flight wave's
// work off the recurrence.  Double-buffered like the mail: iteration t writes tmail[(t+1) & 1] while the flight
// wave reads tmail[t & 1].
template <typename R> struct ThrustMail {
    R v[4][DN_BLOCK];                     // Resultant: fz, tx, ty, zt
};
template <typename R> DN_DEV void post_thrust(ThrustMail<R> &m, unsigned lane, const Thrust &t)
{
    const Resultant<R> r = rotor_resultant<R>(t);
    m.v[0][lane] = r.fz; m.v[1][lane] = r.tx; m.v[2][lane] = r.ty; m.v[3][lane] = r.zt;
}
template <typename R> DN_DEV Resultant<R> take_thrust(const ThrustMail<R> &m, unsigned lane)
{
    Resultant<R> r;
    r.fz = m.v[0][lane]; r.tx = m.v[1][lane]; r.ty = m.v[2][lane]; r.zt = m.v[3][lane];
    return r;
}

template <typename R, bool NORM, bool NOISE, bool ONE, bool XOPT>
__global__ __launch_bounds__(2 * DN_BLOCK) void dn_step_many_2w_kernel(const DnParams p, const DnStepIO io0, const int k_arg)
{
    const int k_steps = ONE ? 1 : k_arg;
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    __shared__ Mail<R> mail[2];
    __shared__ __attribute__((aligned(16))) ThrustMail<R> tmail[2];
    constexpr bool THRUST_AHEAD = !XOPT;        // the XOPT thrust carries float64 forces and the rpm: it stays on the flight wave
    const unsigned lane = threadIdx.x & (DN_BLOCK - 1);
    const bool report_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) != 0;    // wave-uniform role
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const long long i = tile_base + li;
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    const long long n = p.n, words = (p.n + 63) / 64;
    const unsigned long long sc0 = p.st.stats[blockIdx.x].step_count;      // this tile's vector-step counter
    stage_table<R>(p, s_tab);
    if (report_wave) {
        float4 G4 = b.g4[li], G5 = b.g5[li];
        StatAcc acc;
        Rms rms;
        if (NORM) load_rms(p, i, rms);
        RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
        if (XOPT && p.norm_rew) load_rewnorm(p, i, rn);
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (THRUST_AHEAD) {
            A = act[li];
            const float4 A1 = (act + (long long)(k_steps > 1 ? 1 : 0) * n)[li];
            post_thrust<R>(tmail[0], lane, thrust_phase<NOISE>(p, gid, sc0, A));
            A = A1;
        }
        block_lds_barrier();                                               // P: table and thrust(0) published
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            if (THRUST_AHEAD && t + 1 < k_steps) {                         // thrust(t+1), for the flight wave's next iteration
                const float4 A_next = (act + (long long)(t + 2 < k_steps ? t + 2 : t + 1) * n)[li];
                post_thrust<R>(tmail[(t + 1) & 1], lane, thrust_phase<NOISE>(p, gid, sc0 + (unsigned long long)(t + 1), A));
                A = A_next;
            }
            if (t > 0) {                                                   // the step the flight wave finished last iteration
                const int u = t - 1;
                const unsigned long long sc = sc0 + (unsigned long long)u;
                // step u-1's observation tile was parked in LDS at the end of the previous iteration: fetch it now,
                // stream it to HBM after observe(u) -- its LDS round trip hides behind that phase
                TileRegs tile;
                if (u > 0) tile = tile_fetch(s_tile, lane);
                Flight<R> fl;
                Verdict<R> v;
                take_mail<R>(mail[u & 1], lane, fl, v);
                Observed<R> ob = observe_phase<R, NORM, NOISE>(p, c, s_tab, fl, G4, G5, gid, sc, rms);
                if (u > 0) tile_stream(tile, s_tile, io0.obs + ((long long)(u - 1) * n + tile_base) * DN_OBS_DIM, rows, lane);
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                report_phase<R, NORM, NOISE, XOPT, 1>(p, c, s_tile, out, fl, v, ob, gid, sc, li, lane, rows, active, G4, G5, acc, rms, rn);
            }
            if (t < k_steps) block_lds_barrier();                          // barrier t
        }
        {   // the last step's tile
            const TileRegs tile = tile_fetch(s_tile, lane);
            tile_stream(tile, s_tile, io0.obs + ((long long)(k_steps - 1) * n + tile_base) * DN_OBS_DIM, rows, lane);
        }
        flush_stats(p, acc, sc0 + (unsigned long long)k_steps, lane);
        if (NORM && active) store_rms(p, i, rms);
        if (XOPT && p.norm_rew && active) store_rewnorm(p, i, rn);
        if (active) { b.g4[li] = G4; b.g5[li] = G5; }
    } else {
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!THRUST_AHEAD) A = act[li];
        float4 G0 = b.g0[li], G1 = b.g1[li], G2 = b.g2[li], G3 = b.g3[li];
        float4 G7 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (XOPT && p.drag) G7 = b.g7[li];
        block_lds_barrier();                                               // P
        const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};   // waypoint 0: every reset measures against it
#pragma clang loop unroll(disable)
        for (int t = 0; t < k_steps; ++t) {
            float4 rpm_now = make_float4(0.0f, 0.0f, 0.0f, 0.0f), A_next = A;
            const GateRow<R> row_e = load_gate_row<R>(s_tab, unpack_meta(G3.w).idx);
            Flight<R> fl;
            if (THRUST_AHEAD) {
                fl = physics_phase<R, Resultant<R>>(take_thrust<R>(tmail[t & 1], lane), G0, G1, G2, G3, p.max_steps);
            } else {
                // prefetch the next step's action while this step computes
                A_next = (act + (long long)(t + 1 < k_steps ? t + 1 : t) * n)[li];
                fl = fly<R, NOISE, XOPT>(p, gid, sc0 + (unsigned long long)t, A, G0, G1, G2, G3, G7, rpm_now);
            }
            const float4 G0e = G0, G3e = G3;
            const Verdict<R> v = rules_phase<R>(p, c, s_tab, row_e, wp0, fl, G0e, G3e, b.g6, li, active, G0, G1, G2, G3);
            if (XOPT && p.drag) G7 = (v.terminated || fl.truncated) ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : rpm_now;
            attitude_phase<R>(fl);
            post_mail<R>(mail[t & 1], lane, fl, v);
            block_lds_barrier();                                           // barrier t
            A = A_next;
        }
        if (active) { b.g0[li] = G0; b.g1[li] = G1; b.g2[li] = G2; b.g3[li] = G3; }
        if (XOPT && p.drag && active) b.g7[li] = G7;
    }
}

// -----------------------------------------------------------------------------------------------------
// Three-wave kernel (fused launches without the XOPT options): the side work is cut once more, so
// that the recurrence (physics + rules) is alone on its wave and every SIMD of a CU has more than one wave's worth of
// independent instructions to pick from.  192 threads = flight (0..63) + report (64..127) + aux (128..191):
//
//   flight, iteration t:  thrust(t) from the aux wave | physics(t) | rules(t)            -> MailQ[t & 1]      == barrier t ==
//   aux,    iteration t:  thrust(t+1) -> tmail | MailQ[(t-1) & 1] -> attitude(t-1) | observe(t-1) -> MailA[(t-1) & 1]
//   report, iteration t:  MailA[(t-2) & 1] -> [normaliser] report(t-2), observation rows stored straight from registers
//
// The aux wave keeps prev_vel / prev_ang_v (the .xyz of g4 / g5: what the smoothness term reads), the report wave the
// Monitor accumulators (the .w of g4 / g5) and, with NORM, the normaliser statistics.  Same functions, same typed values
// across LDS, one spelled-out arithmetic sequence (see physics_phase): bit-identical to the other shapes.
// -----------------------------------------------------------------------------------------------------
template <typename R> struct MailQ {      // flight -> aux: the post-physics pose with the attitude quaternion
    R f64[8][DN_BLOCK];                   // position (3), quaternion (4), Verdict.d_obs
    float4 f32[4][DN_BLOCK];
};
template <typename R> DN_DEV void post_mailq(MailQ<R> &m, unsigned lane, const Flight<R> &f, const Verdict<R> &v)
{
    const R x[8] = {f.px, f.py, f.pz, f.qx, f.qy, f.qz, f.qw, v.d_obs};
#pragma unroll
    for (int k = 0; k < 8; ++k) m.f64[k][lane] = x[k];
    const int fb = f.idx_e | (f.just_found_e << 8) | (f.truncated << 9), vb = v.coll1 | (v.terminated << 1);
    m.f32[0][lane] = make_float4(f.vx, f.vy, f.vz, f.wx);
    m.f32[1][lane] = make_float4(f.wy, f.wz, f.vex, f.vey);
    m.f32[2][lane] = make_float4(f.vez, f.aex, f.aey, f.aez);
    m.f32[3][lane] = make_float4(f.d_e, f.dprev_e, __int_as_float(fb), __int_as_float(vb));
}
template <typename R> DN_DEV void take_mailq(const MailQ<R> &m, unsigned lane, Flight<R> &f, Verdict<R> &v)
{
    R x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = m.f64[k][lane];
    f.px = x[0]; f.py = x[1]; f.pz = x[2]; f.qx = x[3]; f.qy = x[4]; f.qz = x[5]; f.qw = x[6];
    v.d_obs = x[7];
    const float4 a = m.f32[0][lane], b = m.f32[1][lane], c = m.f32[2][lane], d = m.f32[3][lane];
    f.vx = a.x; f.vy = a.y; f.vz = a.z; f.wx = a.w; f.wy = b.x; f.wz = b.y; f.vex = b.z; f.vey = b.w;
    f.vez = c.x; f.aex = c.y; f.aey = c.z; f.aez = c.w; f.d_e = d.x; f.dprev_e = d.y;
    const int bits = __float_as_int(d.z), vb = __float_as_int(d.w);
    f.idx_e = bits & 0xFF; f.just_found_e = (bits >> 8) & 1; f.truncated = (bits >> 9) & 1;
    v.coll1 = vb & 1; v.terminated = (vb >> 1) & 1;
}
template <typename R> struct MailA {      // aux -> report: the observation, the reward candidates and what the verdict needs
    R f64[2][DN_BLOCK];                   // Observed.r_normal, Verdict.d_obs
    float4 f32[4][DN_BLOCK];              // o[0..11], then (o[12], r_found32, d_e, flag word)
};
template <typename R> DN_DEV void post_maila(MailA<R> &m, unsigned lane, const Flight<R> &f, const Verdict<R> &v, const Observed<R> &ob)
{
    m.f64[0][lane] = ob.r_normal; m.f64[1][lane] = v.d_obs;
    const int bits = f.idx_e | (f.truncated << 9) | (v.coll1 << 10) | (v.terminated << 11);
    m.f32[0][lane] = make_float4(ob.o[0], ob.o[1], ob.o[2], ob.o[3]);
    m.f32[1][lane] = make_float4(ob.o[4], ob.o[5], ob.o[6], ob.o[7]);
    m.f32[2][lane] = make_float4(ob.o[8], ob.o[9], ob.o[10], ob.o[11]);
    m.f32[3][lane] = make_float4(ob.o[12], ob.r_found32, f.d_e, __int_as_float(bits));
}
template <typename R> DN_DEV void take_maila(const MailA<R> &m, unsigned lane, Flight<R> &f, Verdict<R> &v, Observed<R> &ob)
{
    ob.r_normal = m.f64[0][lane]; v.d_obs = m.f64[1][lane];
    const float4 a = m.f32[0][lane], b = m.f32[1][lane], c = m.f32[2][lane], d = m.f32[3][lane];
    ob.o[0] = a.x; ob.o[1] = a.y; ob.o[2] = a.z; ob.o[3] = a.w; ob.o[4] = b.x; ob.o[5] = b.y; ob.o[6] = b.z; ob.o[7] = b.w;
    ob.o[8] = c.x; ob.o[9] = c.y; ob.o[10] = c.z; ob.o[11] = c.w; ob.o[12] = d.x;
    ob.r_found32 = d.y; f.d_e = d.z;
    const int bits = __float_as_int(d.w);
    f.idx_e = bits & 0xFF; f.truncated = (bits >> 9) & 1; v.coll1 = (bits >> 10) & 1; v.terminated = (bits >> 11) & 1;
    f.vex = f.vey = f.vez = f.aex = f.aey = f.aez = 0.0f;     // prev_vel / prev_ang_v live on the aux wave
}

// XOPT: the aux wave's thrust goes over as the float64 carriers with this step's rpm (what the ground-effect and drag
// terms read); the flight wave keeps last_clipped_action (g7), the report wave the reward normaliser.
template <bool XOPT> struct ThrustMailX { double v[9][DN_BLOCK]; };
template <> struct ThrustMailX<false> { double v[1][1]; };
DN_DEV void post_thrust_x(ThrustMailX<true> &m, unsigned lane, const ThrustX &t, const Extras &x)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { m.v[j][lane] = t.f[j]; m.v[5 + j][lane] = x.rpm[j]; }
    m.v[4][lane] = t.zt;
}
DN_DEV void take_thrust_x(const ThrustMailX<true> &m, unsigned lane, ThrustX &t, Extras &x)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { t.f[j] = m.v[j][lane]; x.rpm[j] = m.v[5 + j][lane]; }
    t.zt = m.v[4][lane];
}
DN_DEV void post_thrust_x(ThrustMailX<false> &, unsigned, const ThrustX &, const Extras &) {}
DN_DEV void take_thrust_x(const ThrustMailX<false> &, unsigned, ThrustX &, Extras &) {}
// the plain thrust mail is not needed with XOPT (LDS decides how many tiles a CU holds)
template <typename R, bool XOPT> struct ThrustMailP : ThrustMail<R> {};
template <typename R> struct ThrustMailP<R, true> { R v[1][1]; };
template <typename R> DN_DEV void post_thrust(ThrustMailP<R, true> &, unsigned, const Thrust &) {}
template <typename R> DN_DEV Resultant<R> take_thrust(const ThrustMailP<R, true> &, unsigned) { return Resultant<R>(); }

// NORM (per-drone observation normaliser): the aux wave hands over the raw observation and the report wave, which
// owns the statistics, normalises it (and then, for a finished drone, the reset observation: the reference's order).
template <typename R, bool NORM, bool NOISE, bool XOPT>
__global__ __launch_bounds__(3 * DN_BLOCK, NORM ? 2 : 3) void dn_step_many_3w_kernel(const DnParams p, const DnStepIO io0, const int k_steps)
{
    __shared__ ThrustMailX<XOPT> tmx[2];
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    __shared__ MailQ<R> mailq[2];
    __shared__ MailA<R> maila[2];
    __shared__ __attribute__((aligned(16))) ThrustMailP<R, XOPT> tmail[2];
    const unsigned lane = threadIdx.x & (DN_BLOCK - 1);
    // role of this wave: 0 flight, 1 aux, 2 report.  The order of the waves inside the workgroup decides which of them
    // end up sharing a SIMD when a CU holds two tiles (six waves on four SIMDs, dealt round-robin): with the waves
    // ordered flight, REPORT, aux the two heavy waves of one tile meet the light wave of the other (measured at 32768
    // drones: 1.48 us per step; flight, aux, report: 1.80 us, where a flight wave shares its SIMD with an aux wave).
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wv == 0 ? 0 : (wv == 1 ? 2 : 1);
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const long long i = tile_base + li;
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    const long long n = p.n, words = (p.n + 63) / 64;
    const unsigned long long sc0 = p.st.stats[blockIdx.x].step_count;
    stage_table<R>(p, s_tab);
    // every wave passes barrier P and the barriers of iterations 0 .. k_steps (k_steps + 1 of them)
    if (role == 0) {
        __builtin_amdgcn_s_setprio(3);                                     // the recurrence: first pick where it shares a SIMD
        float4 G0 = b.g0[li], G1 = b.g1[li], G2 = b.g2[li], G3 = b.g3[li];
        float4 G7 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (XOPT && p.drag) G7 = b.g7[li];
        block_lds_barrier();                                               // P
        const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            if (t < k_steps) {
                const GateRow<R> row_e = load_gate_row<R>(s_tab, unpack_meta(G3.w).idx);
                Flight<R> fl;
                float4 rpm_now = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (XOPT) {                                                // fly<R, NOISE, true> with its thrust half on the aux wave
                    Extras x;
                    ThrustX th;
                    take_thrust_x(tmx[t & 1], lane, th, x);
                    x.gnd = p.gnd; x.drag = p.drag; x.rpm_f32 = !p.rpm_actions;
                    x.damp = p.zero_damping ? 0.0 : 0.04;
                    x.last = G7;
                    rpm_now = make_float4((float)x.rpm[0], (float)x.rpm[1], (float)x.rpm[2], (float)x.rpm[3]);
                    fl = physics_phase<R, ThrustX, true>(th, G0, G1, G2, G3, p.max_steps, &x);
                } else {
                    fl = physics_phase<R, Resultant<R>>(take_thrust<R>(tmail[t & 1], lane), G0, G1, G2, G3, p.max_steps);
                }
                const float4 G0e = G0, G3e = G3;
                const Verdict<R> v = rules_phase<R>(p, c, s_tab, row_e, wp0, fl, G0e, G3e, b.g6, li, active, G0, G1, G2, G3);
                if (XOPT && p.drag) G7 = (v.terminated || fl.truncated) ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : rpm_now;
                post_mailq<R>(mailq[t & 1], lane, fl, v);
            }
            block_lds_barrier();                                           // barrier t
        }
        if (active) { b.g0[li] = G0; b.g1[li] = G1; b.g2[li] = G2; b.g3[li] = G3; }
        if (XOPT && p.drag && active) b.g7[li] = G7;
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(2);
        float4 P4 = b.g4[li], P5 = b.g5[li];                               // .xyz: prev_vel, prev_ang_v
        Rms rms;                                                           // never touched here: the observation leaves raw
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = act[li];
        {
            const float4 A1 = (act + (long long)(k_steps > 1 ? 1 : 0) * n)[li];
            if (XOPT) {
                Extras x;
                const ThrustX th = thrust_phase_x<NOISE>(p, gid, sc0, A, x);
                post_thrust_x(tmx[0], lane, th, x);
            } else {
                post_thrust<R>(tmail[0], lane, thrust_phase<NOISE>(p, gid, sc0, A));
            }
            A = A1;
        }
        block_lds_barrier();                                               // P: table and thrust(0) published
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            if (t + 1 < k_steps) {                                         // thrust(t+1), for the flight wave's next iteration
                const float4 A_next = (act + (long long)(t + 2 < k_steps ? t + 2 : t + 1) * n)[li];
                if (XOPT) {
                    Extras x;
                    const ThrustX th = thrust_phase_x<NOISE>(p, gid, sc0 + (unsigned long long)(t + 1), A, x);
                    post_thrust_x(tmx[(t + 1) & 1], lane, th, x);
                } else {
                    post_thrust<R>(tmail[(t + 1) & 1], lane, thrust_phase<NOISE>(p, gid, sc0 + (unsigned long long)(t + 1), A));
                }
                A = A_next;
            }
            if (t > 0) {                                                   // the step the flight wave finished last iteration
                const int u = t - 1;
                Flight<R> fl;
                Verdict<R> v;
                take_mailq<R>(mailq[u & 1], lane, fl, v);
                attitude_phase<R>(fl);
                const Observed<R> ob = observe_phase<R, false, NOISE>(p, c, s_tab, fl, P4, P5, gid, sc0 + (unsigned long long)u, rms);
                post_maila<R>(maila[u & 1], lane, fl, v, ob);
                // prev_vel / prev_ang_v: _update_state_post_step (skipped on a terminated step, quirk Q5), zero after a reset
                // (a terminated step is a finished one, so the old copies never survive: one select per word)
                {
                    const bool fin = v.terminated != 0 || fl.truncated != 0;
                    P4 = fin ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fl.vex, fl.vey, fl.vez, 0.0f);
                    P5 = fin ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fl.aex, fl.aey, fl.aez, 0.0f);
                }
            }
            block_lds_barrier();                                           // barrier t
        }
        if (active) {
            float *g4 = reinterpret_cast<float *>(b.g4 + li), *g5 = reinterpret_cast<float *>(b.g5 + li);
            g4[0] = P4.x; g4[1] = P4.y; g4[2] = P4.z; g5[0] = P5.x; g5[1] = P5.y; g5[2] = P5.z;
        }
    } else {
        float4 G4 = b.g4[li], G5 = b.g5[li];                               // .w: Monitor return / length
        StatAcc acc;
        Rms rms;
        if (NORM) load_rms(p, i, rms);
        RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
        if (XOPT && p.norm_rew) load_rewnorm(p, i, rn);
        block_lds_barrier();                                               // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps + 1; ++t) {
            if (t > 1) {                                                   // the step the aux wave finished last iteration
                const int u = t - 2;
                const unsigned long long sc = sc0 + (unsigned long long)u;
                Flight<R> fl;
                Verdict<R> v;
                Observed<R> ob;
                take_maila<R>(maila[u & 1], lane, fl, v, ob);
                if (NORM) normalize_obs(rms, ob.o);                            // the step observation (= terminal_observation)
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                report_phase<R, NORM, NOISE, XOPT, 2>(p, c, s_tile, out, fl, v, ob, gid, sc, li, lane, rows, active, G4, G5, acc, rms, rn);
            }
            if (t <= k_steps) block_lds_barrier();                         // barrier t
        }
        flush_stats(p, acc, sc0 + (unsigned long long)k_steps, lane);
        if (NORM && active) store_rms(p, i, rms);
        if (XOPT && p.norm_rew && active) store_rewnorm(p, i, rn);
        if (active) {
            reinterpret_cast<float *>(b.g4 + li)[3] = G4.w;
            reinterpret_cast<float *>(b.g5 + li)[3] = G5.w;
            }
    }
}

// -----------------------------------------------------------------------------------------------------
// Four-wave kernel (fused launches of the plain configuration): the recurrence itself is cut in two.
//
// In the three-wave kernel the flight wave carries physics + rules, ~420 vector instructions per step that nothing else
// can start without, and sets the pace (1.27 us per step at 32 768 drones with the SIMDs issuing 84 % of the time).  The
// rigid-body step is two halves that share only their inputs (physics_linear / physics_angular, see there), and the rules
// read the new POSITION only (no ground-contact term here), so the recurrence runs on two waves that exchange one small
// message per step:
//
//   L (linear, rules)    thrust fz(t) | physics_linear(t) with the entry attitude | rules_verdict, rules_commit (t)   -> MailL[t & 1]
//   A (angular, pose)    torques(t)   | physics_angular(t) | attitude of the new pose (Euler terms, forward vector)  -> MailG[t & 1]
//   Q (thrust, observe)  thrust(t+1) -> tmail | MailL, MailG [(t-1) & 1] -> observation columns, reward candidates (t-1)  -> MailA[(t-1) & 1]
//   X (report)           MailA[(t-2) & 1] -> [normaliser] report(t-2), observation rows
//                                                                                                  == barrier t ==
// L needs A's new quaternion of step t-1 (a float4) and applies its own verdict's reset to it; A needs L's verdict of step
// t-1 (one flag) to reset its half of the body.  ~255 instructions each instead of ~420 on one wave.  Same device functions,
// same typed values across LDS, one spelled-out arithmetic sequence: bit-identical to the other shapes
// (test_kernel_shapes_are_bit_identical under DN_WAVES=4).
// -----------------------------------------------------------------------------------------------------
template <typename R> struct MailL {      // L -> Q (and its flag word -> A): the linear half of the new state and the verdict
    R f64[4][DN_BLOCK];                   // position (3), Verdict.d_obs
    float4 f32[2][DN_BLOCK];              // (vx, vy, vz, d_e), (vex, vey, vez, dprev_e)
    int flags[DN_BLOCK];                  // idx_e | just_found_e << 8 | truncated << 9 | coll1 << 10 | terminated << 11
};
template <typename R> struct MailG {      // A -> L (qnew) and A -> Q (the rest): the angular half and the attitude read-outs
    R col[3][DN_BLOCK];                   // attitude_column of the new attitude as it goes back to HBM (float32 words), before any reset:
                                          // all the linear half reads of it (round 6; it used to take the four words and redo quat_terms)
    R fw[3][DN_BLOCK];                    // forward vector of the new pose
    float4 eul[DN_BLOCK];                 // roll_num32, roll_den32, pitch32, yaw32
    float4 w[DN_BLOCK];                   // new angular velocity (float32 state words)
    float4 we[DN_BLOCK];                  // entry angular velocity (prev_ang_v of the smoothness term)
};

// NW = 5 (normaliser on, round 3): the report wave is cut in two.  A wave issues one instruction per ~7.5 cycles however idle its SIMD
// is (profiles/r03_valu_rates.txt), so the launch runs at the pace of the role with the most instructions per step -- with the
// normaliser that was X: the step's scalars plus 247 float64 instructions of Welford update per step plus, on the 86 % of tile-steps
// where a drone finishes, the masked second pass for the reset observation.  X keeps the scalars (reward select, Monitor, statistics),
// a fifth wave N takes MailA's observation through the normaliser and out (report_obs): same functions, same values.
// Timing builds (-DDN_MW_STAMP=<tile>; never shipped): the cycle counter of every role of one tile before and after the barrier of
// iterations 8 .. 55 of a fused launch, read back through dn_debug_mw_stamps (profiles/mw_stamps.py).
#ifdef DN_MW_STAMP
__device__ long long g_mw_stamp[8][48][2];
#define MW_MARK(k) do { if (lane == 0 && blockIdx.x == DN_MW_STAMP && t >= 8 && t < 56) \
        g_mw_stamp[role][t - 8][k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MW_MARK(k) do { } while (0)
#endif
#define MW_BARRIER() do { MW_MARK(0); block_lds_barrier(); MW_MARK(1); } while (0)
#ifdef DN_ROLE_MARK      // ISA inspection only (profiles/role_isa.sh): a comment line at the top of every role's K-step loop
#define MW_ROLE_MARK(name) asm volatile("; DN_ROLE_LOOP " name)
#else
#define MW_ROLE_MARK(name) do { } while (0)
#endif
#ifdef DN_MW_STAMP
__device__ long long g_mw_edge[16][8];    // per role: cycles at entry, after barrier P, --, at exit; wall clock (100 MHz) at entry / exit; HW_ID
#define MW_EDGE(k) do { if (lane == 0 && (blockIdx.x == DN_MW_STAMP || blockIdx.x == DN_MW_STAMP + 256)) { const int r_ = role + (blockIdx.x == DN_MW_STAMP ? 0 : 8); \
        g_mw_edge[r_][k] = (long long)__builtin_readcyclecounter(); \
        if ((k) == 0) { g_mw_edge[r_][4] = (long long)wall_clock64(); g_mw_edge[r_][6] = (long long)__builtin_amdgcn_s_getreg(63492); } \
        if ((k) == 3) g_mw_edge[r_][5] = (long long)wall_clock64(); } } while (0)
#else
#define MW_EDGE(k) do { } while (0)
#endif

template <typename R, bool NORM, bool NOISE, int NW>
DN_DEV void step_many_4w_body(const DnParams &p, const DnStepIO &io0, const int k_steps)
{
    static_assert(NW == 4 || (NW == 5 && NORM), "the fifth wave is the normaliser's");
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ MailL<R> maill[2];
    __shared__ __attribute__((aligned(16))) MailG<R> mailg[2];
    __shared__ MailA<R> maila[2];
    __shared__ __attribute__((aligned(16))) ThrustMail<R> tmail[2];
    const unsigned lane = threadIdx.x & (DN_BLOCK - 1);
    // roles: 0 L, 1 A, 2 Q, 3 X.  Which waves of two co-resident tiles end up sharing a SIMD is decided by the order of the waves
    // inside the workgroup (a workgroup's waves go to the SIMDs round-robin); swept at 32 768 drones (two tiles per CU) over four
    // orders x four permutations for the second tile of a CU (tiles num_cus .. 2 num_cus - 1 land beside tiles 0 .. num_cus - 1): without the
    // normaliser L A X Q with waves 2 <-> 0, 3 <-> 1 swapped in the second tile (1.23 us per step; worst order 1.45), with the
    // normaliser L X A Q in both (1.76; the mirrored orders 1.77).  One tile per CU (<= 16 384 drones) does not care: 1.01-1.02.
    // the thrust chain sits on the observation wave -- except with noise and no normaliser: the observation draws (13 normals a
    // step) already make that wave the longest, so the thrust and its action-noise draw go to the report wave (32 768 drones:
    // 4.47 us per step against 4.74; with the normaliser, which also lives on the report wave, 3.85 against 3.75)
    constexpr bool THRUST_ON_Q = !NOISE || NORM;
    const int wv0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int second_tile = (int)((blockIdx.x / (unsigned)p.num_cus) & 1u);      // the workgroup that lands beside tile blockIdx.x - num_cus on its CU
    const int wv = (!NORM && second_tile) ? (wv0 ^ 2) : wv0;
#ifndef DN_5W_ORDER_A
#define DN_5W_ORDER_A "LQANX"
#define DN_5W_ORDER_B "ANLQX"
#endif
    int role;
    if (NW == 5) {
        // five waves land on four SIMDs as 0 1 2 3 0; the second tile of a CU (tiles 256 .. 511 beside 0 .. 255) takes another order, so
        // that the heavy waves of the two tiles pair with light ones of the other: SIMD 0: L1 X1 A2 X2, 1: Q1 N2, 2: A1 L2, 3: N1 Q2.
        // Swept at 32 768 drones (us per step): this order 1.43, X L A N Q | Q N A L X 1.45, the same order in both tiles 1.56-1.80
        // (profiles/r03_notes.md); the four-wave kernel 1.76.
        constexpr char oa[6] = DN_5W_ORDER_A, ob[6] = DN_5W_ORDER_B;
        const char ch = second_tile ? ob[wv0] : oa[wv0];
        role = ch == 'L' ? 0 : ch == 'A' ? 1 : ch == 'Q' ? 2 : ch == 'X' ? 3 : 4;
    } else
        role = NORM ? (wv == 0 ? 0 : (wv == 1 ? 3 : (wv == 2 ? 1 : 2)))            // L X A Q
                    : (wv == 0 ? 0 : (wv == 1 ? 1 : (wv == 2 ? 3 : 2)));           // L A X Q
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const long long i = tile_base + li;
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    const long long n = p.n, words = (p.n + 63) / 64;
    const unsigned long long sc0 = p.st.stats[blockIdx.x].step_count;
    MW_EDGE(0);
    stage_table<R>(p, s_tab);
    // every wave passes barrier P and the barriers of iterations 0 .. k_steps (k_steps + 1 of them)
#ifndef DN_5W_PRIO
#define DN_5W_PRIO "22311"               // s_setprio of the roles L A Q X N: Q, the role every barrier waits for, first (round 6: 33200 -> 22311 is 3 % of the step)
#endif
    {
        constexpr char pr[6] = DN_5W_PRIO;
        switch (pr[role] - '0') {                                          // s_setprio takes an immediate
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
        }
    }
    if (role == 0) {
        float4 G0 = b.g0[li], G2 = b.g2[li], G3 = b.g3[li];
        AttCol<R> col = attitude_column<R>(b.g1[li]);                      // the thrust direction of the entry attitude; from step 1 on A mails it
        block_lds_barrier(); MW_EDGE(1);                                   // P
        const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};
        bool done_prev = false;
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            MW_ROLE_MARK("L");
            if (t < k_steps) {
                if (t > 0) {                                               // entry attitude: A's step t-1, level after my reset of t-1
                    const MailG<R> &mp = mailg[(t - 1) & 1];
                    const R c0 = mp.col[0][lane], c1 = mp.col[1][lane], c2 = mp.col[2][lane];
                    col.r02 = done_prev ? R(0.0) : c0; col.r12 = done_prev ? R(0.0) : c1; col.r22 = done_prev ? R(1.0) : c2;
                }
                const GateRow<R> row_e = load_gate_row<R>(s_tab, unpack_meta(G3.w).idx);
                const R fz = tmail[t & 1].v[0][lane];
                const Lin<R> lin = physics_linear_col<R>(G0, G2, col, fz, R(0.0), R(0.0), R(0.0), false);
                Flight<R> fl;
                flight_entry<R>(fl, G0, G2, G3, p.max_steps);
                fl.px = lin.px; fl.py = lin.py; fl.pz = lin.pz;
                fl.vx = (float)lin.vx; fl.vy = (float)lin.vy; fl.vz = (float)lin.vz;
                fl.qx = fl.qy = fl.qz = R(0.0); fl.qw = R(1.0);            // the attitude belongs to A (no ground-contact term here)
                fl.wx = fl.wy = fl.wz = 0.0f;
                RulesMid<R> m;
                const Verdict<R> v = rules_verdict<R>(p, c, s_tab, row_e, fl, G3, m);
                MailL<R> &ml = maill[t & 1];
                ml.f64[0][lane] = fl.px; ml.f64[1][lane] = fl.py; ml.f64[2][lane] = fl.pz; ml.f64[3][lane] = v.d_obs;
                ml.f32[0][lane] = make_float4(fl.vx, fl.vy, fl.vz, fl.d_e);
                ml.f32[1][lane] = make_float4(fl.vex, fl.vey, fl.vez, fl.dprev_e);
                ml.flags[lane] = fl.idx_e | (fl.just_found_e << 8) | (fl.truncated << 9) | (v.coll1 << 10) | (v.terminated << 11);
                float4 S0, S1, S2, S3;
                rules_commit<R>(c, wp0, fl, m, G0, G3, b.g6, li, active, S0, S1, S2, S3);
                G0 = S0; G2 = S2; G3.w = S3.w;
                done_prev = v.terminated != 0 || fl.truncated != 0;
            }
            MW_BARRIER();                                                  // barrier t
        }
        MW_EDGE(2);
        if (active) {
            b.g0[li] = G0; b.g2[li] = G2;
            reinterpret_cast<float *>(b.g3 + li)[3] = G3.w;
        }
    } else if (role == 1) {
        float4 G1 = b.g1[li], G3 = b.g3[li];                               // G3.xyz: angular velocity (the .w belongs to L)
        // btMatrix3x3::setRotation's terms of the entry attitude: formed ONCE per step, at the end of the step before -- they give the
        // thrust direction L needs for its next step (mailed) and they start this wave's own next step (same expressions, same bits as
        // attitude_column / physics_angular_pre on the state word)
        QuatTerms<R> qt = quat_terms<R>((R)G1.x, (R)G1.y, (R)G1.z, (R)G1.w);
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            MW_ROLE_MARK("A");
            if (t > 0) {                                                   // L's verdict of step t-1: a finished drone restarts level, at rest
                const int fb = maill[(t - 1) & 1].flags[lane];
                const bool fin = (((fb >> 9) | (fb >> 11)) & 1) != 0;
                if (__ballot(fin) != 0ull) {                               // wave-uniform: most wave-steps of a long flight skip the selects
                    if (fin) { G1 = make_float4(0.0f, 0.0f, 0.0f, 1.0f); G3.x = G3.y = G3.z = 0.0f; qt = quat_terms_identity<R>(); }
                }
            }
            if (t < k_steps) {
                const R tx = tmail[t & 1].v[1][lane], ty = tmail[t & 1].v[2][lane], zt = tmail[t & 1].v[3][lane];
                const Ang<R> ang = physics_angular_post<R>(physics_angular_pre_terms<R>(G1, G3, qt), tx, ty, zt);
                Flight<R> fl;
                fl.qx = ang.qx; fl.qy = ang.qy; fl.qz = ang.qz; fl.qw = ang.qw;
                attitude_phase<R>(fl);
                MailG<R> &mg = mailg[t & 1];
                const float4 qn = make_float4((float)ang.qx, (float)ang.qy, (float)ang.qz, (float)ang.qw);
                const float4 wn = make_float4((float)ang.wx, (float)ang.wy, (float)ang.wz, 0.0f);
                qt = quat_terms<R>((R)qn.x, (R)qn.y, (R)qn.z, (R)qn.w);   // of the state word, as the next step reads it
                const AttCol<R> cn = attitude_column_terms<R>(qn, qt);
                mg.col[0][lane] = cn.r02; mg.col[1][lane] = cn.r12; mg.col[2][lane] = cn.r22;
                mg.fw[0][lane] = fl.fwx; mg.fw[1][lane] = fl.fwy; mg.fw[2][lane] = fl.fwz;
                mg.eul[lane] = make_float4(fl.roll_num32, fl.roll_den32, fl.pitch32, fl.yaw32);
                mg.w[lane] = wn;
                mg.we[lane] = make_float4(G3.x, G3.y, G3.z, 0.0f);
                G1 = qn; G3.x = wn.x; G3.y = wn.y; G3.z = wn.z;
            }
            MW_BARRIER();                                                  // barrier t
        }
        MW_EDGE(2);
        if (active) {
            b.g1[li] = G1;
            float *g3 = reinterpret_cast<float *>(b.g3 + li);
            g3[0] = G3.x; g3[1] = G3.y; g3[2] = G3.z;
        }
    } else if (role == 2) {
        float4 P4 = b.g4[li], P5 = b.g5[li];                               // .xyz: prev_vel, prev_ang_v
        Rms rms;                                                           // never touched here: the observation leaves raw
        // the thrust chain lives here: with it on the report wave (its stores, the episode statistics, the normaliser) that wave was
        // the longest of the four and set the pace (16 384 drones: 1.17 us per step against 1.02 with it here)
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (THRUST_ON_Q) {
            A = act[li];
            const float4 A1 = (act + (long long)(k_steps > 1 ? 1 : 0) * n)[li];
            post_thrust<R>(tmail[0], lane, thrust_phase<NOISE>(p, gid, sc0, A));
            A = A1;
        }
        block_lds_barrier(); MW_EDGE(1);                                   // P: table and thrust(0) published
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            MW_ROLE_MARK("Q");
            if (THRUST_ON_Q && t + 1 < k_steps) {                          // thrust(t+1), for the next iteration of L and A
                const float4 A_next = (act + (long long)(t + 2 < k_steps ? t + 2 : t + 1) * n)[li];
                post_thrust<R>(tmail[(t + 1) & 1], lane, thrust_phase<NOISE>(p, gid, sc0 + (unsigned long long)(t + 1), A));
                A = A_next;
            }
            if (t > 0) {                                                   // the step L and A finished last iteration
                const int u = t - 1;
                const MailL<R> &ml = maill[u & 1];
                const MailG<R> &mg = mailg[u & 1];
                Flight<R> fl;
                Verdict<R> v;
                fl.px = ml.f64[0][lane]; fl.py = ml.f64[1][lane]; fl.pz = ml.f64[2][lane]; v.d_obs = ml.f64[3][lane];
                const float4 a0 = ml.f32[0][lane], a1 = ml.f32[1][lane];
                fl.vx = a0.x; fl.vy = a0.y; fl.vz = a0.z; fl.d_e = a0.w;
                fl.vex = a1.x; fl.vey = a1.y; fl.vez = a1.z; fl.dprev_e = a1.w;
                const int fb = ml.flags[lane];
                fl.idx_e = fb & 0xFF; fl.just_found_e = (fb >> 8) & 1; fl.truncated = (fb >> 9) & 1;
                v.coll1 = (fb >> 10) & 1; v.terminated = (fb >> 11) & 1;
                const float4 e = mg.eul[lane], wn = mg.w[lane], we = mg.we[lane];
                fl.qx = fl.qy = fl.qz = R(0.0); fl.qw = R(1.0);            // not read by the observation / reward (the attitude's read-outs came over)
                fl.fwx = mg.fw[0][lane]; fl.fwy = mg.fw[1][lane]; fl.fwz = mg.fw[2][lane];
                fl.roll_num32 = e.x; fl.roll_den32 = e.y; fl.pitch32 = e.z; fl.yaw32 = e.w;
                fl.wx = wn.x; fl.wy = wn.y; fl.wz = wn.z;
                fl.aex = we.x; fl.aey = we.y; fl.aez = we.z;
                const Observed<R> ob = observe_phase<R, false, NOISE>(p, c, s_tab, fl, P4, P5, gid, sc0 + (unsigned long long)u, rms);
                post_maila<R>(maila[u & 1], lane, fl, v, ob);
                // prev_vel / prev_ang_v: _update_state_post_step (skipped on a terminated step, quirk Q5), zero after a reset
                // (a terminated step is a finished one, so the old copies never survive: one select per word)
                {
                    const bool fin = v.terminated != 0 || fl.truncated != 0;
                    P4 = fin ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fl.vex, fl.vey, fl.vez, 0.0f);
                    P5 = fin ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fl.aex, fl.aey, fl.aez, 0.0f);
                }
            }
            MW_BARRIER();                                                  // barrier t
        }
        MW_EDGE(2);
        if (active) {
            float *g4 = reinterpret_cast<float *>(b.g4 + li), *g5 = reinterpret_cast<float *>(b.g5 + li);
            g4[0] = P4.x; g4[1] = P4.y; g4[2] = P4.z; g5[0] = P5.x; g5[1] = P5.y; g5[2] = P5.z;
        }
    } else if (role == 4) {
        // ---- N (NW == 5): observation -> normaliser -> rows, the reset observation of a finished drone included
        Rms rms;
        load_rms(p, i, rms);                                               // (walked bases, load_rms_walk, cost this kernel 2 %: profiles/r06_notes.md section 12)
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps + 1; ++t) {
            MW_ROLE_MARK("N");
            if (t > 1) {
                const int u = t - 2;
                Flight<R> fl;
                Verdict<R> v;
                Observed<R> ob;
                take_maila<R>(maila[u & 1], lane, fl, v, ob);
                normalize_obs(rms, ob.o);                                  // the step observation (= terminal_observation)
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                report_obs<R, NORM, NOISE, 2>(p, c, nullptr, out, fl.truncated != 0, v, ob.o, gid, sc0 + (unsigned long long)u, li, lane, rows, active, rms);
            }
            if (t <= k_steps) MW_BARRIER();                                // barrier t
        }
        MW_EDGE(2);
        if (active) store_rms(p, i, rms);
    } else {
        float4 G4 = b.g4[li], G5 = b.g5[li];                               // .w: Monitor return / length
        StatAcc acc;
        Rms rms;
        if (NORM && NW == 4) load_rms_walk(p, tile_base, li, rms);         // four waves with the normaliser: 195 -> 149 registers, three tiles per CU resident
        RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!THRUST_ON_Q) {
            A = act[li];
            const float4 A1 = (act + (long long)(k_steps > 1 ? 1 : 0) * n)[li];
            post_thrust<R>(tmail[0], lane, thrust_phase<NOISE>(p, gid, sc0, A));
            A = A1;
        }
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps + 1; ++t) {
            MW_ROLE_MARK("X");
            if (!THRUST_ON_Q && t + 1 < k_steps) {                         // thrust(t+1), for the next iteration of L and A
                const float4 A_next = (act + (long long)(t + 2 < k_steps ? t + 2 : t + 1) * n)[li];
                post_thrust<R>(tmail[(t + 1) & 1], lane, thrust_phase<NOISE>(p, gid, sc0 + (unsigned long long)(t + 1), A));
                A = A_next;
            }
            if (t > 1) {                                                   // the step Q finished last iteration
                const int u = t - 2;
                const unsigned long long sc = sc0 + (unsigned long long)u;
                Flight<R> fl;
                Verdict<R> v;
                Observed<R> ob;
                take_maila<R>(maila[u & 1], lane, fl, v, ob);
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                if (NW == 5) report_scalars<R, false>(p, c, out, fl, v, ob.r_normal, ob.r_found32, li, lane, active, G4, G5, acc, rn);
                else {
                    if (NORM) normalize_obs(rms, ob.o);                        // the step observation (= terminal_observation)
                    report_phase<R, NORM, NOISE, false, 2>(p, c, nullptr, out, fl, v, ob, gid, sc, li, lane, rows, active, G4, G5, acc, rms, rn);
                }
            }
            if (t <= k_steps) MW_BARRIER();                                // barrier t
        }
        MW_EDGE(2);
        flush_stats(p, acc, sc0 + (unsigned long long)k_steps, lane);
        if (NORM && NW == 4 && active) store_rms_walk(p, tile_base, li, rms);
        if (active) {
            reinterpret_cast<float *>(b.g4 + li)[3] = G4.w;
            reinterpret_cast<float *>(b.g5 + li)[3] = G5.w;
        }
    }
    MW_EDGE(3);
}

template <typename R, bool NORM, bool NOISE>
__global__ __launch_bounds__(4 * DN_BLOCK, 2) void dn_step_many_4w_kernel(const DnParams p, const DnStepIO io0, const int k_steps)
{
    step_many_4w_body<R, NORM, NOISE, 4>(p, io0, k_steps);
}
// five waves: two tiles per CU are ten waves on four SIMDs, i.e. FOUR waves on one of them -- 128 registers a wave
// (three tiles, where noise keeps this shape selected: fifteen waves, four on three of the SIMDs, at the same 128 registers)
template <typename R, bool NOISE>
#ifndef DN_5W_WAVES_PER_EU
#define DN_5W_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(5 * DN_BLOCK) __attribute__((amdgpu_waves_per_eu(DN_5W_WAVES_PER_EU, DN_5W_WAVES_PER_EU))) void dn_step_many_5w_kernel(const DnParams p, const DnStepIO io0, const int k_steps)
{
    step_many_4w_body<R, true, NOISE, 5>(p, io0, k_steps);
}

// -----------------------------------------------------------------------------------------------------
// Role-pipelined kernel (round 4; fused launches of the plain configuration with the normaliser and without noise): eight roles.
//
// Per-role stamps of the four- / five-wave kernels (profiles/mw_stamps.py, profiles/r04_stamps_5w_4w.txt, r04_stamps_rp8.txt) showed what paces them: not the
// recurrence waves L and A but Q (thrust of step t + 1, observation row and reward candidates of step t - 1: busy 2 450 cycles of a
// 2 620-cycle iteration at one tile per CU, 3 000-3 400 at two), then N and L.  Here the step is cut into EIGHT roles, and the one that
// carries the rules keeps only what the NEXT step's flags need:
//
//   T   thrust(t + 1)                                                                          -> tmail[(t + 1) & 1]
//   L   the linear recurrence: physics_linear(t) on A's attitude column, collision / corridor / gate flags, auto-reset of position,
//       velocity, gate index, step counter; squared distances only (see found_from_d2)           -> RMailL[t & 1]
//   A   the angular recurrence: physics_angular(t), thrust direction of the new attitude           -> RMailA[t & 1]
//   E   step t - 1: Euler terms, forward vector, orientation term, observation columns 3 4 5 9 10 11   -> RMailE[(t - 1) & 1]
//   Q   step t - 1: the distance bookkeeping (d, d_prev, just_found; quirks Q1-Q3), observation columns 0 1 2 6 7 8 12, the entry-state
//       reward terms, prev_vel / prev_ang_v                                                        -> RMailC[(t - 1) & 1]
//   X   step t - 2: reward assembly and select, Monitor, statistics, scalar outputs (without the normaliser also the rows)
//   N1  step t - 2: normaliser + rows, columns 0..6        N2  columns 7..12     (terminal / reset observation of a finished drone included)
//                                                                                                  == one LDS-only barrier per iteration ==
// Same device functions / same spelled-out expressions, same typed values across LDS: bit-identical to every other shape.
// -----------------------------------------------------------------------------------------------------
template <typename R> struct RMailL {      // L -> A (flags), E (position, flags), Q (everything)
    R f64[5][DN_BLOCK];                    // new position (3), squared distance after the step, squared distance after the reset
    float4 f32[DN_BLOCK];                  // new velocity (float32 state words), flag word
};
// flag word: idx_e [0:8) | truncated << 9 | coll1 << 10 | terminated << 11 | found_now << 12; Q adds pen_lin << 16 | pen_ang << 17
template <typename R> struct RMailA {      // A -> L (thrust direction of the new attitude), E (q, w), Q (we)
    R q[4][DN_BLOCK];                      // the new attitude in R (what attitude_phase reads)
    R col[3][DN_BLOCK];                    // attitude_column of the new attitude as it goes back to HBM (float32 words)
    float4 w[DN_BLOCK];                    // new angular velocity (float32 state words)
    float4 we[DN_BLOCK];                   // entry angular velocity (prev_ang_v of the smoothness term)
};
template <typename R> struct RMailE {      // E -> X (orientation term), N / X (columns 3 4 5 9 10 11)
    float4 oa[DN_BLOCK];                   // o3 o4 o5 o9
    float4 ob[DN_BLOCK];                   // o10 o11, orientation term (int bits), --
};
template <typename R> struct RMailC {      // Q -> X (reward terms, flags), N (d_obs, flags), N / X (columns 0 1 2 6 7 8 12)
    R r[4][DN_BLOCK];                      // RewardPre: r0, s_lin, s_ang; Verdict.d_obs
    float4 oa[DN_BLOCK];                   // o0 o1 o2 o6
    float4 ob[DN_BLOCK];                   // o7 o8 o12, flag word
};
struct __attribute__((packed, aligned(4))) ObsTri { float x, y, z; };

// found_now of the NEXT step from the squared distance: the reference tests the stored distance, (R)(float)sqrt(d2) <= threshold
// (PBDroneEnv.py:539 on the float32 state word).  Away from the radius the squared compare decides the same way -- the float32 rounding
// moves d by 6e-8 relative, the band below is 1e-6 --, and inside the band (never, in practice: one test in ~1e6 per unit of band) the
// stored form itself is evaluated, so the flag is the stored form's bit for bit.
template <typename R>
DN_DEV bool found_from_d2(const DnConsts<R> &c, const R d2)
{
    bool found = d2 < c.thr2;
    if (__builtin_expect(__ballot(fabs(d2 - c.thr2) <= c.thr2 * R(1e-6)) != 0ull, 0))
        found = (R)(float)FM<R>::sqrt0(d2) <= c.threshold;
    return found;
}

// report_obs for the columns [K0, K1) of the row (TILE = 2 form: straight from the lane's registers)
template <typename R, bool NORM, int K0, int K1>
DN_DEV void report_obs_cols(const DnParams &p, const DnConsts<R> &c, const StepOut &out, const bool done, const R d_obs, float *o,
                            const unsigned li, const bool active, Rms &rms)
{
    if (__ballot(done && active) != 0ull) {
        if (done) {
            if (active && out.terminal_obs) {
#pragma unroll
                for (int k = K0; k < K1; ++k) out.terminal_obs[li * DN_OBS_DIM + k] = o[k];
            }
#pragma unroll
            for (int k = K0; k < K1; ++k)                                 // reset_obs<R>: BaseAviary.py:318 before :617-658 (Q2)
                o[k] = k < 12 ? c.reset_obs32[k < 12 ? k : 0] : (p.include_distance ? (float)(d_obs * c.inv_max_target_dist) : 0.0f);
            if (NORM) normalize_obs_cols<K0, K1>(rms, o);
        }
    }
    if (active) {
        float *row = out.obs + li * DN_OBS_DIM;
        static_assert((K0 == 0 && K1 == 7) || (K0 == 7 && K1 == DN_OBS_DIM) || (K0 == 0 && K1 == DN_OBS_DIM), "column split of the row stores");
        if (K0 == 0 && K1 == DN_OBS_DIM) store_obs_direct(row, o);
        else if (K0 == 0) {
            *reinterpret_cast<ObsQuad *>(row) = ObsQuad{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<ObsTri *>(row + 4) = ObsTri{o[4], o[5], o[6]};
        } else {
            row[7] = o[7];
            *reinterpret_cast<ObsQuad *>(row + 8) = ObsQuad{o[8], o[9], o[10], o[11]};
            row[12] = o[12];
        }
    }
}

#ifndef DN_RP_ORDER_A
#define DN_RP_ORDER_A "LAEQTXMN"
#define DN_RP_ORDER_B "EQLAMNTX"
#endif
#ifndef DN_RP_PRIO
#define DN_RP_PRIO "33211000"            // s_setprio of the roles L A T E Q X N1 N2
#endif
#ifndef DN_RP_PRIO_B
#define DN_RP_PRIO_B DN_RP_PRIO          // ... in the workgroup that arrives second on its CU (the SIMD arbiter favours the older one)
#endif
// Two tiles share a CU, and the SIMD arbiter picks by priority, then AGE: left alone, the workgroup that arrived first runs at its
// uncontended pace and the second one on what is left (stamps: 2 150 against 3 300-5 400 cycles per iteration), finishes long after it and
// runs its tail alone with the SIMDs mostly idle.  The two tiles therefore take turns at the higher priority, in slices of the shader
// clock (both read the same counter): each is the favoured one half of the time and both finish together.
#ifndef DN_RP_SLICE_BITS
#define DN_RP_SLICE_BITS 0                // measured: no gain (profiles/r04_notes.md); kept for experiments
#endif
#if DN_RP_SLICE_BITS > 0
#define RP_TAKE_TURNS() do { if ((((unsigned)(__builtin_readcyclecounter() >> DN_RP_SLICE_BITS) ^ (unsigned)second_tile) & 1u) != 0u) \
        __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); } while (0)
#else
#define RP_TAKE_TURNS() do { } while (0)
#endif
template <typename R>
DN_DEV void step_many_rp_body(const DnParams &p, const DnStepIO &io0, const int k_steps)
{
    constexpr bool NORM = true;                            // the eight-role cut exists with the normaliser only (a six-role form without it lost to
                                                           // the four-wave kernel at every fleet size, 32 768 drones: 1.68 against 1.40 us per step, and was removed)
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) RMailL<R> maill[2];
    __shared__ __attribute__((aligned(16))) RMailA<R> maila[2];
    __shared__ __attribute__((aligned(16))) RMailE<R> maile[2];
    __shared__ __attribute__((aligned(16))) RMailC<R> mailc[2];
    __shared__ __attribute__((aligned(16))) ThrustMail<R> tmail[2];
    constexpr bool NOISE = false;
    const unsigned lane = threadIdx.x & (DN_BLOCK - 1);
    const int wv0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int second_tile = (int)((blockIdx.x / (unsigned)p.num_cus) & 1u);
    // roles: 0 L, 1 A, 2 T, 3 E, 4 Q, 5 X, 6 N1 ('M'), 7 N2 ('N').  A workgroup's waves go to the SIMDs round-robin (wave w -> SIMD w % 4), so
    // SIMD s carries waves s and s + 4 of both tiles of its CU; the second tile takes another order so that the heavy pairs spread out.
    constexpr char oa[9] = DN_RP_ORDER_A, ob[9] = DN_RP_ORDER_B;
    const char ch = second_tile ? ob[wv0] : oa[wv0];
    const int role = ch == 'L' ? 0 : ch == 'A' ? 1 : ch == 'T' ? 2 : ch == 'E' ? 3 : ch == 'Q' ? 4 : ch == 'X' ? 5 : ch == 'M' ? 6 : 7;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + tile_base + li);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    const long long n = p.n, words = (p.n + 63) / 64;
    const unsigned long long sc0 = p.st.stats[blockIdx.x].step_count;
    const bool seg_track = p.cylinder && !p.circle;
    MW_EDGE(0);
    {
        constexpr char pra[9] = DN_RP_PRIO, prb[9] = DN_RP_PRIO_B;
        switch ((second_tile ? prb[role] : pra[role]) - '0') {             // s_setprio takes an immediate
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
        }
    }
    stage_table<R>(p, s_tab);
    // every wave passes barrier P and the barriers of iterations 0 .. k_steps (k_steps + 1 of them); iteration k_steps + 1 has none
    if (role == 0) {
        // ---- L: the linear half of the recurrence and the flags.  State: position, velocity (float32 words), gate index, step counter,
        // next step's found_now, the stale _current_position (quirk Q3; g6 in registers for the launch, written back only if it changed).
        float4 G0 = b.g0[li], G2 = b.g2[li];
        const float4 G1 = b.g1[li], G3 = b.g3[li];
        AttCol<R> col = attitude_column<R>(G1);
        int steps = unpack_meta(G3.w).steps, idx = unpack_meta(G3.w).idx;
        bool found_now = (R)G0.w <= c.threshold;                           // PBDroneEnv.py:539 on the stored distance
        float g6x = 0.0f, g6y = 0.0f, g6z = 0.0f;
        bool g6_own = false;                                               // the registers hold _current_position (else: still in HBM)
        block_lds_barrier(); MW_EDGE(1);                                   // P
        const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};
        bool done_prev = false;
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            RP_TAKE_TURNS();
            if (t < k_steps) {
                if (t > 0) {                                               // thrust direction: A's step t-1, level after my reset of t-1
                    const RMailA<R> &ma = maila[(t - 1) & 1];
                    const R c0 = ma.col[0][lane], c1 = ma.col[1][lane], c2 = ma.col[2][lane];
                    col.r02 = done_prev ? R(0.0) : c0; col.r12 = done_prev ? R(0.0) : c1; col.r22 = done_prev ? R(1.0) : c2;
                }
                const GateRow<R> row_e = load_gate_row<R>(s_tab, idx);
                const R fz = tmail[t & 1].v[0][lane];
                const Lin<R> lin = physics_linear_col<R>(G0, G2, col, fz, R(0.0), R(0.0), R(0.0), false);
                const R px = lin.px, py = lin.py, pz = lin.pz;
                // rules_verdict's flags (no ground-contact term in this kernel: r22 of the new attitude is not needed)
                const bool truncated = p.max_steps <= steps;               // PBDroneEnv.py:444-454 on the un-incremented _steps
                const bool coll1 = collision_common<R>(p, c, px, py, pz, R(1.0)) ||
                                   (seg_track && outside_segment_corridor_row<R>(c, row_e, px, py, pz));
                const bool last_gate = idx + 1 == p.num_waypoints;
                const int idx_e = idx, steps_e = steps;
                bool terminated;
                if (coll1) terminated = true;                              // :489-490
                else if (found_now) {
                    idx += 1;
                    if (last_gate) terminated = true;                      // :542-546
                    else terminated = seg_track && outside_segment_corridor<R>(c, s_tab, px, py, pz, idx);
                } else terminated = false;
                // _update_state_post_step's distance, squared (Q takes the root)
                R d2n = R(0.0);
                if (!terminated) {
                    steps += 1;
                    R wx = row_e.wp[0], wy = row_e.wp[1], wz = row_e.wp[2];
                    if (idx != idx_e) {                                    // a gate was passed this step: the next waypoint (rare)
                        const R *wp = s_tab + idx * DN_T_STRIDE;
                        wx = wp[0]; wy = wp[1]; wz = wp[2];
                    }
                    const R ex = wx - px, ey = wy - py, ez = wz - pz;
                    d2n = FM<R>::fma(ez, ez, FM<R>::fma(ey, ey, ex * ex));
                }
                const bool done = terminated || truncated;
                const float nvx = (float)lin.vx, nvy = (float)lin.vy, nvz = (float)lin.vz;
                float4 S0 = make_float4((float)px, (float)py, (float)pz, 0.0f), S2 = make_float4(nvx, nvy, nvz, 0.0f);
                R d2r = R(0.0);
                if (__ballot(done) != 0ull) {                              // rules_commit's reset of the body (wave-uniform skip)
                    if (done) {
                        R cpx, cpy, cpz;
                        if (!terminated) { cpx = px; cpy = py; cpz = pz; }            // post-step ran: it is the new position
                        else if (__builtin_expect(steps_e > 0, 1)) { cpx = G0.x; cpy = G0.y; cpz = G0.z; }
                        else if (g6_own) { cpx = g6x; cpy = g6y; cpz = g6z; }
                        else { const float4 G6 = b.g6[li]; cpx = G6.x; cpy = G6.y; cpz = G6.z; }
                        if (!(terminated && steps_e == 0)) { g6x = (float)cpx; g6y = (float)cpy; g6z = (float)cpz; g6_own = true; }
                        S0 = make_float4((float)c.spawn[0], (float)c.spawn[1], (float)c.spawn[2], 0.0f);
                        S2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        const R ex = cpx - wp0[0], ey = cpy - wp0[1], ez = cpz - wp0[2];
                        d2r = FM<R>::fma(ez, ez, FM<R>::fma(ey, ey, ex * ex));     // :651, squared
                        idx = 0; steps = 0;
                    }
                }
                RMailL<R> &ml = maill[t & 1];
                ml.f64[0][lane] = px; ml.f64[1][lane] = py; ml.f64[2][lane] = pz; ml.f64[3][lane] = d2n; ml.f64[4][lane] = d2r;
                ml.f32[lane] = make_float4(nvx, nvy, nvz, __int_as_float(idx_e | ((int)truncated << 9) | ((int)coll1 << 10) |
                                                                          ((int)terminated << 11) | ((int)found_now << 12)));
                found_now = found_from_d2<R>(c, done ? d2r : d2n);
                G0 = S0; G2 = S2;
                done_prev = done;
            }
            MW_BARRIER();                                                  // barrier t
        }
        if (active) {
            float *g0 = reinterpret_cast<float *>(b.g0 + li), *g2 = reinterpret_cast<float *>(b.g2 + li);
            g0[0] = G0.x; g0[1] = G0.y; g0[2] = G0.z; g2[0] = G2.x; g2[1] = G2.y; g2[2] = G2.z;
            if (g6_own) { float *g6f = reinterpret_cast<float *>(b.g6 + li); g6f[0] = g6x; g6f[1] = g6y; g6f[2] = g6z; }
        }
    } else if (role == 1) {
        // ---- A: the angular half of the recurrence
        float4 G1 = b.g1[li], G3 = b.g3[li];                               // G3.xyz: angular velocity (the .w belongs to Q)
        QuatTerms<R> qt = quat_terms<R>((R)G1.x, (R)G1.y, (R)G1.z, (R)G1.w);   // formed once per step: see the five-wave kernel's A
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            RP_TAKE_TURNS();
            if (t > 0) {                                                   // L's verdict of step t-1: a finished drone restarts level, at rest
                const int fb = __float_as_int(maill[(t - 1) & 1].f32[lane].w);
                const bool fin = (((fb >> 9) | (fb >> 11)) & 1) != 0;
                if (__ballot(fin) != 0ull) {
                    if (fin) { G1 = make_float4(0.0f, 0.0f, 0.0f, 1.0f); G3.x = G3.y = G3.z = 0.0f; qt = quat_terms_identity<R>(); }
                }
            }
            if (t < k_steps) {
                const R tx = tmail[t & 1].v[1][lane], ty = tmail[t & 1].v[2][lane], zt = tmail[t & 1].v[3][lane];
                const Ang<R> ang = physics_angular_post<R>(physics_angular_pre_terms<R>(G1, G3, qt), tx, ty, zt);
                RMailA<R> &ma = maila[t & 1];
                const float4 qn = make_float4((float)ang.qx, (float)ang.qy, (float)ang.qz, (float)ang.qw);
                const float4 wn = make_float4((float)ang.wx, (float)ang.wy, (float)ang.wz, 0.0f);
                qt = quat_terms<R>((R)qn.x, (R)qn.y, (R)qn.z, (R)qn.w);   // of the state word, as the next step reads it
                const AttCol<R> cn = attitude_column_terms<R>(qn, qt);     // what physics_linear(t + 1) reads of the state word
                ma.q[0][lane] = ang.qx; ma.q[1][lane] = ang.qy; ma.q[2][lane] = ang.qz; ma.q[3][lane] = ang.qw;
                ma.col[0][lane] = cn.r02; ma.col[1][lane] = cn.r12; ma.col[2][lane] = cn.r22;
                ma.w[lane] = wn;
                ma.we[lane] = make_float4(G3.x, G3.y, G3.z, 0.0f);
                G1 = qn; G3.x = wn.x; G3.y = wn.y; G3.z = wn.z;
            }
            MW_BARRIER();                                                  // barrier t
        }
        if (active) {
            b.g1[li] = G1;
            float *g3 = reinterpret_cast<float *>(b.g3 + li);
            g3[0] = G3.x; g3[1] = G3.y; g3[2] = G3.z;
        }
    } else if (role == 2) {
        // ---- T: the action chain, one step ahead
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = act[li];
        {
            const float4 A1 = (act + (long long)(k_steps > 1 ? 1 : 0) * n)[li];
            post_thrust<R>(tmail[0], lane, thrust_phase<NOISE>(p, gid, sc0, A));
            A = A1;
        }
        block_lds_barrier(); MW_EDGE(1);                                   // P: table and thrust(0) published
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            RP_TAKE_TURNS();
            if (t + 1 < k_steps) {                                         // thrust(t+1), for the next iteration of L and A
                const float4 A_next = (act + (long long)(t + 2 < k_steps ? t + 2 : t + 1) * n)[li];
                post_thrust<R>(tmail[(t + 1) & 1], lane, thrust_phase<NOISE>(p, gid, sc0 + (unsigned long long)(t + 1), A));
                A = A_next;
            }
            MW_BARRIER();                                                  // barrier t
        }
    } else if (role == 3) {
        // ---- E: everything that reads the new attitude, one step behind A
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            RP_TAKE_TURNS();
            if (t > 0) {
                const int u = t - 1;
                const RMailA<R> &ma = maila[u & 1];
                const RMailL<R> &ml = maill[u & 1];
                Flight<R> fl;
                fl.qx = ma.q[0][lane]; fl.qy = ma.q[1][lane]; fl.qz = ma.q[2][lane]; fl.qw = ma.q[3][lane];
                const float4 wn = ma.w[lane];
                fl.wx = wn.x; fl.wy = wn.y; fl.wz = wn.z;
                fl.px = ml.f64[0][lane]; fl.py = ml.f64[1][lane]; fl.pz = ml.f64[2][lane];
                const int fb = __float_as_int(ml.f32[lane].w);
                fl.idx_e = fb & 0xFF;
                attitude_phase<R>(fl);
                float o[DN_OBS_DIM];
                observe_columns_att<R>(fl, o);
                const int ori = reward_orientation<R>(p, s_tab, fl, ((fb >> 12) & 1) != 0, fl.idx_e + 1 == p.num_waypoints);
                RMailE<R> &me = maile[u & 1];
                me.oa[lane] = make_float4(o[3], o[4], o[5], o[9]);
                me.ob[lane] = make_float4(o[10], o[11], __int_as_float(ori), 0.0f);
            }
            MW_BARRIER();                                                  // barrier t
        }
    } else if (role == 4) {
        // ---- Q: the distance bookkeeping and everything that reads it, one step behind L.  State: d, d_prev (float32 words), just_found,
        // copies of the gate index / step counter (for the meta word), the entry velocity, prev_vel / prev_ang_v.
        float4 P4 = b.g4[li], P5 = b.g5[li];                               // .xyz: prev_vel, prev_ang_v
        const float4 G0 = b.g0[li], G2 = b.g2[li], G3 = b.g3[li];
        float d = G0.w, dprev = G2.w, vex = G2.x, vey = G2.y, vez = G2.z;
        int steps = unpack_meta(G3.w).steps, idx = unpack_meta(G3.w).idx, jf = unpack_meta(G3.w).just_found;
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps; ++t) {
            RP_TAKE_TURNS();
            if (t > 0) {
                const int u = t - 1;
                const RMailL<R> &ml = maill[u & 1];
                Flight<R> fl;
                fl.px = ml.f64[0][lane]; fl.py = ml.f64[1][lane]; fl.pz = ml.f64[2][lane];
                const R d2n = ml.f64[3][lane], d2r = ml.f64[4][lane];
                const float4 a0 = ml.f32[lane];
                const int fb = __float_as_int(a0.w);
                fl.vx = a0.x; fl.vy = a0.y; fl.vz = a0.z;
                fl.d_e = d; fl.dprev_e = dprev; fl.just_found_e = jf;
                fl.vex = vex; fl.vey = vey; fl.vez = vez;
                fl.idx_e = fb & 0xFF; fl.truncated = (fb >> 9) & 1;
                const bool coll1 = ((fb >> 10) & 1) != 0, terminated = ((fb >> 11) & 1) != 0, found_now = ((fb >> 12) & 1) != 0;
                const bool done = terminated || fl.truncated != 0;
                const float4 we = maila[u & 1].we[lane];
                fl.aex = we.x; fl.aey = we.y; fl.aez = we.z;
                float o[DN_OBS_DIM];
                observe_columns_lin<R>(p, c, fl, o);
                const RewardPre<R> pre = reward_entry<R>(p, c, fl, P4, P5);
                // rules_verdict / rules_commit's distance bookkeeping (the roots of L's squared distances)
                const bool last_gate = fl.idx_e + 1 == p.num_waypoints;
                R d_post = (R)d;                                           // Verdict.d_obs: what the reset observation shows (quirk Q2)
                if (!terminated) { d_post = FM<R>::sqrt0(d2n); steps += 1; }
                if (!coll1) {
                    dprev = d;
                    if (found_now) { idx += 1; if (!last_gate) jf = 1; } else jf = 0;
                }
                d = (float)d_post;
                if (__ballot(done) != 0ull) {
                    if (done) {
                        d = (float)FM<R>::sqrt0(d2r);                      // :651
                        dprev = d;                                         // :652
                        idx = 0; steps = 0; jf = 0;
                    }
                }
                RMailC<R> &mc = mailc[u & 1];
                mc.r[0][lane] = pre.r0; mc.r[1][lane] = pre.s_lin; mc.r[2][lane] = pre.s_ang; mc.r[3][lane] = d_post;
                mc.oa[lane] = make_float4(o[0], o[1], o[2], o[6]);
                mc.ob[lane] = make_float4(o[7], o[8], o[12], __int_as_float(fb | ((int)pre.pen_lin << 16) | ((int)pre.pen_ang << 17)));
                // prev_vel / prev_ang_v: _update_state_post_step (skipped on a terminated step, quirk Q5), zero after a reset
                // (a terminated step is a finished one, so the old copies never survive: one select per word)
                P4 = done ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fl.vex, fl.vey, fl.vez, 0.0f);
                P5 = done ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fl.aex, fl.aey, fl.aez, 0.0f);
                vex = done ? 0.0f : fl.vx; vey = done ? 0.0f : fl.vy; vez = done ? 0.0f : fl.vz;      // the next step's entry velocity
            }
            MW_BARRIER();                                                  // barrier t
        }
        if (active) {
            float *g0 = reinterpret_cast<float *>(b.g0 + li), *g2 = reinterpret_cast<float *>(b.g2 + li), *g3 = reinterpret_cast<float *>(b.g3 + li);
            float *g4 = reinterpret_cast<float *>(b.g4 + li), *g5 = reinterpret_cast<float *>(b.g5 + li);
            g0[3] = d; g2[3] = dprev; g3[3] = pack_meta(steps, idx, jf);
            g4[0] = P4.x; g4[1] = P4.y; g4[2] = P4.z; g5[0] = P5.x; g5[1] = P5.y; g5[2] = P5.z;
        }
    } else if (role == 5) {
        // ---- X: the scalars of step t - 2 (and, without the normaliser, its observation rows)
        float4 G4 = b.g4[li], G5 = b.g5[li];                               // .w: Monitor return / length
        StatAcc acc;
        RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps + 1; ++t) {
            RP_TAKE_TURNS();
            if (t > 1) {
                const int u = t - 2;
                const RMailE<R> &me = maile[u & 1];
                const RMailC<R> &mc = mailc[u & 1];
                Flight<R> fl;
                Verdict<R> v;
                const float4 cb = mc.ob[lane], eb = me.ob[lane];
                const int fb = __float_as_int(cb.w);
                fl.idx_e = fb & 0xFF; fl.truncated = (fb >> 9) & 1;
                v.coll1 = (fb >> 10) & 1; v.terminated = (fb >> 11) & 1; v.d_obs = R(0.0);
                const bool found_now = ((fb >> 12) & 1) != 0;
                // report_scalars re-forms found_now from the entry distance: hand it one that decides the same way
                fl.d_e = found_now ? 0.0f : FLT_MAX;
                RewardPre<R> pre;
                pre.r0 = mc.r[0][lane]; pre.s_lin = mc.r[1][lane]; pre.s_ang = mc.r[2][lane];
                pre.pen_lin = ((fb >> 16) & 1) != 0; pre.pen_ang = ((fb >> 17) & 1) != 0;
                pre.found_now = found_now;
                pre.last_gate = fl.idx_e + 1 == p.num_waypoints;
                R r_normal;
                float r_found32;
                reward_assemble<R>(pre, __float_as_int(eb.z), r_normal, r_found32);
                fl.vex = fl.vey = fl.vez = fl.aex = fl.aey = fl.aez = 0.0f;    // prev_vel / prev_ang_v live on Q
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                report_scalars<R, false>(p, c, out, fl, v, r_normal, r_found32, li, lane, active, G4, G5, acc, rn);
            }
            if (t <= k_steps) MW_BARRIER();                                // barrier t
        }
        flush_stats(p, acc, sc0 + (unsigned long long)k_steps, lane);
        if (active) {
            reinterpret_cast<float *>(b.g4 + li)[3] = G4.w;
            reinterpret_cast<float *>(b.g5 + li)[3] = G5.w;
        }
    } else if (NORM && role == 6) {
        // ---- N1: columns 0..6 of step t - 2 through the normaliser and out
        Rms rms;
        load_rms_cols<0, 7>(p, tile_base, li, rms);
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps + 1; ++t) {
            RP_TAKE_TURNS();
            if (t > 1) {
                const int u = t - 2;
                const RMailC<R> &mc = mailc[u & 1];
                const int fb = __float_as_int(mc.ob[lane].w);
                const float4 ea = maile[u & 1].oa[lane], ca = mc.oa[lane];
                float o[DN_OBS_DIM] = {ca.x, ca.y, ca.z, ea.x, ea.y, ea.z, ca.w, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                normalize_obs_cols<0, 7>(rms, o);                          // the step observation (= terminal_observation)
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                report_obs_cols<R, true, 0, 7>(p, c, out, (((fb >> 9) | (fb >> 11)) & 1) != 0, R(0.0), o, li, active, rms);
            }
            if (t <= k_steps) MW_BARRIER();                                // barrier t
        }
        if (active) store_rms_cols<0, 7, true>(p, tile_base, li, rms);
    } else if (NORM) {
        // ---- N2: columns 7..12
        Rms rms;
        load_rms_cols<7, DN_OBS_DIM>(p, tile_base, li, rms);
        block_lds_barrier(); MW_EDGE(1);                                   // P
#pragma clang loop unroll(disable)
        for (int t = 0; t <= k_steps + 1; ++t) {
            RP_TAKE_TURNS();
            if (t > 1) {
                const int u = t - 2;
                const RMailC<R> &mc = mailc[u & 1];
                const float4 cb = mc.ob[lane];
                const int fb = __float_as_int(cb.w);
                const R d_obs = mc.r[3][lane];
                const float4 ea = maile[u & 1].oa[lane], eb = maile[u & 1].ob[lane];
                float o[DN_OBS_DIM] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, cb.x, cb.y, ea.w, eb.x, eb.y, cb.z};
                normalize_obs_cols<7, DN_OBS_DIM>(rms, o);
                const StepOut out = block_out(io0, tile_base, (long long)u * n, (long long)u * words);
                report_obs_cols<R, true, 7, DN_OBS_DIM>(p, c, out, (((fb >> 9) | (fb >> 11)) & 1) != 0, d_obs, o, li, active, rms);
            }
            if (t <= k_steps) MW_BARRIER();                                // barrier t
        }
        if (active) store_rms_cols<7, DN_OBS_DIM, false>(p, tile_base, li, rms);
    }
    MW_EDGE(3);
}

// sixteen waves of two tiles on four SIMDs: four waves a SIMD, 128 registers a wave
template <typename R>
__global__ __launch_bounds__(8 * DN_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void dn_step_many_rp8_kernel(const DnParams p, const DnStepIO io0, const int k_steps)
{
    step_many_rp_body<R>(p, io0, k_steps);
}

// -----------------------------------------------------------------------------------------------------
// Single-step kernel on three waves, cut by DEPENDENCY (dn_step / dn_step_sampled at small fleets).
//
// A lone control step has no "previous step" to overlap with, so the skewed pipelines above buy nothing for it; what
// a step does have is three chains that only meet at three points.  192 threads = three waves over the same 64 drones:
//
//   X (thrust, rewards)   [sample] thrust A1-A3 -> tmail  |B1| entry-state reward terms           |B2| forward vector, orientation term
//                                                                                  |B3| A7 select, Monitor, statistics, scalar outputs; g4 g5
//   P (position, rules)   loads g0 g1 g2 g3, table        |B1| fz -> physics_linear -> pos, vel -> pmail  |B2| rules_verdict -> vmail
//                                                                                  |B3| rules_commit; g0 g2 g3.w g6
//   Q (attitude, obs)     loads g0 g1 g3 [statistics]     |B1| tx ty zt -> physics_angular -> quat -> qmail  |B2| attitude, observation
//                                                              columns [noise, normaliser]    |B3| terminal / reset observation, obs rows; g1 g3.xyz
//
// The linear half of Bullet's step needs only the thrust direction (third column of R), the angular half only the
// torques: they run side by side, and the collision / gate rules start as soon as the new position exists -- they do not
// wait for the quaternion update.  Every value is computed by the same device function, with the same spelled-out
// arithmetic, as in the one-wave kernel: the bits are the same (test_kernel_shapes_are_bit_identical under DN_WAVES=3).
// Not built for the XOPT options or the ground-contact term (rules would need the new attitude): those keep one wave.
// -----------------------------------------------------------------------------------------------------
template <typename R> struct PosMail {
    R p[3][DN_BLOCK];                     // new position
    float4 v[DN_BLOCK];                   // new velocity as it goes back to HBM (float32)
};
template <typename R> struct VerdictMail {
    R d_obs[DN_BLOCK];
    int flags[DN_BLOCK];                  // coll1 | terminated << 1
};
// Timing builds (-DDN_PQX_STAMP=<tile>; never shipped): the cycle counter of every role of one tile at the marks of pqx_step, read back
// through dn_debug_pqx_stamps (profiles/r03_pqx_stamps.txt).
#ifdef DN_PQX_STAMP
__device__ long long g_pqx_stamp[3][16];
#define PQX_MARK(k) do { if (lane == 0 && tile == DN_PQX_STAMP) { g_pqx_stamp[role][k] = (long long)__builtin_readcyclecounter(); \
        if ((k) == 0 || (k) == 7) g_pqx_stamp[role][8 + (k) / 7] = (long long)wall_clock64(); } } while (0)
#else
#define PQX_MARK(k) do { } while (0)
#endif
// The LDS of one three-wave single step (one 64-drone tile): the table and the four mails.  A struct, so that a kernel that runs the
// step as its TAIL (dn_fused.hip: the policy network's workgroup steps the drones it has just evaluated) can place it in LDS it
// already owns.
template <typename R> struct __attribute__((aligned(16))) PqxShared {
    ThrustMail<R> tmail;
    PosMail<R> pmail;
    R qmail[4][DN_BLOCK];
    VerdictMail<R> vmail;
    float zmail[5][DN_BLOCK];             // NOISE: the observation-noise draws of columns 8..12, drawn by P (see pqx_step)
    float zreset[15][DN_BLOCK];           // NOISE: draw_obs_noise_across's scratch (Q)
    R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
};
// One control step of tile `tile` on three waves; `role` 0 = P, 1 = Q, 2 = X, `tid2` = this thread's index among the 128 threads of
// the tile's P and Q waves (they stage the table).  EXT: the action's mean row (and log_std row) is handed over in `ext_m` / `ext_l`
// instead of being read from io.mean.  Every barrier is a WORKGROUP barrier: all waves of the launch that are still alive must run
// this function together (two tiles side by side in one workgroup keep step, which is harmless).
template <typename R, bool NORM, bool NOISE, bool SAMPLE, bool EXT = false>
DN_DEV void pqx_step(const DnParams &p, const DnStepIO &io0, PqxShared<R> &sh, const long long tile, const int role, const unsigned lane,
                     const unsigned tid2, const float4 ext_m = make_float4(0.0f, 0.0f, 0.0f, 0.0f),
                     const float4 ext_l = make_float4(0.0f, 0.0f, 0.0f, 0.0f))
{
    R *const s_tab = sh.s_tab;
    ThrustMail<R> &tmail = sh.tmail;
    PosMail<R> &pmail = sh.pmail;
    R (&qmail)[4][DN_BLOCK] = sh.qmail;
    VerdictMail<R> &vmail = sh.vmail;
    const long long tile_base = tile * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const long long i = tile_base + li;
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    const BlockState b = block_state(p.st, tile_base);
    const DnConsts<R> &c = consts<R>(p);
    const StepOut out = block_out(io0, tile_base, 0, 0);
    PQX_MARK(0);
    if (role == 2) {
        // ---- X: the action chain first (nothing else can start without it), the value side of the step last
        const float4 *act = reinterpret_cast<const float4 *>(io0.actions) + tile_base;
        float4 A = SAMPLE ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : act[li];
        const DnStatSlot slot0 = p.st.stats[tile];                   // the whole slot now (see flush_stats_preloaded)
        const unsigned long long sc0 = slot0.step_count;
        if (SAMPLE) A = EXT ? sample_action_from(io0, ext_m, ext_l, gid, sc0, i, active) : sample_action(io0, gid, sc0, i, active);
        post_thrust<R>(tmail, lane, thrust_phase<NOISE>(p, gid, sc0, A));
        // the entry state this wave reads is not needed before B2: requested only now, so that the action is the first
        // word back from memory and nothing queues ahead of it
        const float4 G0 = b.g0[li], G2 = b.g2[li], G3 = b.g3[li];
        float4 G4 = b.g4[li], G5 = b.g5[li];
        PQX_MARK(1); block_lds_barrier(); PQX_MARK(2);                    // B1
        Flight<R> fl;
        flight_entry<R>(fl, G0, G2, G3, p.max_steps);
        RewardPre<R> pre = reward_entry<R>(p, c, fl, G4, G5);              // while P and Q integrate
        pin(pre.r0); pin(pre.s_lin); pin(pre.s_ang); pin(pre.pen_lin); pin(pre.pen_ang);
        PQX_MARK(3); block_lds_barrier(); PQX_MARK(4);                    // B2
        fl.px = pmail.p[0][lane]; fl.py = pmail.p[1][lane]; fl.pz = pmail.p[2][lane];
        fl.qx = qmail[0][lane]; fl.qy = qmail[1][lane]; fl.qz = qmail[2][lane]; fl.qw = qmail[3][lane];
        attitude_phase<R, false>(fl);                                      // the forward vector of the new pose
        R r_normal;
        float r_found32;
        reward_pose<R>(p, s_tab, fl, pre, r_normal, r_found32);
        pin(r_normal); pin(r_found32);
        PQX_MARK(5); block_lds_barrier(); PQX_MARK(6);                    // B3
        Verdict<R> v;
        v.d_obs = vmail.d_obs[lane];
        const int vf = vmail.flags[lane];
        v.coll1 = vf & 1; v.terminated = (vf >> 1) & 1;
        StatAcc acc;
        RewNorm rn = {0.0, 0.0, 1.0, 1e-4};
        report_scalars<R, false>(p, c, out, fl, v, r_normal, r_found32, li, lane, active, G4, G5, acc, rn);
        flush_stats_preloaded(p, slot0, acc, sc0 + 1ull, lane, tile);
        if (active) { b.g4[li] = G4; b.g5[li] = G5; }
        PQX_MARK(7);
    } else if (role == 0) {
        // ---- P: the linear half of the rigid-body step, then the rules on the new position
        const bool obs_noise = NOISE && p.obs_noise_sigma > 0.0f;
        const unsigned long long sc0 = NOISE ? p.st.stats[tile].step_count : 0ull;     // first: the draws below wait for nothing else
        const float4 G0 = b.g0[li], G1 = b.g1[li], G2 = b.g2[li], G3 = b.g3[li];
        stage_table_by<R>(p, s_tab, tid2, 2 * DN_BLOCK);            // P and Q (threads 0..127) stage the table; X is busy with the thrust
        if (obs_noise) {
            // The 13 Gaussian draws of the step observation's noise depend on (seed, drone, vector step) only.  Drawn by Q between B2
            // and B3 -- where the observation exists -- their seven Box-Muller pairs were ~1.7 us of the tile's critical path; here
            // P (columns 8..12) and Q (columns 0..7) draw them while X draws the action's and computes the thrust.
            float z[4], z4[4];
            obs_noise4(p, gid, sc0, 3u, z);
            obs_noise4(p, gid, sc0, 4u, z4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sh.zmail[j][lane] = z[j];
            sh.zmail[4][lane] = z4[0];
        }
        // everything of the linear step that does not read the thrust, while X still computes it (this wave's loads are back
        // ~400-700 cycles before B1 releases: profiles/r03_pqx_stamps.txt)
        AttCol<R> col = attitude_column<R>(G1);
        LinPre<R> lpre = physics_linear_pre<R>(G2);
        pin(col.r02); pin(col.r12); pin(col.r22); pin(lpre.kl); pin(lpre.vkx); pin(lpre.vky);
        PQX_MARK(1); block_lds_barrier(); PQX_MARK(2);                    // B1
        const GateRow<R> row_e = load_gate_row<R>(s_tab, unpack_meta(G3.w).idx);
        const R wp0[3] = {s_tab[DN_T_WP], s_tab[DN_T_WP + 1], s_tab[DN_T_WP + 2]};
        const R fz = tmail.v[0][lane];
        const Lin<R> lin = physics_linear_post<R>(G0, G2, col, lpre, fz, R(0.0), R(0.0), R(0.0), false);
        Flight<R> fl;
        flight_entry<R>(fl, G0, G2, G3, p.max_steps);
        fl.px = lin.px; fl.py = lin.py; fl.pz = lin.pz;
        fl.vx = (float)lin.vx; fl.vy = (float)lin.vy; fl.vz = (float)lin.vz;
        fl.qx = fl.qy = fl.qz = R(0.0); fl.qw = R(1.0);                    // the attitude belongs to Q (no ground-contact term here)
        fl.wx = fl.wy = fl.wz = 0.0f;
        pmail.p[0][lane] = fl.px; pmail.p[1][lane] = fl.py; pmail.p[2][lane] = fl.pz;
        pmail.v[lane] = make_float4(fl.vx, fl.vy, fl.vz, 0.0f);
        PQX_MARK(3); block_lds_barrier(); PQX_MARK(4);                    // B2
        RulesMid<R> m;
        const Verdict<R> v = rules_verdict<R>(p, c, s_tab, row_e, fl, G3, m);
        vmail.d_obs[lane] = v.d_obs;
        vmail.flags[lane] = v.coll1 | (v.terminated << 1);
        PQX_MARK(5); block_lds_barrier(); PQX_MARK(6);                    // B3
        float4 S0, S1, S2, S3;
        rules_commit<R>(c, wp0, fl, m, G0, G3, b.g6, li, active, S0, S1, S2, S3);
        if (active) {
            b.g0[li] = S0; b.g2[li] = S2;
            reinterpret_cast<float *>(b.g3 + li)[3] = S3.w;
        }
        PQX_MARK(7);
    } else {
        // ---- Q: the angular half, then everything that reads the attitude
        const bool obs_noise = NOISE && p.obs_noise_sigma > 0.0f;
        const unsigned long long sc0 = NOISE ? p.st.stats[tile].step_count : 0ull;
        const float4 G0 = b.g0[li], G1 = b.g1[li], G3 = b.g3[li];
        stage_table_by<R>(p, s_tab, tid2, 2 * DN_BLOCK);            // before the statistics: memory returns in order, and B1 waits for the table
        Rms rms;
        if (NORM) load_rms(p, i, rms);
        float zn[DN_OBS_DIM];
        if (obs_noise) { obs_noise4(p, gid, sc0, 1u, zn); obs_noise4(p, gid, sc0, 2u, zn + 4); }    // columns 0..7 (see P)
        // the part of the angular step that reads only the entry state (rotation matrix, body rates, gyroscopic and damping terms:
        // ~75 of its ~170 instructions), while X still computes the thrust
        AngPre<R> apre = physics_angular_pre<R>(G1, G3);
        pin(apre.r00); pin(apre.r01); pin(apre.r02); pin(apre.r10); pin(apre.r11); pin(apre.r12); pin(apre.r20); pin(apre.r21); pin(apre.r22);
        pin(apre.ka); pin(apre.Iwx); pin(apre.Iwy); pin(apre.Iwz); pin(apre.gx); pin(apre.gy); pin(apre.gz);
        PQX_MARK(1); block_lds_barrier(); PQX_MARK(2);                    // B1
        const R tx = tmail.v[1][lane], ty = tmail.v[2][lane], zt = tmail.v[3][lane];
        const Ang<R> ang = physics_angular_post<R>(apre, tx, ty, zt);
        qmail[0][lane] = ang.qx; qmail[1][lane] = ang.qy; qmail[2][lane] = ang.qz; qmail[3][lane] = ang.qw;
        PQX_MARK(3); block_lds_barrier(); PQX_MARK(4);                    // B2
        Flight<R> fl;
        flight_entry<R>(fl, G0, make_float4(0.0f, 0.0f, 0.0f, 0.0f), G3, p.max_steps);     // d_e, truncated: all this wave reads of it
        fl.px = pmail.p[0][lane]; fl.py = pmail.p[1][lane]; fl.pz = pmail.p[2][lane];
        const float4 nv = pmail.v[lane];
        fl.vx = nv.x; fl.vy = nv.y; fl.vz = nv.z;
        fl.qx = ang.qx; fl.qy = ang.qy; fl.qz = ang.qz; fl.qw = ang.qw;
        fl.wx = (float)ang.wx; fl.wy = (float)ang.wy; fl.wz = (float)ang.wz;
        attitude_phase<R>(fl);
        float o[DN_OBS_DIM];
        observe_columns<R>(p, c, fl, o);
        if (obs_noise) {
#pragma unroll
            for (int j = 0; j < 5; ++j) zn[8 + j] = sh.zmail[j][lane];
            add_obs_noise_drawn(p, zn, o);
        }
        if (NORM) normalize_obs(rms, o);
        PQX_MARK(5); block_lds_barrier(); PQX_MARK(6);                    // B3
        Verdict<R> v;
        v.d_obs = vmail.d_obs[lane];
        const int vf = vmail.flags[lane];
        v.coll1 = vf & 1; v.terminated = (vf >> 1) & 1;
        report_obs<R, NORM, NOISE, 2>(p, c, NOISE ? &sh.zreset[0][0] : nullptr, out, fl.truncated != 0, v, o, gid, sc0, li, lane, rows, active, rms);
        if (NORM && active) store_rms(p, i, rms);
        if (active) {
            const bool done = v.terminated != 0 || fl.truncated != 0;      // the body is reloaded at rest, level (rules_commit's S1 / S3)
            b.g1[li] = done ? make_float4(0.0f, 0.0f, 0.0f, 1.0f) : make_float4((float)ang.qx, (float)ang.qy, (float)ang.qz, (float)ang.qw);
            float *g3 = reinterpret_cast<float *>(b.g3 + li);
            g3[0] = done ? 0.0f : fl.wx; g3[1] = done ? 0.0f : fl.wy; g3[2] = done ? 0.0f : fl.wz;
        }
        PQX_MARK(7);
    }
}

template <typename R, bool NORM, bool NOISE, bool SAMPLE>
__global__ __launch_bounds__(3 * DN_BLOCK) void dn_step_pqx_kernel(const DnParams p, const DnStepIO io0)
{
    __shared__ PqxShared<R> sh;
    pqx_step<R, NORM, NOISE, SAMPLE>(p, io0, sh, (long long)blockIdx.x, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6),
                                     threadIdx.x & (DN_BLOCK - 1), threadIdx.x);
}

// =====================================================================================================
// VecEnv.reset(): every drone goes through Monitor.reset / NormalizeObservation.reset / PBDroneEnv.reset.
// =====================================================================================================
template <typename R>
__global__ __launch_bounds__(DN_BLOCK) void dn_reset_kernel(const DnParams p, float *obs)
{
    __shared__ R s_tab[DN_MAX_WAYPOINTS * DN_T_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_tile[DN_BLOCK * DN_OBS_DIM];
    const unsigned lane = threadIdx.x;
    const long long tile_base = (long long)blockIdx.x * DN_BLOCK;
    const long long left = p.n - tile_base;
    const unsigned rows = left < DN_BLOCK ? (unsigned)left : DN_BLOCK;
    const bool active = lane < rows;
    const unsigned li = active ? lane : rows - 1;
    __builtin_assume(li < DN_BLOCK);
    const long long i = tile_base + li;
    const DnConsts<R> &c = consts<R>(p);
    const BlockState b = block_state(p.st, tile_base);
    const float4 G0 = b.g0[li], G3 = b.g3[li], G6 = b.g6[li];
    stage_table<R>(p, s_tab);
    block_lds_barrier();
    const Meta m = unpack_meta(G3.w);
    R cpx, cpy, cpz;
    if (m.steps > 0) { cpx = G0.x; cpy = G0.y; cpz = G0.z; } else { cpx = G6.x; cpy = G6.y; cpz = G6.z; }
    float o[DN_OBS_DIM];
    reset_obs<R>(p, c, (R)G0.w, o);
    const unsigned long long gid = (unsigned long long)(p.env_id_offset + i);
    float sx = (float)c.spawn[0], sy = (float)c.spawn[1], sz = (float)c.spawn[2];
    if (p.random_spawn) {                  // this episode's INIT_XYZS[0]; _current_position follows it (PBDroneEnv.py:624-626)
        double q[3];
        spawn_point(p, gid, p.st.stats[blockIdx.x].step_count, q);
        sx = (float)q[0]; sy = (float)q[1]; sz = (float)q[2];
        cpx = (R)sx; cpy = (R)sy; cpz = (R)sz;
        o[0] = (float)((R)sx * c.inv_dim[0]); o[1] = (float)((R)sy * c.inv_dim[1]); o[2] = (float)((R)sz * c.inv_dim[2]);
    }
    if (p.obs_noise_sigma > 0.0f) add_obs_noise(p, gid, p.st.stats[blockIdx.x].step_count, 5u, o);
    if (p.normalize_obs) {
        Rms rms;
        load_rms(p, i, rms);
        normalize_obs(rms, o);
        if (active) store_rms(p, i, rms);
    }
    const R ex = cpx - s_tab[0], ey = cpy - s_tab[1], ez = cpz - s_tab[2];
    const R d = FM<R>::sqrt0(FM<R>::fma(ez, ez, FM<R>::fma(ey, ey, ex * ex)));
    if (active) {
        b.g0[li] = make_float4(sx, sy, sz, (float)d);
        b.g1[li] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        b.g2[li] = make_float4(0.0f, 0.0f, 0.0f, (float)d);
        b.g3[li] = make_float4(0.0f, 0.0f, 0.0f, pack_meta(0, 0, 0));
        b.g4[li] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        b.g5[li] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0));
        b.g6[li] = make_float4((float)cpx, (float)cpy, (float)cpz, 0.0f);
        if (p.drag) b.g7[li] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);      // last_clipped_action, BaseAviary.py:545
    }
    store_obs_tile(s_tile, obs + tile_base * DN_OBS_DIM, rows, lane, o);
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
// One workgroup of 1024 lanes per 1024 words (65 536 drones), one word per lane.  A workgroup's place in the list is the
// number of set bits in all earlier words, which it counts itself (coalesced popcount sweep + reduction: at 2 M drones the
// last of 32 workgroups re-reads 256 KB out of L2) -- no second launch, no scratch buffer, no atomics, and the list stays in
// ascending order whatever order the workgroups run in.
// =====================================================================================================
__global__ __launch_bounds__(1024) void dn_compact_kernel(const unsigned long long *__restrict__ mask, long long n,
                                                          int32_t *__restrict__ indices, int32_t *__restrict__ count)
{
    __shared__ int s_wave[16], s_before[16];
    const long long words = (n + 63) / 64;
    const long long w_base = (long long)blockIdx.x * 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int before = 0;
    for (long long w = threadIdx.x; w < w_base; w += 1024) before += __popcll(mask[w]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    const long long w = w_base + threadIdx.x;
    unsigned long long mword = w < words ? mask[w] : 0ull;
    const int mine = __popcll(mword);
    // exclusive scan: within the wave by shuffles, across the 16 waves through LDS
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    if (lane == 0) s_before[wave] = before;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < 16; ++k) base += s_before[k];
    for (int k = 0; k < wave; ++k) base += s_wave[k];
    int pos = base + incl - mine;
    while (mword) {
        int b = __ffsll((long long)mword) - 1;
        indices[pos++] = (int32_t)(w * 64 + b);
        mword &= mword - 1;
    }
    if (threadIdx.x == 1023 && blockIdx.x == gridDim.x - 1) *count = base + incl;
}

// dn_pack_done: the compaction above with, next to every index, the episode-end record of that drone as ONE 64-byte row --
// terminal_observation (13), Monitor return, Monitor length (int bits), TimeLimit.truncated | found_targets << 8 (int bits) -- so that
// the NumPy step() brings the few finished drones' records to the host in one short copy instead of four whole per-drone arrays.
__global__ __launch_bounds__(1024) void dn_compact_pack_kernel(const unsigned long long *__restrict__ mask, long long n,
                                                               const float *__restrict__ terminal_obs, const float *__restrict__ ep_return,
                                                               const int32_t *__restrict__ ep_length, const uint8_t *__restrict__ truncated,
                                                               const int32_t *__restrict__ found, int32_t *__restrict__ indices,
                                                               int32_t *__restrict__ count, float *__restrict__ packed)
{
    __shared__ int s_wave[16], s_before[16];
    const long long words = (n + 63) / 64;
    const long long w_base = (long long)blockIdx.x * 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int before = 0;
    for (long long w = threadIdx.x; w < w_base; w += 1024) before += __popcll(mask[w]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    const long long w = w_base + threadIdx.x;
    unsigned long long mword = w < words ? mask[w] : 0ull;
    const int mine = __popcll(mword);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    if (lane == 0) s_before[wave] = before;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < 16; ++k) base += s_before[k];
    for (int k = 0; k < wave; ++k) base += s_wave[k];
    int pos = base + incl - mine;
    while (mword) {
        const int b = __ffsll((long long)mword) - 1;
        const long long i = w * 64 + b;
        indices[pos] = (int32_t)i;
        float *row = packed + (long long)pos * 16;
#pragma unroll
        for (int k = 0; k < DN_OBS_DIM; ++k) row[k] = terminal_obs[i * DN_OBS_DIM + k];
        row[13] = ep_return[i];
        row[14] = __int_as_float(ep_length[i]);
        row[15] = __int_as_float((int)truncated[i] | (found[i] << 8));
        ++pos;
        mword &= mword - 1;
    }
    if (threadIdx.x == 1023 && blockIdx.x == gridDim.x - 1) *count = base + incl;
}

// A1-A3 on their own (dn_preprocess_action): N x PBDroneEnv._preprocessAction + the force/torque lines of
// BaseAviary._physics, so the float32 chain can be checked bit for bit against the reference's golden vectors.
__global__ __launch_bounds__(256) void dn_action_chain_kernel(const float4 *__restrict__ actions, long long n, int normalize_actions,
                                                              float4 *__restrict__ rpm, float4 *__restrict__ forces,
                                                              float *__restrict__ z_torque)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 A = actions[i];
    const float a[4] = {A.x, A.y, A.z, A.w};
    float tq[4], f[4], r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float tq_chain;
        rotor_force_from_action(a[j], normalize_actions != 0, tq_chain, &r[j]);          // the rpm output: the chain itself
        f[j] = rotor_force_sat(a[j], normalize_actions != 0, tq[j]);                     // forces / torque: the path the step kernels take
    }
    if (rpm) rpm[i] = make_float4(r[0], r[1], r[2], r[3]);
    if (forces) forces[i] = make_float4(f[0], f[1], f[2], f[3]);
    if (z_torque) z_torque[i] = z_torque32(tq);
}

// Rollout-step glue around the policy network (the host loop's small element-wise kernels, fused):
//   dn_policy_sample_kernel  -- SB3 DiagGaussianDistribution.sample / log_prob and the np.clip of collect_rollouts
//                               [3P-recall]: actions = mean + exp(log_std) z, z ~ N(0,1) from the environment's Philox
//                               stream (seed, global drone id, the tile's vector-step counter, stream 9), the clipped copy
//                               that goes to dn_step, and log N(actions; mean, std) summed over the four action dims.
//   dn_add_bootstrap_kernel  -- reward += gamma * V(terminal_observation) where TimeLimit.truncated (SB3's
//                               collect_rollouts bootstrap) [3P-recall].
__global__ __launch_bounds__(256) void dn_policy_sample_kernel(const DnParams p, const float4 *__restrict__ mean,
                                                               const float4 log_std, const unsigned long long seed,
                                                               const int deterministic, float4 *__restrict__ actions,
                                                               float4 *__restrict__ clipped, float *__restrict__ log_prob)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const float4 m = mean[i];
    const float mu[4] = {m.x, m.y, m.z, m.w};
    const float ls[4] = {log_std.x, log_std.y, log_std.z, log_std.w};
    float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!deterministic)
        noise4(seed, (unsigned long long)(p.env_id_offset + i), p.st.stats[i / DN_BLOCK].step_count, 9u, z);
    float a[4], lp = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = mu[j] + expf(ls[j]) * z[j];
        // log N(a; mu, sigma) with (a - mu)/sigma = z
        lp += -0.5f * z[j] * z[j] - ls[j] - 0.91893853320467274178f;
    }
    actions[i] = make_float4(a[0], a[1], a[2], a[3]);
    clipped[i] = make_float4(clipv(a[0], -1.0f, 1.0f), clipv(a[1], -1.0f, 1.0f), clipv(a[2], -1.0f, 1.0f), clipv(a[3], -1.0f, 1.0f));
    log_prob[i] = lp;
}

// dn_squashed_sample_kernel -- SB3 SAC's Actor on top of the (mu | log_std) rows of dn_mlp_forward(arch = SAC) [3P-recall of
// sac/policies.py and SquashedDiagGaussianDistribution]: log_std clamped to [-20, 2], pre = mu + exp(log_std) z with z ~ N(0,1)
// from the environment's Philox stream (seed, global drone id, the tile's vector-step counter, stream 9: as dn_policy_sample),
// action = tanh(pre) -- already inside dn_step's [-1, 1] --, log_prob = sum_j log N(pre_j; mu_j, sigma_j) - log(1 - a_j^2 + 1e-6).
__global__ __launch_bounds__(256) void dn_squashed_sample_kernel(const DnParams p, const float4 *__restrict__ mu_log_std,
                                                                 const unsigned long long seed, const int deterministic,
                                                                 float4 *__restrict__ actions, float *__restrict__ log_prob)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!deterministic)
        noise4(seed, (unsigned long long)(p.env_id_offset + i), p.st.stats[i / DN_BLOCK].step_count, 9u, z);
    float a[4], lp;
    squashed_draw(mu_log_std[2 * i], mu_log_std[2 * i + 1], z, log_prob != nullptr, a, lp);
    actions[i] = make_float4(a[0], a[1], a[2], a[3]);
    if (log_prob) log_prob[i] = lp;
}

__global__ __launch_bounds__(256) void dn_add_bootstrap_kernel(float *__restrict__ reward, const float *__restrict__ terminal_value,
                                                               const uint8_t *__restrict__ truncated, float gamma, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && truncated[i]) reward[i] = reward[i] + gamma * terminal_value[i];
}

__global__ __launch_bounds__(256) void dn_set_step_count_kernel(DnStatSlot *slots, long long blocks, unsigned long long value)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < blocks; i += (long long)gridDim.x * blockDim.x)
        slots[i].step_count = value;
}

__global__ __launch_bounds__(256) void dn_fill4_kernel(float4 *dst, float4 v, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}
// The copy ceiling of the box (dn_stream_copy): ONE 16-byte load and one 16-byte store per lane, no loop -- the form that measured
// fastest on MI355X (profiles/r04_copy_sweep.txt: 6 237 GB/s read + write over 1 GiB, against 4 100-5 200 for grid-stride loops of
// 4-64 workgroups per CU with or without non-temporal hints, 5 300-5 750 for a contiguous chunk per workgroup, 4 689 for hipMemcpyAsync
// and ~5 500 for torch's copy_): the dispatcher streams workgroups onto CUs as they drain, and every wave has exactly one load in flight.
typedef float dn_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void dn_stream_copy_kernel(const float4 *__restrict__ src4, float4 *__restrict__ dst4, long long n16)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) reinterpret_cast<dn_v4f *>(dst4)[i] = reinterpret_cast<const dn_v4f *>(src4)[i];
}
__global__ __launch_bounds__(256) void dn_filld_kernel(double *dst, double v, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}

}  // namespace

#if DN_TU == 1
int dn_norm_exact_compiled_in() { return DN_NORM_EXACT; }

hipError_t dn_launch_fill4(float4 *dst, float4 v, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(dn_fill4_kernel, dim3(grid), dim3(256), 0, stream, dst, v, n);
    return hipGetLastError();
}

hipError_t dn_launch_stream_copy(void *dst, const void *src, long long n16, int num_cus, hipStream_t stream)
{
    (void)num_cus;
    const long long per = 256ll * 0x40000000ll;                  // grid.x <= 2^30 workgroups per launch
    for (long long off = 0; off < n16; off += per) {
        const long long m = n16 - off < per ? n16 - off : per;
        hipLaunchKernelGGL(dn_stream_copy_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, (const float4 *)src + off, (float4 *)dst + off, m);
    }
    return hipGetLastError();
}

hipError_t dn_launch_filld(double *dst, double v, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(dn_filld_kernel, dim3(grid), dim3(256), 0, stream, dst, v, n);
    return hipGetLastError();
}

#endif

// Kernel shape: two or three waves per 64 drones where the chip would otherwise idle (the step is bound by one wave's
// dependent instruction stream and the other waves run on other SIMDs), one wave per 64 drones where there are enough
// drones to fill every SIMD with whole steps.  All produce identical bits (test_kernel_shapes_are_bit_identical).
//
// This file is compiled twice (build.py).  DN_TU == 1 (with -mllvm -disable-machine-licm): everything except the
// multi-wave kernels of the configuration without the normaliser; those live in DN_TU == 2 (dn_kernels_mw.hip, machine
// LICM on).  Hoisting the float64 literals of the step body out of the K-step loop costs registers: it slows the
// one-wave kernels that run several waves per SIMD (2 M drones fused: 72 -> 107 us per step) and speeds up the
// kernels that are alone or nearly alone on their SIMD (32768 drones, three waves: 1.48 -> 1.37 us per step).
#define DN_LAUNCH3(R, NORM, NOISE, ONE, XOPT)                                                                           \
    do {                                                                                                                \
        if (two_wave)                                                                                                   \
            DN_KLAUNCH((dn_step_many_2w_kernel<R, NORM, NOISE, ONE, XOPT>), dim3(grid), dim3(2 * DN_BLOCK), 0, stream, p, io, k); \
        else                                                                                                            \
            DN_KLAUNCH((dn_step_many_1w_kernel<R, NORM, NOISE, ONE, XOPT>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, k);     \
    } while (0)
#define DN_LAUNCH2(R, NORM, NOISE, ONE)                                                                                 \
    do {                                                                                                                \
        if (rew) DN_LAUNCH3(R, NORM, NOISE, ONE, true); else DN_LAUNCH3(R, NORM, NOISE, ONE, false);                    \
    } while (0)
#define DN_LAUNCH(R, NORM, NOISE)                                                                                       \
    do {                                                                                                                \
        if (k == 1) DN_LAUNCH2(R, NORM, NOISE, true); else DN_LAUNCH2(R, NORM, NOISE, false);                           \
    } while (0)

#if DN_TU == 2
hipError_t dn_launch_step_many_mw(const DnParams &p, const DnStepIO &io, int k, bool f32, int waves, hipStream_t stream)
{   // multi-wave kernels, normaliser off
    constexpr bool two_wave = true;
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
    // the rarely used options share one set of instantiations (runtime switches inside): reward wrappers, N4 physics terms
    const bool rew = p.clip_rew != 0 || p.norm_rew != 0 || p.gnd != 0 || p.drag != 0 || p.rpm_actions != 0 || p.pid_mode != 0 || p.random_spawn != 0 || p.zero_damping != 0;
    if (waves == 8 && k > 1 && !noise && !rew && p.normalize_obs) {   // role-pipelined kernel (round 4): eight roles, normaliser on
        if (f32) DN_KLAUNCH((dn_step_many_rp8_kernel<float>), dim3(grid), dim3(8 * DN_BLOCK), 0, stream, p, io, k);
        else DN_KLAUNCH((dn_step_many_rp8_kernel<double>), dim3(grid), dim3(8 * DN_BLOCK), 0, stream, p, io, k);
        return hipGetLastError();
    }
    if (waves == 5 && k > 1) {                             // four waves + the normaliser's: fused launches of the plain configuration, normaliser on
        const dim3 blk(5 * DN_BLOCK);
        if (f32) {
            if (noise) DN_KLAUNCH((dn_step_many_5w_kernel<float, true>), dim3(grid), blk, 0, stream, p, io, k);
            else DN_KLAUNCH((dn_step_many_5w_kernel<float, false>), dim3(grid), blk, 0, stream, p, io, k);
        } else {
            if (noise) DN_KLAUNCH((dn_step_many_5w_kernel<double, true>), dim3(grid), blk, 0, stream, p, io, k);
            else DN_KLAUNCH((dn_step_many_5w_kernel<double, false>), dim3(grid), blk, 0, stream, p, io, k);
        }
        return hipGetLastError();
    }
    if (waves == 4 && k > 1) {                             // four waves per tile: fused launches of the plain configuration (the caller checked)
        const dim3 blk(4 * DN_BLOCK);
        const bool norm = p.normalize_obs != 0;
#define DN_L4(R, NORM)                                                                                                      \
        do {                                                                                                                \
            if (noise) DN_KLAUNCH((dn_step_many_4w_kernel<R, NORM, true>), dim3(grid), blk, 0, stream, p, io, k);    \
            else DN_KLAUNCH((dn_step_many_4w_kernel<R, NORM, false>), dim3(grid), blk, 0, stream, p, io, k);         \
        } while (0)
        if (f32) { if (norm) DN_L4(float, true); else DN_L4(float, false); }
        else { if (norm) DN_L4(double, true); else DN_L4(double, false); }
#undef DN_L4
        return hipGetLastError();
    }
    if (waves == 3 && k > 1) {                             // three waves per tile: fused launches
        const dim3 blk(3 * DN_BLOCK);
        const bool norm = p.normalize_obs != 0;
#define DN_L3X(R, NORM, NOISE)                                                                                              \
        do {                                                                                                                \
            if (rew) DN_KLAUNCH((dn_step_many_3w_kernel<R, NORM, NOISE, true>), dim3(grid), blk, 0, stream, p, io, k);   \
            else DN_KLAUNCH((dn_step_many_3w_kernel<R, NORM, NOISE, false>), dim3(grid), blk, 0, stream, p, io, k);      \
        } while (0)
#define DN_L3(R, NORM)                                                                                                      \
        do {                                                                                                                \
            if (noise) DN_L3X(R, NORM, true); else DN_L3X(R, NORM, false);                                                  \
        } while (0)
        if (f32) { if (norm) DN_L3(float, true); else DN_L3(float, false); }
        else { if (norm) DN_L3(double, true); else DN_L3(double, false); }
#undef DN_L3
#undef DN_L3X
        return hipGetLastError();
    }
    if (f32) { if (noise) DN_LAUNCH(float, false, true); else DN_LAUNCH(float, false, false); }
    else { if (noise) DN_LAUNCH(double, false, true); else DN_LAUNCH(double, false, false); }
    return hipGetLastError();
}
#ifdef DN_MW_STAMP
extern "C" int dn_debug_mw_edges(long long *out)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mw_edge), sizeof(long long) * 128) == hipSuccess ? 0 : 1;
}
extern "C" int dn_debug_mw_stamps(long long *out)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mw_stamp), sizeof(long long) * 8 * 48 * 2) == hipSuccess ? 0 : 1;
}
#endif
#elif DN_TU == 1
hipError_t dn_launch_step_many(const DnParams &p, const DnStepIO &io, int k, bool f32, int waves, hipStream_t stream)
{
    const bool norm = p.normalize_obs != 0;
    if (io.mean && waves == 3) {            // dn_step_sampled on three waves
        const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
        const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
#define DN_LPS(R, NORM, NOISE) DN_KLAUNCH((dn_step_pqx_kernel<R, NORM, NOISE, true>), dim3(grid), dim3(3 * DN_BLOCK), 0, stream, p, io)
        if (f32) {
            if (norm) { if (noise) DN_LPS(float, true, true); else DN_LPS(float, true, false); }
            else { if (noise) DN_LPS(float, false, true); else DN_LPS(float, false, false); }
        } else {
            if (norm) { if (noise) DN_LPS(double, true, true); else DN_LPS(double, true, false); }
            else { if (noise) DN_LPS(double, false, true); else DN_LPS(double, false, false); }
        }
#undef DN_LPS
        return hipGetLastError();
    }
    if (io.mean) {                          // dn_step_sampled: one-wave single-step kernels with the sampler compiled in
        const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
        const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
#define DN_LS(R, NORM, NOISE) DN_KLAUNCH((dn_step_many_1w_kernel<R, NORM, NOISE, true, false, true>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, 1)
        if (f32) {
            if (norm) { if (noise) DN_LS(float, true, true); else DN_LS(float, true, false); }
            else { if (noise) DN_LS(float, false, true); else DN_LS(float, false, false); }
        } else {
            if (norm) { if (noise) DN_LS(double, true, true); else DN_LS(double, true, false); }
            else { if (noise) DN_LS(double, false, true); else DN_LS(double, false, false); }
        }
#undef DN_LS
        return hipGetLastError();
    }
    if (waves == 3 && k == 1) {             // dn_step on three waves cut by dependency (plain configuration; the caller checked)
        const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
        const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
#define DN_LP(R, NORM, NOISE) DN_KLAUNCH((dn_step_pqx_kernel<R, NORM, NOISE, false>), dim3(grid), dim3(3 * DN_BLOCK), 0, stream, p, io)
        if (f32) {
            if (norm) { if (noise) DN_LP(float, true, true); else DN_LP(float, true, false); }
            else { if (noise) DN_LP(float, false, true); else DN_LP(float, false, false); }
        } else {
            if (norm) { if (noise) DN_LP(double, true, true); else DN_LP(double, true, false); }
            else { if (noise) DN_LP(double, false, true); else DN_LP(double, false, false); }
        }
#undef DN_LP
        return hipGetLastError();
    }
    if ((waves >= 2 && !norm) || (waves >= 3 && k > 1)) return dn_launch_step_many_mw(p, io, k, f32, waves, stream);
    const bool two_wave = waves >= 2;       // with the normaliser: the two-wave kernels (there is no three-wave one)
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    const bool noise = p.act_noise_sigma > 0.0f || p.obs_noise_sigma > 0.0f;
    const bool rew = p.clip_rew != 0 || p.norm_rew != 0 || p.gnd != 0 || p.drag != 0 || p.rpm_actions != 0 || p.pid_mode != 0 || p.random_spawn != 0 || p.zero_damping != 0;
    if (two_wave) {                         // normaliser on
        if (f32) { if (noise) DN_LAUNCH(float, true, true); else DN_LAUNCH(float, true, false); }
        else { if (noise) DN_LAUNCH(double, true, true); else DN_LAUNCH(double, true, false); }
        return hipGetLastError();
    }
    if (f32) {
        if (norm) { if (noise) DN_LAUNCH(float, true, true); else DN_LAUNCH(float, true, false); }
        else { if (noise) DN_LAUNCH(float, false, true); else DN_LAUNCH(float, false, false); }
    } else {
        if (norm) { if (noise) DN_LAUNCH(double, true, true); else DN_LAUNCH(double, true, false); }
        else { if (noise) DN_LAUNCH(double, false, true); else DN_LAUNCH(double, false, false); }
    }
    return hipGetLastError();
}
#endif
#undef DN_LAUNCH
#undef DN_LAUNCH2
#undef DN_LAUNCH3

#if DN_TU == 1
hipError_t dn_launch_reset(const DnParams &p, float *obs, bool f32, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    if (f32) hipLaunchKernelGGL(dn_reset_kernel<float>, dim3(grid), dim3(DN_BLOCK), 0, stream, p, obs);
    else hipLaunchKernelGGL(dn_reset_kernel<double>, dim3(grid), dim3(DN_BLOCK), 0, stream, p, obs);
    return hipGetLastError();
}

hipError_t dn_launch_eval_kinematics(const DnParams &p, const DnStepIO &io, const double *kin, bool f32, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + DN_BLOCK - 1) / DN_BLOCK);
    const bool norm = p.normalize_obs != 0;
    if (f32) {
        if (norm) hipLaunchKernelGGL((dn_eval_kinematics_kernel<float, true>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, kin);
        else hipLaunchKernelGGL((dn_eval_kinematics_kernel<float, false>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, kin);
    } else {
        if (norm) hipLaunchKernelGGL((dn_eval_kinematics_kernel<double, true>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, kin);
        else hipLaunchKernelGGL((dn_eval_kinematics_kernel<double, false>), dim3(grid), dim3(DN_BLOCK), 0, stream, p, io, kin);
    }
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

hipError_t dn_launch_action_chain(const float *actions, long long n, int normalize_actions, float *rpm, float *forces,
                                  float *z_torque, hipStream_t stream)
{
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dn_action_chain_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4 *>(actions), n,
                       normalize_actions, reinterpret_cast<float4 *>(rpm), reinterpret_cast<float4 *>(forces), z_torque);
    return hipGetLastError();
}

hipError_t dn_launch_squashed_sample(const DnParams &p, const float *mu_log_std, unsigned long long seed, int deterministic,
                                     float *actions, float *log_prob, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + 255) / 256);
    hipLaunchKernelGGL(dn_squashed_sample_kernel, dim3(grid), dim3(256), 0, stream, p, reinterpret_cast<const float4 *>(mu_log_std),
                       seed, deterministic, reinterpret_cast<float4 *>(actions), log_prob);
    return hipGetLastError();
}

hipError_t dn_launch_policy_sample(const DnParams &p, const float *mean, const float *log_std4, unsigned long long seed, int deterministic,
                                   float *actions, float *clipped, float *log_prob, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + 255) / 256);
    hipLaunchKernelGGL(dn_policy_sample_kernel, dim3(grid), dim3(256), 0, stream, p, reinterpret_cast<const float4 *>(mean),
                       make_float4(log_std4[0], log_std4[1], log_std4[2], log_std4[3]), seed, deterministic,
                       reinterpret_cast<float4 *>(actions), reinterpret_cast<float4 *>(clipped), log_prob);
    return hipGetLastError();
}

hipError_t dn_launch_add_bootstrap(float *reward, const float *terminal_value, const uint8_t *truncated, float gamma, long long n,
                                   hipStream_t stream)
{
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dn_add_bootstrap_kernel, dim3(grid), dim3(256), 0, stream, reward, terminal_value, truncated, gamma, n);
    return hipGetLastError();
}

hipError_t dn_launch_set_step_count(DnStatSlot *slots, long long blocks, unsigned long long value, hipStream_t stream)
{
    const unsigned grid = (unsigned)((blocks + 255) / 256 < 1024 ? (blocks + 255) / 256 : 1024);
    hipLaunchKernelGGL(dn_set_step_count_kernel, dim3(grid), dim3(256), 0, stream, slots, blocks, value);
    return hipGetLastError();
}

hipError_t dn_launch_compact(const unsigned long long *mask, long long n, int32_t *indices, int32_t *count,
                             hipStream_t stream)
{
    const unsigned blocks = (unsigned)(((n + 63) / 64 + 1023) / 1024);
    hipLaunchKernelGGL(dn_compact_kernel, dim3(blocks), dim3(1024), 0, stream, mask, n, indices, count);
    return hipGetLastError();
}
hipError_t dn_launch_compact_pack(const unsigned long long *mask, long long n, const float *terminal_obs, const float *ep_return,
                                  const int32_t *ep_length, const uint8_t *truncated, const int32_t *found, int32_t *indices, int32_t *count,
                                  float *packed, hipStream_t stream)
{
    const unsigned blocks = (unsigned)(((n + 63) / 64 + 1023) / 1024);
    hipLaunchKernelGGL(dn_compact_pack_kernel, dim3(blocks), dim3(1024), 0, stream, mask, n, terminal_obs, ep_return, ep_length, truncated, found,
                       indices, count, packed);
    return hipGetLastError();
}
#ifdef DN_PQX_STAMP
extern "C" int dn_debug_pqx_stamps(long long *out)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pqx_stamp), sizeof(long long) * 48) == hipSuccess ? 0 : 1;
}
#endif
#endif  // DN_TU == 1
