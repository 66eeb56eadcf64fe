// dn_capi.cpp -- host side of the C ABI declared in include/dronenav.h.
//
// Owns the device arena, the waypoint/corridor tables and the launch parameters; every entry point
// validates its arguments, turns HIP failures into dn_status codes + a thread-local message, and
// never falls back to a CPU implementation.
#include "dn_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int32_t fail(dn_status st, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return (int32_t)st;
}

#define DN_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? DN_ERR_OUT_OF_MEMORY : DN_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                   \
    } while (0)

inline double norm3d(const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// Shape crossovers, measured on the 256-CU MI355X and kept as TILES PER CU (a 64-drone tile is one workgroup): what decides a
// shape is how many waves meet on a SIMD, so a partitioned (CPX) or smaller device scales the fleet sizes with its own
// hipDeviceProp_t.multiProcessorCount (dn_create) instead of inheriting the 256-CU figures.
constexpr long long DN_CALIBRATION_CUS = 256;
constexpr long long DN_TWO_WAVE_TILES_PER_CU = 4;   // 1024 tiles = 65536 drones on 256 CUs: one tile per SIMD
constexpr long long DN_PQX_TILES_PER_CU = 4;        // three-wave single step: while the tiles alone leave SIMDs idle
constexpr long long DN_FIVE_WAVE_TILES_PER_CU = 2;  // the same with the normaliser on a fifth wave: up to two tiles per CU (round 6: beyond, the four-wave kernel's X wave carries the normaliser -- 149 registers since its statistics are addressed from a walked pair -- and wins: dn_create)
constexpr long long DN_ROLE_PIPE_TILES_PER_CU = 6;  // role-pipelined fused step (eight roles per tile): up to one tile per CU and from three to six, see dn_create
constexpr long long DN_FOUR_WAVE_TILES_PER_CU = 3;  // four-wave fused step: up to three tiles per CU (768 tiles on 256 CUs)
constexpr double DN_CONTACT_MARGIN = 0.02;          // Bullet's contact-breaking threshold (dn_kernels.hip collision_common)
constexpr double DN_COLL_R = 0.06, DN_COLL_H = 0.025;   // base_link collision cylinder, cf2x.urdf:34

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// dn_set_state: the second moment the device carries for a given RunningMeanStd.var / .count -- the double m nearest to var x count
// with m / count == var where one exists (so that dn_get_state of what was just set returns the var it was given, and a blob taken
// with dn_get_state restores to statistics that read back identically).
double second_moment(double var, double count)
{
    const double m = var * count;
    if (!(count > 0.0) || !std::isfinite(m) || m / count == var) return m;
    const double lo = std::nextafter(m, -HUGE_VAL), hi = std::nextafter(m, HUGE_VAL);
    if (lo / count == var) return lo;
    if (hi / count == var) return hi;
    return m;
}

// An on / off environment switch read by dn_create: 1 / 0, or -1 for a value that is neither (the create then fails: a typo
// must not silently select the other arithmetic).  Unset or empty = off.
int env_switch(const char *name)
{
    const char *x = getenv(name);
    if (!x || !x[0]) return 0;
    char b[8] = "";
    size_t k = 0;
    for (; x[k] && k < sizeof b - 1; ++k) b[k] = (char)((x[k] >= 'A' && x[k] <= 'Z') ? x[k] - 'A' + 'a' : x[k]);
    if (x[k]) return -1;
    if (!strcmp(b, "1") || !strcmp(b, "true") || !strcmp(b, "on") || !strcmp(b, "yes")) return 1;
    if (!strcmp(b, "0") || !strcmp(b, "false") || !strcmp(b, "off") || !strcmp(b, "no")) return 0;
    return -1;
}

}  // namespace

struct dn_env {
    dn_config cfg;
    DnParams p;
    void *arena = nullptr;
    size_t arena_bytes = 0;
    double *tab64 = nullptr;
    float *tab32 = nullptr;
    long long blocks = 0;
    int num_cus = (int)DN_CALIBRATION_CUS;   // hipDeviceProp_t.multiProcessorCount of cfg.device_id
    int waves_fused = 2;        // kernel shape of dn_step_many (k > 1), see dn_launch_step_many
    int waves_single = 1;       // kernel shape of dn_step (k == 1)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // dn_set_launch_events: attached to the next step kernel's dispatch, then cleared
};

thread_local hipEvent_t dn_tl_ev_start = nullptr, dn_tl_ev_stop = nullptr;

// Takes the events dn_set_launch_events armed: constructed FIRST THING in every step-family entry point, so that the call they were
// armed for consumes them (the hooked launches) or drops them (a validation failure, an entry point without the hook) -- they never
// survive into a later call, whose caller may have destroyed them by then.  Nests (dn_step_many(k = 1) -> dn_step).
thread_local int dn_tl_ev_depth = 0;
struct LaunchEvents {
    explicit LaunchEvents(dn_env *e)
    {
        if (dn_tl_ev_depth++ == 0) { dn_tl_ev_start = e->ev_start; dn_tl_ev_stop = e->ev_stop; }
        e->ev_start = e->ev_stop = nullptr;
    }
    ~LaunchEvents() { if (--dn_tl_ev_depth == 0) dn_tl_ev_start = dn_tl_ev_stop = nullptr; }
    LaunchEvents(const LaunchEvents &) = delete;
    LaunchEvents &operator=(const LaunchEvents &) = delete;
};
// an armed launch inside a stream capture would record the events into the graph (undefined on replay): refused
static bool armed_while_capturing(hipStream_t stream)
{
    if (!dn_tl_ev_start && !dn_tl_ev_stop) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
#define DN_REFUSE_ARMED_CAPTURE(stream)                                                                                           \
    do {                                                                                                                          \
        if (armed_while_capturing((hipStream_t)(stream)))                                                                         \
            return fail(DN_ERR_INVALID_ARGUMENT, "dn_set_launch_events: an armed launch cannot be captured into a hipGraph");     \
    } while (0)

namespace {

// Device arena: seven float4 groups, optional normaliser statistics, statistics slots, tables.
struct Layout {
    size_t off_g[7], off_g7, off_mean, off_var, off_count, off_rr, off_pid, off_stats, off_tab64, off_tab32, total;
};

Layout make_layout(long long n, int normalize_obs, int norm_rew = 0, int drag = 0, int pid = 0)
{
    Layout L;
    size_t o = 0;
    for (int k = 0; k < 7; ++k) { L.off_g[k] = o; o = align_up(o + (size_t)n * sizeof(float4), 256); }
    L.off_mean = o;  o = align_up(o + (normalize_obs ? (size_t)n * DN_OBS_DIM * sizeof(double) : 0), 256);
    L.off_var = o;   o = align_up(o + (normalize_obs ? (size_t)n * DN_OBS_DIM * sizeof(double) : 0), 256);
    L.off_count = o; o = align_up(o + (normalize_obs ? (size_t)n * sizeof(double) : 0), 256);
    L.off_rr = o;    o = align_up(o + (norm_rew ? (size_t)n * 4 * sizeof(double) : 0), 256);
    L.off_g7 = o;    o = align_up(o + (drag ? (size_t)n * sizeof(float4) : 0), 256);
    L.off_pid = o;   o = align_up(o + (pid ? (size_t)n * 9 * sizeof(double) : 0), 256);
    L.off_stats = o; o = align_up(o + (size_t)((n + DN_BLOCK - 1) / DN_BLOCK) * sizeof(DnStatSlot), 256);
    L.off_tab64 = o; o = align_up(o + DN_MAX_WAYPOINTS * DN_T_STRIDE * sizeof(double), 256);
    L.off_tab32 = o; o = align_up(o + DN_MAX_WAYPOINTS * DN_T_STRIDE * sizeof(float), 256);
    L.total = o;
    return L;
}

// Corridor table, float64, in the reference's operation order (PBDroneEnv.py:746-786).
void build_table(const dn_config &c, double *tab)
{
    const double ext = 0.2;
    for (int k = 0; k < c.num_waypoints; ++k) {
        double *e = tab + k * DN_T_STRIDE;
        const double *b1 = (k == 0) ? c.spawn : &c.waypoints[3 * (k - 1)];
        const double *b2 = &c.waypoints[3 * k];
        double lv[3] = {b2[0] - b1[0], b2[1] - b1[1], b2[2] - b1[2]};
        double ll = norm3d(lv);
        for (int j = 0; j < 3; ++j) { e[DN_T_WP + j] = b2[j]; e[DN_T_B1 + j] = b1[j]; }
        e[DN_T_LL] = ll;
        if (ll == 0.0) {
            for (int j = 0; j < 3; ++j) { e[DN_T_U + j] = 0.0; e[DN_T_E1 + j] = b1[j]; }
            e[DN_T_LEXT] = 0.0;
            continue;
        }
        double u[3] = {lv[0] / ll, lv[1] / ll, lv[2] / ll};
        double e1[3] = {b1[0] - ext * u[0], b1[1] - ext * u[1], b1[2] - ext * u[2]};
        double e2[3] = {b2[0] + ext * u[0], b2[1] + ext * u[1], b2[2] + ext * u[2]};
        double ee[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
        for (int j = 0; j < 3; ++j) { e[DN_T_U + j] = u[j]; e[DN_T_E1 + j] = e1[j]; }
        e[DN_T_LEXT] = norm3d(ee);
    }
}

template <typename R>
void build_consts(const dn_config &c, DnConsts<R> &k)
{
    for (int j = 0; j < 6; ++j) k.dim[j] = (R)c.aviary_dim[j];
    for (int j = 0; j < 3; ++j) k.spawn[j] = (R)c.spawn[j];
    k.threshold = (R)c.threshold;
    k.thr_ext = (R)(c.threshold + 0.2);
    k.thr2 = (R)(c.threshold * c.threshold);
    k.thr_ext2 = (R)((c.threshold + 0.2) * (c.threshold + 0.2));
    double a = std::fabs(c.aviary_dim[0]) + c.aviary_dim[3], b = std::fabs(c.aviary_dim[1]) + c.aviary_dim[4];
    double m = a > b ? a : b;
    k.max_target_dist = (R)(m > c.aviary_dim[5] ? m : c.aviary_dim[5]);           // PBDroneEnv.py:91
    k.inv_max_target_dist = (R)1.0 / k.max_target_dist;
    for (int j = 0; j < 3; ++j) k.inv_dim[j] = (R)1.0 / k.dim[3 + j];
    // BaseAviary.reset -> _computeObs on the freshly loaded body: pos = spawn, quat = (0,0,0,1), at rest.
    // getEulerFromQuaternion(identity) = (atan2(0,1), asin(-0.0), atan2(0,1)) = (0, -0, 0).
    const R pi = (R)3.14159265358979323846;
    k.reset_obs[0] = k.spawn[0] / k.dim[3];
    k.reset_obs[1] = k.spawn[1] / k.dim[4];
    k.reset_obs[2] = k.spawn[2] / k.dim[5];
    k.reset_obs[3] = (R)0.0 / pi;
    k.reset_obs[4] = (R)-0.0 / pi;
    k.reset_obs[5] = (R)0.0 / pi;
    for (int j = 6; j < 12; ++j) k.reset_obs[j] = (R)0.0;
    for (int j = 0; j < 12; ++j) k.reset_obs32[j] = (float)k.reset_obs[j];
}

int32_t validate(const dn_config *c)
{
    if (!c) return fail(DN_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (c->num_envs < 1) return fail(DN_ERR_INVALID_ARGUMENT, "num_envs must be >= 1 (got %lld)", (long long)c->num_envs);
    if (c->num_envs > (1ll << 31) - 64) return fail(DN_ERR_INVALID_ARGUMENT, "num_envs too large (got %lld)", (long long)c->num_envs);
    if (c->num_waypoints < 1 || c->num_waypoints > DN_MAX_WAYPOINTS)
        return fail(DN_ERR_INVALID_ARGUMENT, "num_waypoints must be in 1..%d (got %d)", DN_MAX_WAYPOINTS, c->num_waypoints);
    if (c->max_steps < 0 || c->max_steps > (1 << 24) - 2)
        return fail(DN_ERR_INVALID_ARGUMENT, "max_steps must be in 0..%d (got %d)", (1 << 24) - 2, c->max_steps);
    if (!(c->threshold >= 0.0)) return fail(DN_ERR_INVALID_ARGUMENT, "threshold must be >= 0");
    for (int j = 0; j < 3; ++j)
        if (!(c->aviary_dim[3 + j] != 0.0)) return fail(DN_ERR_INVALID_ARGUMENT, "aviary_dim high bounds must be non-zero");
    if (c->act_noise_sigma < 0.0f || c->obs_noise_sigma < 0.0f) return fail(DN_ERR_INVALID_ARGUMENT, "noise sigma must be >= 0");
    if (c->ground_contact < 0 || c->ground_contact > DN_GROUND_CONTACT_AUTO)
        return fail(DN_ERR_INVALID_ARGUMENT, "ground_contact must be 0 (off), 1 (on) or 2 (DN_GROUND_CONTACT_AUTO; got %d)", c->ground_contact);
    if (c->physics < 0 || c->physics > 4) return fail(DN_ERR_INVALID_ARGUMENT, "physics must be 0..4 (PYB, PYB_GND, PYB_DRAG, PYB_DW, PYB_GND_DRAG_DW; got %d)", c->physics);
    if (c->action_type < 0 || c->action_type > 5)
        return fail(DN_ERR_INVALID_ARGUMENT, "action_type must be 0 THRUST | 1 RPM | 2 PID | 3 VEL | 4 ONE_D_RPM | 5 ONE_D_PID (got %d)", c->action_type);
    for (int j = 0; j < c->num_waypoints * 3; ++j)
        if (!std::isfinite(c->waypoints[j])) return fail(DN_ERR_INVALID_ARGUMENT, "waypoint %d is not finite", j / 3);
    return DN_OK;
}

// DN_GROUND_CONTACT_AUTO: is the ground-contact term of _has_collision_occurred (PBDroneEnv.py:699) reachable at all?
// The approximated contact fires when the lowest point of the collision cylinder is within the contact margin of z = 0, i.e.
// only for z <= margin + max over tilt of (H/2 |c| + R s) = margin + sqrt(H^2/4 + R^2).  Every other term of the same
// predicate is evaluated on the same fresh position (:678-707), so if the corridor test is on and every point that low is
// already outside the corridor of EVERY segment (a drone is only ever tested against the segment of its current target), the
// term can never change `terminated`: the predicate is an OR.  Then it is dropped (and the kernels built without it are used);
// otherwise -- corridor off, or a track whose corridor reaches down to the floor (spawn / gates at z = 0.1 ... 0.5 with the
// 0.3 + 0.2 corridor) -- it stays on, as in the reference, which always has it.
bool ground_contact_reachable(const dn_config &c)
{
    if (!c.cylinder) return true;                                      // no corridor: nothing else keeps a drone off the floor
    // random_spawn: segment 0 then starts at this episode's drawn spawn point (rules_commit / dn_reset_kernel), which may sit up to 0.1
    // below the lowest waypoint -- the bound below, on the fixed spawn, does not cover it.  The option is dormant in the reference
    // (PBDroneEnv.py:622-627); keep the term, as the reference does.
    if (c.random_spawn) return true;
    const double z_contact = DN_CONTACT_MARGIN + std::sqrt(0.25 * DN_COLL_H * DN_COLL_H + DN_COLL_R * DN_COLL_R);
    double z_min;                                                       // lowest z inside any corridor
    if (c.circle) z_min = 1.0 - c.threshold;                            // torus around the unit circle at z = 1 (:723-741)
    else {
        z_min = 1e300;
        for (int k = 0; k < c.num_waypoints; ++k) {
            const double *b1 = (k == 0) ? c.spawn : &c.waypoints[3 * (k - 1)];
            const double *b2 = &c.waypoints[3 * k];
            const double lv[3] = {b2[0] - b1[0], b2[1] - b1[1], b2[2] - b1[2]};
            const double ll = norm3d(lv);
            double lo;
            if (ll == 0.0) lo = b1[2] - c.threshold;                    // degenerate segment: a ball of the bare threshold (:756-757)
            else {
                const double uz = lv[2] / ll;                           // capsule around the segment extended by 0.2 at both ends
                const double e1z = b1[2] - 0.2 * uz, e2z = b2[2] + 0.2 * uz;
                lo = (e1z < e2z ? e1z : e2z) - (c.threshold + 0.2);
            }
            if (lo < z_min) z_min = lo;
        }
    }
    return !(z_min > z_contact + 1e-9);
}

int32_t init_state_rms(dn_env *e, hipStream_t s)
{   // normalize.RunningMeanStd.__init__, normalize.py:14-18
    const long long n = e->cfg.num_envs;
    DN_HIP(dn_launch_filld(e->p.st.rms_mean, 0.0, n * DN_OBS_DIM, s));
    DN_HIP(dn_launch_filld(e->p.st.rms_m2, 1.0 * 1e-4, n * DN_OBS_DIM, s));     // var = 1, held as the second moment var x count
    DN_HIP(dn_launch_filld(e->p.st.rms_count, 1e-4, n, s));
    return DN_OK;
}

int32_t init_state(dn_env *e, hipStream_t s)
{
    const dn_config &c = e->cfg;
    const long long n = c.num_envs;
    // PBDroneEnv.__init__ (PBDroneEnv.py:122-145) followed by make_env's env.reset (PBDroneSimulator.py:173)
    double df[3] = {c.spawn[0] - c.waypoints[0], c.spawn[1] - c.waypoints[1], c.spawn[2] - c.waypoints[2]};
    const float d = (float)norm3d(df);
    const float sx = (float)c.spawn[0], sy = (float)c.spawn[1], sz = (float)c.spawn[2];
    DN_HIP(dn_launch_fill4(e->p.st.g0, make_float4(sx, sy, sz, d), n, s));
    DN_HIP(dn_launch_fill4(e->p.st.g1, make_float4(0.f, 0.f, 0.f, 1.f), n, s));
    DN_HIP(dn_launch_fill4(e->p.st.g2, make_float4(0.f, 0.f, 0.f, d), n, s));
    DN_HIP(dn_launch_fill4(e->p.st.g3, make_float4(0.f, 0.f, 0.f, 0.f), n, s));
    DN_HIP(dn_launch_fill4(e->p.st.g4, make_float4(0.f, 0.f, 0.f, 0.f), n, s));
    DN_HIP(dn_launch_fill4(e->p.st.g5, make_float4(0.f, 0.f, 0.f, 0.f), n, s));
    DN_HIP(dn_launch_fill4(e->p.st.g6, make_float4(sx, sy, sz, 0.f), n, s));
    if (c.normalize_obs) { int32_t rc = init_state_rms(e, s); if (rc != DN_OK) return rc; }
    if (c.norm_rew) {                              // NormalizeReward.__init__, normalize.py:124-128
        DN_HIP(dn_launch_filld(e->p.st.rr, 0.0, 2 * n, s));            // returns, return_rms.mean
        DN_HIP(dn_launch_filld(e->p.st.rr + 2 * n, 1.0, n, s));        // .var
        DN_HIP(dn_launch_filld(e->p.st.rr + 3 * n, 1e-4, n, s));       // .count
    }
    if (e->p.drag) DN_HIP(dn_launch_fill4(e->p.st.g7, make_float4(0.f, 0.f, 0.f, 0.f), n, s));   // BaseAviary.py:545
    if (e->p.pid_mode) DN_HIP(dn_launch_filld(e->p.st.pid, 0.0, 9 * n, s));                       // DSLPIDControl.reset(), DSLPIDControl.py:63-76
    DN_HIP(hipMemsetAsync(e->p.st.stats, 0, (size_t)e->blocks * sizeof(DnStatSlot), s));
    return DN_OK;
}

}  // namespace

extern "C" {

int32_t dn_abi_version(void) { return DN_ABI_VERSION; }

const char *dn_last_error(void) { return g_err; }

int32_t dn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void dn_config_default(dn_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->num_envs = 1;
    cfg->threshold = 0.3;             // PBDroneSimulator.py:116
    cfg->max_steps = 4096;            // parameter_manager.py:25
    cfg->cylinder = 1;                // PBDroneSimulator.py:167
    cfg->include_distance = 1;        // PBDroneSimulator.py:661
    cfg->normalize_actions = 1;       // PBDroneSimulator.py:662
    cfg->ground_contact = DN_GROUND_CONTACT_AUTO;   // on wherever the term can fire at all (the reference always has it, PBDroneEnv.py:699)
    cfg->aviary_dim[0] = cfg->aviary_dim[1] = -1.0;   // make_env default aviary_dim, PBDroneSimulator.py:141
    cfg->aviary_dim[3] = cfg->aviary_dim[4] = cfg->aviary_dim[5] = 1.0;
}

int64_t dn_state_bytes(int64_t num_envs, int32_t normalize_obs)
{
    if (num_envs < 1) return 0;
    return (int64_t)make_layout(num_envs, normalize_obs).total;
}

int32_t dn_create(const dn_config *cfg, dn_env **out)
{
    if (!out) return fail(DN_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int32_t rc = validate(cfg);
    if (rc != DN_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(DN_ERR_NO_DEVICE, "no HIP device is visible: libdronenav has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(DN_ERR_INVALID_ARGUMENT, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    DN_HIP(hipSetDevice(cfg->device_id));

    dn_env *e = new (std::nothrow) dn_env();
    if (!e) return fail(DN_ERR_OUT_OF_MEMORY, "host allocation failed");
    e->cfg = *cfg;
    if (e->cfg.ground_contact == DN_GROUND_CONTACT_AUTO) e->cfg.ground_contact = ground_contact_reachable(*cfg) ? 1 : 0;
    cfg = &e->cfg;                                      // from here on: the resolved configuration (dn_get_config returns it)
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id) == hipSuccess && cus > 0) e->num_cus = cus;
    }
    const long long n = cfg->num_envs;
    e->blocks = (n + DN_BLOCK - 1) / DN_BLOCK;
    const long long DN_TWO_WAVE_MAX_TILES = DN_TWO_WAVE_TILES_PER_CU * e->num_cus;
    const long long DN_PQX_MAX_TILES = DN_PQX_TILES_PER_CU * e->num_cus;
    const long long DN_FOUR_WAVE_MAX_TILES = DN_FOUR_WAVE_TILES_PER_CU * e->num_cus;
    // Measured on MI355X (profiles/r01_r_sweep_shapes.txt): the fused K-step kernel is bound by the dependent
    // instruction stream of a wave, and two or three waves per tile win while the tiles alone leave SIMDs idle
    // (<= 1024 tiles = 65536 drones on 1024 SIMDs); the single-step launch is latency bound (launch + load round
    // trip) and one wave is never slower.  DN_WAVES=1|2|3 forces a shape (A/B measurements, the bit-identity test).
    // The three-wave shape (flight / report / aux) exists for fused launches.  Measured (profiles/r01_r_sweep_shapes.txt,
    // profiles/r01_r_sweep_options.txt, with the report wave yielding to the other two by s_setprio): without the
    // normaliser it wins wherever more than one wave per tile wins, i.e. up to 1024 tiles (65536 drones: 2.3 us per step
    // against 2.5 with two waves and 2.9 with one); with the normaliser (27 more float64 per drone in a wave) up to 512
    // tiles (32768 drones: 1.8 us against 2.1 with two waves), one wave beyond (49152 drones: 3.3 us against 3.5).
    // The option kernels are wider in registers (the three-wave kernels are compiled for three waves per SIMD, two with
    // the normaliser: __launch_bounds__), so their crossovers sit lower (profiles/r01_r_sweep_options.txt, us per step):
    //   XOPT options (reward wrappers, force terms, rpm actions), all on: 32768 drones 2.15 (3w) / 2.97 (2w) / 3.70 (1w);
    //     49152: 2.43 / 2.99 / 3.66; 65536: 4.48 / 3.14 / 3.72  -> three waves up to 768 tiles, two up to 1024;
    //     with the normaliser 32768: 2.43 / 2.80 / 4.50; 65536: 4.74 / 5.42 / 4.56 -> three up to 512 tiles, one beyond;
    //   noise (Philox + Box-Muller per observation column): 16384 drones 3.51 / 4.63 / 5.86; 32768: 4.86 / 4.78 / 6.02;
    //     49152: 5.36 / 4.88 / 6.04 -> three waves up to 256 tiles, two beyond (same with the normaliser, up to 512);
    //     noise + XOPT without the normaliser follows the XOPT row (49152: 5.75 / 6.23 / 6.99).
    const bool plain = !cfg->clip_rew && !cfg->norm_rew && cfg->physics == 0 && cfg->action_type == 0 && !cfg->random_spawn && !cfg->zero_damping;
    const bool noisy = cfg->act_noise_sigma > 0.0f || cfg->obs_noise_sigma > 0.0f;
    const long long max_multi = cfg->normalize_obs ? DN_TWO_WAVE_MAX_TILES / 2 : DN_TWO_WAVE_MAX_TILES;
    long long max_three = max_multi;
    if (!plain && !cfg->normalize_obs) max_three = DN_TWO_WAVE_MAX_TILES * 3 / 4;
    else if (noisy) max_three = DN_TWO_WAVE_MAX_TILES / 4;
    e->waves_fused = e->blocks <= max_multi ? (e->blocks <= max_three ? 3 : 2) : 1;
    // Four waves (dn_step_many_4w_kernel: the recurrence itself on two waves; plain configuration without the ground-contact term,
    // us per step against the three- / two- / one-wave pick of the table above):
    //   normaliser off   8 192 / 16 384 / 24 576 / 32 768 / 49 152 drones   1.01 / 1.02 / 1.23 / 1.23 / 1.66   against 1.28 / 1.29 / 1.29 / 1.28 / 1.71
    //                    65 536: 2.58 against 1.92 -> up to 768 tiles
    //   normaliser on    8 192 / 16 384 / 24 576 / 32 768 / 49 152           1.37 / 1.39 / 1.73 / 1.76 / 3.27   against 1.48 / 1.50 / 1.87 / 1.89 / 3.36 -> up to 768 tiles
    //   with noise       16 384 / 32 768 / 49 152                            2.99 / 4.47 / 4.84   against 3.75 / 4.85 / 4.84 (normaliser on: 3.43 / 3.75 / 7.44 against
    //                    3.62 / 5.73 / 6.98) -> up to 512 tiles
    if (plain && !cfg->ground_contact && e->blocks <= (noisy ? DN_FOUR_WAVE_MAX_TILES * 2 / 3 : DN_FOUR_WAVE_MAX_TILES)) e->waves_fused = 4;
    // Five waves (normaliser on; round 3): the four-wave kernel with the normaliser on a wave of its own (NW = 5), where the report wave set
    // the pace.  DN_WAVES=4 keeps the four-wave shape for A/B runs; see profiles/r03_notes.md.
    if (e->waves_fused == 4 && cfg->normalize_obs && e->blocks <= DN_FIVE_WAVE_TILES_PER_CU * e->num_cus) e->waves_fused = 5;
    // Role-pipelined kernel (round 4, dn_step_many_rp8_kernel: eight roles per tile; plain configuration with the normaliser, no noise).
    // Fleet sweep, 64-step launches, us per step (five waves | eight roles; profiles/r04_sweep_rp.txt): 4 096 drones 1.08 | 0.93, 8 192
    // 1.11 | 0.99, 16 384 1.18 | 1.05, 20 480 1.52 | 1.53, 24 576 1.53 | 1.57, 32 768 1.63 | 1.66, 40 960 2.79 | 2.37, 49 152 2.98 | 2.47
    // -> eight roles up to one tile per CU and from two to three tiles per CU (where two of its workgroups fit a CU and the third tile
    // follows), five waves in between, where sixteen waves saturate the vector ALUs either way.  Without the normaliser the six-role form
    // lost to the four-wave kernel at every size (32 768 drones: 1.68 against 1.40) and was removed in round 5.
    // Round 6, beyond three tiles per CU (one wave | eight roles, 64-step launches, the one-wave kernel 27 % faster than it was:
    // profiles/r06_sweep_large.txt): 57 344 drones 3.41 | 2.53, 65 536 3.53 | 2.70, 81 920 4.56 | 3.37, 98 304 4.67 | 3.93, 114 688
    // 4.76 | 4.58 (K = 20: 5.07 | 5.22), 131 072 4.86 | 5.16 -> eight roles up to six tiles per CU, one wave beyond.
    const bool rp_ok = plain && !cfg->ground_contact && !noisy;
    // ... and from two to three tiles per CU the four-wave kernel WITH the normaliser on its X wave, once its statistics are addressed from a walked
    // scalar pair (195 -> 149 registers: three tiles = twelve waves per CU resident; four waves | five | eight roles, profiles/r06_sweep_4wnorm.txt):
    // 34 816 drones 1.80 | 2.06 | 1.93, 40 960 1.83 | 2.13 | 1.96, 49 152 1.95 | 2.23 | 2.06, 53 248 2.66 | 3.14 | 2.44 -> four waves in (2, 3] tiles per CU.
    if (rp_ok && cfg->normalize_obs && (e->blocks <= e->num_cus || (e->blocks > 3 * e->num_cus && e->blocks <= DN_ROLE_PIPE_TILES_PER_CU * e->num_cus)))
        e->waves_fused = 8;
    // dn_step (one control step per launch) is latency bound: 1.65 us of kernel boundary that an empty kernel already pays
    // (profiles/r03_dispatch_floor.txt; the 2.9 us of round 2 was the host's eager launch cadence), a ~1.2 us memory round trip
    // with nothing to overlap it, plus the dependent instruction stream of the step.  Cutting the step
    // by dependency over three waves (dn_step_pqx_kernel) shortens that stream while the chip has idle SIMDs; built for the
    // plain configuration without the ground-contact term.  DN_WAVES_SINGLE=1|3 overrides the pick (sweeps).
    // Measured (profiles/r02_sweep_single.txt, us per step, one wave / three waves): 32768 drones 6.4 / 4.7, 65536: 7.7 / 6.8,
    // 98304: 9.5 / 9.8; with the normaliser 32768: 8.7 / 6.8, 49152: 9.5 / 10.9 -> three waves up to 1024 tiles, 512 with the
    // normaliser's statistics (27 float64 per drone through one wave's loads and stores).
    const bool pqx_ok = plain && !cfg->ground_contact;
    e->waves_single = (pqx_ok && e->blocks <= (cfg->normalize_obs ? DN_PQX_MAX_TILES / 2 : DN_PQX_MAX_TILES)) ? 3 : 1;
    if (const char *w = getenv("DN_WAVES")) {
        if (w[0] == '1') e->waves_fused = e->waves_single = 1;
        else if (w[0] == '2') e->waves_fused = e->waves_single = 2;
        else if (w[0] == '3') {
            e->waves_fused = 3;
            e->waves_single = pqx_ok ? 3 : 1;
        } else if (w[0] == '4') {                              // the recurrence itself on two waves (dn_step_many_4w_kernel): plain configuration only
            e->waves_fused = pqx_ok ? 4 : 3;
            e->waves_single = pqx_ok ? 3 : 1;
        } else if (w[0] == '5') {                              // + the normaliser on a fifth wave (normaliser on only)
            e->waves_fused = pqx_ok ? (cfg->normalize_obs ? 5 : 4) : 3;
            e->waves_single = pqx_ok ? 3 : 1;
        } else if (w[0] == '8') {                              // the role-pipelined kernel (eight roles; normaliser on, no noise)
            e->waves_fused = rp_ok && cfg->normalize_obs ? 8 : (pqx_ok ? (cfg->normalize_obs ? 5 : 4) : 3);
            e->waves_single = pqx_ok ? 3 : 1;
        }
    }
    if (const char *w = getenv("DN_WAVES_SINGLE")) {
        if (w[0] == '1') e->waves_single = 1;
        else if (w[0] == '3' && pqx_ok) e->waves_single = 3;
    }
    const int drag = cfg->physics == 2 || cfg->physics == 4;
    const int pid_mode = (cfg->action_type == 2 || cfg->action_type == 3 || cfg->action_type == 5) ? cfg->action_type : 0;
    if (pid_mode) e->waves_fused = e->waves_single = 1;      // the controller reads the step's entry state: one-wave kernels (dn_kernels.hip, PidCtx)
    if (cfg->random_spawn) e->waves_fused = e->waves_single = 1;   // the spawn draw lives in the one-wave option kernels only
    const Layout L = make_layout(n, cfg->normalize_obs, cfg->norm_rew, drag, pid_mode);
    hipError_t he = hipMalloc(&e->arena, L.total);
    if (he != hipSuccess) {
        delete e;
        return fail(DN_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes) for %lld drones failed: %s", L.total, n, hipGetErrorString(he));
    }
    e->arena_bytes = L.total;
    char *base = (char *)e->arena;
    DnParams &p = e->p;
    memset(&p, 0, sizeof p);
    p.st.g0 = (float4 *)(base + L.off_g[0]); p.st.g1 = (float4 *)(base + L.off_g[1]);
    p.st.g2 = (float4 *)(base + L.off_g[2]); p.st.g3 = (float4 *)(base + L.off_g[3]);
    p.st.g4 = (float4 *)(base + L.off_g[4]); p.st.g5 = (float4 *)(base + L.off_g[5]);
    p.st.g6 = (float4 *)(base + L.off_g[6]);
    p.st.g7 = drag ? (float4 *)(base + L.off_g7) : nullptr;
    p.st.rms_mean = (double *)(base + L.off_mean);
    p.st.rms_m2 = (double *)(base + L.off_var);
    p.st.rms_count = (double *)(base + L.off_count);
    p.st.rr = (double *)(base + L.off_rr);
    p.st.pid = pid_mode ? (double *)(base + L.off_pid) : nullptr;
    p.st.stats = (DnStatSlot *)(base + L.off_stats);
    e->tab64 = (double *)(base + L.off_tab64);
    e->tab32 = (float *)(base + L.off_tab32);
    p.tab64 = e->tab64;
    p.tab32 = e->tab32;
    p.n = n;
    p.num_waypoints = cfg->num_waypoints;
    p.num_cus = e->num_cus;
    p.max_steps = cfg->max_steps;
    p.circle = cfg->circle != 0; p.cylinder = cfg->cylinder != 0; p.include_distance = cfg->include_distance != 0;
    p.normalize_actions = cfg->normalize_actions != 0; p.normalize_obs = cfg->normalize_obs != 0;
    p.ground_contact = cfg->ground_contact != 0;
    p.clip_rew = cfg->clip_rew != 0; p.norm_rew = cfg->norm_rew != 0;
    p.gnd = cfg->physics == 1 || cfg->physics == 4; p.drag = drag; p.rpm_actions = cfg->action_type == 1 ? 1 : (cfg->action_type == 4 ? 2 : 0);
    p.pid_mode = pid_mode;
    p.random_spawn = cfg->random_spawn != 0;
    p.zero_damping = cfg->zero_damping != 0;
    {   // DN_EXACT_OBS_NOISE=1: observation noise in the exact float64 Box-Muller form (reproducible across GPU generations; ~10 % slower noisy steps)
        // DN_EXACT_NORM=1: the normaliser's output stage in float64 (the float32 nearest to the float64 evaluation instead of <= 3 ulp) -- a
        // compile-time form (dn_kernels.hip, normalize_obs_cols) that lives in libdronenav_exact.so: here the request is only CHECKED against the build.
        // Both are readable back through dn_get_exact_flags; a value that is neither a yes nor a no fails the create.
        int v = env_switch("DN_EXACT_OBS_NOISE");
        if (v < 0) { (void)hipFree(e->arena); delete e; return fail(DN_ERR_INVALID_ARGUMENT, "DN_EXACT_OBS_NOISE=%s: expected 1 | true | on | yes or 0 | false | off | no", getenv("DN_EXACT_OBS_NOISE")); }
        p.exact_obs_noise = v;
        v = env_switch("DN_EXACT_NORM");
        if (v < 0) { (void)hipFree(e->arena); delete e; return fail(DN_ERR_INVALID_ARGUMENT, "DN_EXACT_NORM=%s: expected 1 | true | on | yes or 0 | false | off | no", getenv("DN_EXACT_NORM")); }
        if (v == 1 && !dn_norm_exact_compiled_in()) {
            (void)hipFree(e->arena); delete e;
            return fail(DN_ERR_INVALID_ARGUMENT, "DN_EXACT_NORM=1 asks for the normaliser's float64 output stage, which is a build of its own: load "
                                                 "libdronenav_exact.so (the Python package does when the variable is set) instead of this library");
        }
    }
    p.act_noise_sigma = cfg->act_noise_sigma; p.obs_noise_sigma = cfg->obs_noise_sigma;
    p.seed = cfg->seed; p.env_id_offset = cfg->env_id_offset;
    build_consts<double>(*cfg, p.c64);
    build_consts<float>(*cfg, p.c32);

    std::vector<double> t64(DN_MAX_WAYPOINTS * DN_T_STRIDE, 0.0);
    std::vector<float> t32(DN_MAX_WAYPOINTS * DN_T_STRIDE, 0.0f);
    build_table(*cfg, t64.data());
    for (size_t j = 0; j < t64.size(); ++j) t32[j] = (float)t64[j];
    int32_t st = DN_OK;
    do {
        if (hipMemcpy(e->tab64, t64.data(), t64.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e->tab32, t32.data(), t32.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            st = fail(DN_ERR_HIP, "uploading the waypoint table failed");
            break;
        }
        st = init_state(e, nullptr);
        if (st != DN_OK) break;
        if (cfg->random_spawn) {                // make_env's env.reset(seed=...) (PBDroneSimulator.py:173) already draws a spawn point
            float *scratch = nullptr;
            if (hipMalloc(&scratch, (size_t)n * DN_OBS_DIM * sizeof(float)) != hipSuccess) { st = fail(DN_ERR_OUT_OF_MEMORY, "scratch observation buffer"); break; }
            const bool launched = dn_launch_reset(e->p, scratch, cfg->compute_f32 != 0, nullptr) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
            (void)hipFree(scratch);
            if (!launched) { st = fail(DN_ERR_HIP, "initial random spawn failed"); break; }
            if (cfg->normalize_obs) {           // that reset happens BEFORE the NormalizeObservation wrapper exists: statistics stay pristine
                st = init_state_rms(e, nullptr);
                if (st != DN_OK) break;
            }
        }
        if (hipStreamSynchronize(nullptr) != hipSuccess) { st = fail(DN_ERR_HIP, "state initialisation failed"); break; }
    } while (0);
    if (st != DN_OK) {
        (void)hipFree(e->arena);
        delete e;
        return st;
    }
    *out = e;
    return DN_OK;
}

int32_t dn_destroy(dn_env *env)
{
    if (!env) return DN_OK;
    hipError_t he = hipSuccess;
    if (env->arena) {
        (void)hipSetDevice(env->cfg.device_id);
        he = hipFree(env->arena);
    }
    delete env;
    if (he != hipSuccess) return fail(DN_ERR_HIP, "hipFree failed: %s", hipGetErrorString(he));
    return DN_OK;
}

int64_t dn_num_envs(const dn_env *env) { return env ? env->cfg.num_envs : 0; }

int32_t dn_get_config(const dn_env *env, dn_config *out)
{
    if (!env || !out) return fail(DN_ERR_INVALID_ARGUMENT, "env and out are required");
    *out = env->cfg;                                  // ground_contact resolved to 0 / 1
    return DN_OK;
}

int32_t dn_get_num_cus(const dn_env *env) { return env ? env->num_cus : 0; }

int32_t dn_get_exact_flags(const dn_env *env)
{
    if (!env) return 0;
    return (env->p.exact_obs_noise ? DN_EXACT_FLAG_OBS_NOISE : 0) | (dn_norm_exact_compiled_in() ? DN_EXACT_FLAG_NORM : 0);
}

int32_t dn_resolve_ground_contact(const dn_config *cfg)
{
    const int32_t rc = validate(cfg);
    if (rc != DN_OK) return rc;
    if (cfg->ground_contact != DN_GROUND_CONTACT_AUTO) return cfg->ground_contact;
    return ground_contact_reachable(*cfg) ? 1 : 0;
}

int32_t dn_reset(dn_env *env, float *obs, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    if (!obs) return fail(DN_ERR_INVALID_ARGUMENT, "obs is NULL");
    DN_HIP(dn_launch_reset(env->p, obs, env->cfg.compute_f32 != 0, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_step(dn_env *env, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *truncated,
                int32_t *found_targets, float *terminal_obs, float *ep_return, int32_t *ep_length,
                uint64_t *done_mask, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    const LaunchEvents armed(env);                         // dn_set_launch_events: consumed by this call, or dropped if it fails below
    if (!actions || !obs || !reward || !done || !truncated || !found_targets)
        return fail(DN_ERR_INVALID_ARGUMENT, "actions, obs, reward, done, truncated and found_targets are required");
    if (((uintptr_t)actions & 15u) || ((uintptr_t)obs & 15u))
        return fail(DN_ERR_INVALID_ARGUMENT, "actions and obs must be 16-byte aligned");
    DN_REFUSE_ARMED_CAPTURE(stream);
    DnStepIO io;
    io.actions = actions; io.obs = obs; io.reward = reward; io.done = done; io.truncated = truncated;
    io.found_targets = found_targets; io.terminal_obs = terminal_obs; io.ep_return = ep_return;
    io.ep_length = ep_length; io.done_mask = (unsigned long long *)done_mask;
    io.mean = nullptr; io.act_out = nullptr; io.logp_out = nullptr; io.sample_squash = 0;
    DN_HIP(dn_launch_step_many(env->p, io, 1, env->cfg.compute_f32 != 0, env->waves_single, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_step_sampled(dn_env *env, const float *mean, const float *log_std, uint64_t seed, int32_t deterministic,
                        float *actions_out, float *log_prob_out, float *obs, float *reward, uint8_t *done, uint8_t *truncated,
                        int32_t *found_targets, float *terminal_obs, float *ep_return, int32_t *ep_length,
                        uint64_t *done_mask, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    const LaunchEvents armed(env);                         // dn_set_launch_events: this launch takes the hook as well
    if (!mean || !log_std || !actions_out || !log_prob_out || !obs || !reward || !done || !truncated || !found_targets)
        return fail(DN_ERR_INVALID_ARGUMENT, "mean, log_std, actions_out, log_prob_out, obs, reward, done, truncated and found_targets are required");
    if (((uintptr_t)mean & 15u) || ((uintptr_t)actions_out & 15u) || ((uintptr_t)obs & 15u))
        return fail(DN_ERR_INVALID_ARGUMENT, "mean, actions_out and obs must be 16-byte aligned");
    if (env->cfg.clip_rew || env->cfg.norm_rew || env->cfg.physics != 0 || env->cfg.action_type != 0 || env->cfg.random_spawn || env->cfg.zero_damping)
        return fail(DN_ERR_INVALID_ARGUMENT, "dn_step_sampled is built for the configuration without reward wrappers / extra physics terms / RPM actions / random spawn / zero damping; "
                                             "use dn_policy_sample + dn_step there");
    DnStepIO io;
    io.actions = nullptr; io.obs = obs; io.reward = reward; io.done = done; io.truncated = truncated;
    io.found_targets = found_targets; io.terminal_obs = terminal_obs; io.ep_return = ep_return;
    io.ep_length = ep_length; io.done_mask = (unsigned long long *)done_mask;
    io.mean = mean; io.act_out = actions_out; io.logp_out = log_prob_out;
    for (int j = 0; j < 4; ++j) io.log_std[j] = log_std[j];
    io.sample_seed = seed; io.sample_deterministic = deterministic != 0; io.sample_squash = 0;
    // the sampling lives in the single-step kernels (one wave, or three waves cut by dependency)
    DN_REFUSE_ARMED_CAPTURE(stream);
    DN_HIP(dn_launch_step_many(env->p, io, 1, env->cfg.compute_f32 != 0, env->waves_single == 3 ? 3 : 1, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_step_squashed(dn_env *env, const float *mu_log_std, uint64_t seed, int32_t deterministic, float *actions_out,
                         float *log_prob_out, float *obs, float *reward, uint8_t *done, uint8_t *truncated, int32_t *found_targets,
                         float *terminal_obs, float *ep_return, int32_t *ep_length, uint64_t *done_mask, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    const LaunchEvents armed(env);                         // dn_set_launch_events: this launch takes the hook as well
    if (!mu_log_std || !actions_out || !obs || !reward || !done || !truncated || !found_targets)
        return fail(DN_ERR_INVALID_ARGUMENT, "mu_log_std, actions_out, obs, reward, done, truncated and found_targets are required");
    if (((uintptr_t)mu_log_std & 15u) || ((uintptr_t)actions_out & 15u) || ((uintptr_t)obs & 15u))
        return fail(DN_ERR_INVALID_ARGUMENT, "mu_log_std, actions_out and obs must be 16-byte aligned");
    if (env->cfg.clip_rew || env->cfg.norm_rew || env->cfg.physics != 0 || env->cfg.action_type != 0 || env->cfg.random_spawn || env->cfg.zero_damping)
        return fail(DN_ERR_INVALID_ARGUMENT, "dn_step_squashed is built for the configuration without reward wrappers / extra physics terms / RPM actions / random spawn / zero damping; "
                                             "use dn_squashed_sample + dn_step there");
    DnStepIO io;
    io.actions = nullptr; io.obs = obs; io.reward = reward; io.done = done; io.truncated = truncated;
    io.found_targets = found_targets; io.terminal_obs = terminal_obs; io.ep_return = ep_return;
    io.ep_length = ep_length; io.done_mask = (unsigned long long *)done_mask;
    io.mean = mu_log_std; io.act_out = actions_out; io.logp_out = log_prob_out;
    for (int j = 0; j < 4; ++j) io.log_std[j] = 0.0f;
    io.sample_seed = seed; io.sample_deterministic = deterministic != 0; io.sample_squash = 1;
    DN_REFUSE_ARMED_CAPTURE(stream);
    DN_HIP(dn_launch_step_many(env->p, io, 1, env->cfg.compute_f32 != 0, env->waves_single == 3 ? 3 : 1, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_mlp_step_sampled(dn_env *env, const dn_mlp_net *nets, int32_t num_nets, const float *policy_obs, int32_t obs_dim,
                            const float *log_std, uint64_t seed, int32_t deterministic, float *actions_out, float *log_prob_out,
                            float *obs, float *reward, uint8_t *done, uint8_t *truncated, int32_t *found_targets, float *terminal_obs,
                            float *ep_return, int32_t *ep_length, uint64_t *done_mask, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    const LaunchEvents dropped(env);                       // no hook on this launch: armed events are dropped here, not left for a later call
    dn_tl_ev_start = dn_tl_ev_stop = nullptr;
    if (!nets || !policy_obs || !log_std || !actions_out || !log_prob_out || !obs || !reward || !done || !truncated || !found_targets)
        return fail(DN_ERR_INVALID_ARGUMENT, "nets, policy_obs, log_std, actions_out, log_prob_out, obs, reward, done, truncated and found_targets are required");
    if (num_nets < 1 || num_nets > 2) return fail(DN_ERR_INVALID_ARGUMENT, "num_nets must be 1 (actor) or 2 (actor, critic)");
    if (obs_dim < 1 || obs_dim > 16) return fail(DN_ERR_INVALID_ARGUMENT, "obs_dim must be in 1..16 (got %d)", obs_dim);
    if (((uintptr_t)actions_out & 15u) || ((uintptr_t)obs & 15u)) return fail(DN_ERR_INVALID_ARGUMENT, "actions_out and obs must be 16-byte aligned");
    const dn_config &c = env->cfg;
    for (int k = 0; k < num_nets; ++k) {
        const dn_mlp_net &n = nets[k];
        if (n.arch != DN_MLP_ARCH_PPO) return fail(DN_ERR_INVALID_ARGUMENT, "net %d: dn_mlp_step_sampled runs the PPO networks (arch 0)", k);
        if (!n.w1 || !n.w2 || !n.w3 || !n.wh || !n.b1 || !n.b2 || !n.b3 || !n.bh || !n.out)
            return fail(DN_ERR_INVALID_ARGUMENT, "net %d: every weight, bias and output pointer is required", k);
        if (n.grade < 0 || n.grade > 2 || n.grade != nets[0].grade) return fail(DN_ERR_INVALID_ARGUMENT, "net %d: grade must be 0 | 1 | 2 and the same for both networks", k);
        if (n.out_dim < 1 || n.out_dim > 32) return fail(DN_ERR_INVALID_ARGUMENT, "net %d: out_dim must be in 1..32", k);
        if (((uintptr_t)n.w1 | (uintptr_t)n.w2 | (uintptr_t)n.w3 | (uintptr_t)n.wh) & 15u)
            return fail(DN_ERR_INVALID_ARGUMENT, "net %d: packed weights must be 16-byte aligned", k);
    }
    if (nets[0].out_dim != DN_ACT_DIM) return fail(DN_ERR_INVALID_ARGUMENT, "nets[0] must be the actor (out_dim 4)");
    {   // one launch: the critic's workgroups and later actor workgroups still READ policy_obs while earlier actor tails already WRITE obs
        // and the networks' outputs -- overlapping buffers would be a silent data race, so they are refused
        const long long nn = c.num_envs;
        const uintptr_t pb = (uintptr_t)policy_obs, pe = pb + (uintptr_t)nn * obs_dim * sizeof(float);
        const uintptr_t ob = (uintptr_t)obs, oe = ob + (uintptr_t)nn * DN_OBS_DIM * sizeof(float);
        if (pb < oe && ob < pe) return fail(DN_ERR_INVALID_ARGUMENT, "policy_obs must not overlap obs (the launch reads one while it writes the other)");
        for (int k = 0; k < num_nets; ++k) {
            const uintptr_t nb = (uintptr_t)nets[k].out, ne = nb + (uintptr_t)nn * nets[k].out_dim * sizeof(float);
            if ((nb < pe && pb < ne) || (nb < oe && ob < ne))
                return fail(DN_ERR_INVALID_ARGUMENT, "net %d: out must not overlap policy_obs or obs", k);
        }
    }
    // the tail is the three-wave single step: the plain configuration dn_step_sampled covers, without noise and without the
    // ground-contact term, on fleets for which dn_create picked that shape; whole workgroups of the policy kernel only
    if (c.clip_rew || c.norm_rew || c.physics != 0 || c.action_type != 0 || c.random_spawn || c.zero_damping || c.ground_contact ||
        c.act_noise_sigma > 0.0f || c.obs_noise_sigma > 0.0f || c.compute_f32 || env->waves_single != 3)
        return fail(DN_ERR_INVALID_ARGUMENT, "dn_mlp_step_sampled is built for the float64 reference configuration without noise / reward wrappers / extra "
                                             "physics / ground contact on fleets that take the three-wave single step; use dn_mlp_forward + dn_step_sampled");
    const long long per_wg = nets[0].grade == 1 ? 64 : 128;
    if (c.num_envs % per_wg) return fail(DN_ERR_INVALID_ARGUMENT, "dn_mlp_step_sampled needs num_envs %% %lld == 0 (got %lld)", per_wg, (long long)c.num_envs);
    DnStepIO io;
    io.actions = nullptr; io.obs = obs; io.reward = reward; io.done = done; io.truncated = truncated;
    io.found_targets = found_targets; io.terminal_obs = terminal_obs; io.ep_return = ep_return;
    io.ep_length = ep_length; io.done_mask = (unsigned long long *)done_mask;
    io.mean = nullptr; io.act_out = actions_out; io.logp_out = log_prob_out;
    for (int j = 0; j < 4; ++j) io.log_std[j] = log_std[j];
    io.sample_seed = seed; io.sample_deterministic = deterministic != 0; io.sample_squash = 0;
    DN_HIP(hipSetDevice(c.device_id));
    DN_HIP(dn_launch_mlp_step(env->p, io, nets, num_nets, policy_obs, obs_dim, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_eval_kinematics(dn_env *env, const double *kinematics, float *obs, float *reward, uint8_t *done, uint8_t *truncated,
                           int32_t *found_targets, float *terminal_obs, float *ep_return, int32_t *ep_length, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    const LaunchEvents dropped(env);                       // no hook on this launch: armed events are dropped here, not left for a later call
    dn_tl_ev_start = dn_tl_ev_stop = nullptr;
    if (!kinematics || !obs || !reward || !done || !truncated || !found_targets)
        return fail(DN_ERR_INVALID_ARGUMENT, "kinematics, obs, reward, done, truncated and found_targets are required");
    if (((uintptr_t)kinematics & 7u) || ((uintptr_t)obs & 15u))
        return fail(DN_ERR_INVALID_ARGUMENT, "kinematics must be 8-byte and obs 16-byte aligned");
    const dn_config &c = env->cfg;
    if (c.act_noise_sigma > 0.0f || c.obs_noise_sigma > 0.0f || c.clip_rew || c.norm_rew || c.physics != 0 || c.action_type != 0 || c.random_spawn)
        return fail(DN_ERR_INVALID_ARGUMENT, "dn_eval_kinematics is built for the reference configuration (no noise, no reward wrappers, "
                                             "Physics.PYB, ActionType.THRUST, fixed spawn)");
    DnStepIO io;
    memset(&io, 0, sizeof io);
    io.obs = obs; io.reward = reward; io.done = done; io.truncated = truncated; io.found_targets = found_targets;
    io.terminal_obs = terminal_obs; io.ep_return = ep_return; io.ep_length = ep_length;
    DN_HIP(dn_launch_eval_kinematics(env->p, io, kinematics, c.compute_f32 != 0, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_step_many(dn_env *env, int64_t k, const float *actions, float *obs, float *reward, uint8_t *done,
                     uint8_t *truncated, int32_t *found_targets, float *terminal_obs, float *ep_return,
                     int32_t *ep_length, uint64_t *done_mask, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    const LaunchEvents armed(env);                         // dn_set_launch_events: consumed by this call, or dropped if it fails below
    if (k < 1) return fail(DN_ERR_INVALID_ARGUMENT, "k must be >= 1 (got %lld)", (long long)k);
    const long long n = env->cfg.num_envs;
    if (k > 1 && (n & 3)) return fail(DN_ERR_INVALID_ARGUMENT, "dn_step_many needs num_envs %% 4 == 0 (got %lld)", n);
    if (k == 1)
        return dn_step(env, actions, obs, reward, done, truncated, found_targets, terminal_obs, ep_return, ep_length,
                       done_mask, stream);
    if (!actions || !obs || !reward || !done || !truncated || !found_targets)
        return fail(DN_ERR_INVALID_ARGUMENT, "actions, obs, reward, done, truncated and found_targets are required");
    if (((uintptr_t)actions & 15u) || ((uintptr_t)obs & 15u))
        return fail(DN_ERR_INVALID_ARGUMENT, "actions and obs must be 16-byte aligned");
    if (k > (1 << 30)) return fail(DN_ERR_INVALID_ARGUMENT, "k too large");
    // one fused launch: the state stays in registers for all k steps (dn_step_many_kernel)
    DnStepIO io;
    io.actions = actions; io.obs = obs; io.reward = reward; io.done = done; io.truncated = truncated;
    io.found_targets = found_targets; io.terminal_obs = terminal_obs; io.ep_return = ep_return;
    io.ep_length = ep_length; io.done_mask = (unsigned long long *)done_mask;
    io.mean = nullptr; io.act_out = nullptr; io.logp_out = nullptr; io.sample_squash = 0;
    DN_REFUSE_ARMED_CAPTURE(stream);
    DN_HIP(dn_launch_step_many(env->p, io, (int)k, env->cfg.compute_f32 != 0, env->waves_fused, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_set_launch_events(dn_env *env, void *start_event, void *stop_event)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    env->ev_start = (hipEvent_t)start_event;
    env->ev_stop = (hipEvent_t)stop_event;
    return DN_OK;
}

int32_t dn_compact_done(const uint64_t *done_mask, int64_t num_envs, int32_t *indices, int32_t *count,
                        int32_t device_id, void *stream)
{
    if (!done_mask || !indices || !count || num_envs < 1)
        return fail(DN_ERR_INVALID_ARGUMENT, "done_mask, indices, count are required and num_envs >= 1");
    DN_HIP(hipSetDevice(device_id));
    DN_HIP(dn_launch_compact((const unsigned long long *)done_mask, num_envs, indices, count, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_pack_done(const uint64_t *done_mask, int64_t num_envs, const float *terminal_obs, const float *ep_return, const int32_t *ep_length,
                     const uint8_t *truncated, const int32_t *found_targets, int32_t *indices, int32_t *count, float *packed,
                     int32_t device_id, void *stream)
{
    if (!done_mask || !terminal_obs || !ep_return || !ep_length || !truncated || !found_targets || !indices || !count || !packed || num_envs < 1)
        return fail(DN_ERR_INVALID_ARGUMENT, "every pointer is required and num_envs >= 1");
    DN_HIP(hipSetDevice(device_id));
    DN_HIP(dn_launch_compact_pack((const unsigned long long *)done_mask, num_envs, terminal_obs, ep_return, ep_length, truncated, found_targets,
                                  indices, count, packed, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_stream_copy(void *dst, const void *src, int64_t bytes, int32_t device_id, void *stream)
{
    if (!dst || !src || bytes < 16 || (bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return fail(DN_ERR_INVALID_ARGUMENT, "dst, src: 16-byte aligned device pointers; bytes: a positive multiple of 16");
    DN_HIP(hipSetDevice(device_id));
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus < 1) cus = 256;
    DN_HIP(dn_launch_stream_copy(dst, src, bytes / 16, cus, (hipStream_t)stream));
    return DN_OK;
}

// Monitor's running return is held as the float32 hi (g4.w) + k * ulp(hi) / 256 with k the signed top byte of the length
// word (dn_kernels.hip report_scalars); dn_env_state shows the low part as a float.
static float ret_lo_decode(float hi, int k)
{
    uint32_t bits;
    memcpy(&bits, &hi, 4);
    const int eb = (int)((bits >> 23) & 0xFFu);
    return (eb > 31 && eb != 255) ? (float)ldexp((double)k, eb - 127 - 31) : 0.0f;      // inf / NaN carry no low part (ret_lo_encode)
}
static int ret_lo_encode(float hi, float lo)
{
    uint32_t bits;
    memcpy(&bits, &hi, 4);
    const int eb = (int)((bits >> 23) & 0xFFu);
    if (eb <= 31 || eb == 255) return 0;
    const double q = nearbyint(ldexp((double)lo, 127 + 31 - eb));
    return (int)fmin(fmax(q, -128.0), 127.0) & 0xFF;
}

int32_t dn_get_state(dn_env *env, dn_env_state *states, int64_t count)
{
    if (!env || !states) return fail(DN_ERR_INVALID_ARGUMENT, "env and states are required");
    const long long n = env->cfg.num_envs;
    if (count != n) return fail(DN_ERR_INVALID_ARGUMENT, "count (%lld) != num_envs (%lld)", (long long)count, n);
    DN_HIP(hipSetDevice(env->cfg.device_id));
    DN_HIP(hipDeviceSynchronize());
    std::vector<float4> g[7];
    float4 *src[7] = {env->p.st.g0, env->p.st.g1, env->p.st.g2, env->p.st.g3, env->p.st.g4, env->p.st.g5, env->p.st.g6};
    for (int k = 0; k < 7; ++k) {
        g[k].resize((size_t)n);
        DN_HIP(hipMemcpy(g[k].data(), src[k], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    std::vector<double> mean, var, cnt;
    if (env->cfg.normalize_obs) {
        mean.resize((size_t)n * DN_OBS_DIM); var.resize((size_t)n * DN_OBS_DIM); cnt.resize((size_t)n);
        DN_HIP(hipMemcpy(mean.data(), env->p.st.rms_mean, mean.size() * sizeof(double), hipMemcpyDeviceToHost));
        DN_HIP(hipMemcpy(var.data(), env->p.st.rms_m2, var.size() * sizeof(double), hipMemcpyDeviceToHost));
        DN_HIP(hipMemcpy(cnt.data(), env->p.st.rms_count, cnt.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    std::vector<double> rr;
    if (env->cfg.norm_rew) {
        rr.resize((size_t)n * 4);
        DN_HIP(hipMemcpy(rr.data(), env->p.st.rr, rr.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    std::vector<float4> g7;
    if (env->p.drag) {
        g7.resize((size_t)n);
        DN_HIP(hipMemcpy(g7.data(), env->p.st.g7, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    std::vector<double> pid;
    if (env->p.pid_mode) {
        pid.resize((size_t)n * 9);
        DN_HIP(hipMemcpy(pid.data(), env->p.st.pid, pid.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    for (long long i = 0; i < n; ++i) {
        dn_env_state &s = states[i];
        memset(&s, 0, sizeof s);
        if (env->p.pid_mode) for (int k = 0; k < 9; ++k) s.pid[k] = pid[(size_t)k * n + i];
        if (env->p.drag) { s.last_rpm[0] = g7[(size_t)i].x; s.last_rpm[1] = g7[(size_t)i].y; s.last_rpm[2] = g7[(size_t)i].z; s.last_rpm[3] = g7[(size_t)i].w; }
        if (env->cfg.norm_rew) { s.rr_returns = rr[(size_t)i]; s.rr_mean = rr[(size_t)n + i]; s.rr_var = rr[(size_t)2 * n + i]; s.rr_count = rr[(size_t)3 * n + i]; }
        s.pos[0] = g[0][i].x; s.pos[1] = g[0][i].y; s.pos[2] = g[0][i].z; s.d = g[0][i].w;
        s.quat[0] = g[1][i].x; s.quat[1] = g[1][i].y; s.quat[2] = g[1][i].z; s.quat[3] = g[1][i].w;
        s.vel[0] = g[2][i].x; s.vel[1] = g[2][i].y; s.vel[2] = g[2][i].z; s.d_prev = g[2][i].w;
        s.ang_v[0] = g[3][i].x; s.ang_v[1] = g[3][i].y; s.ang_v[2] = g[3][i].z;
        uint32_t meta;
        memcpy(&meta, &g[3][i].w, 4);
        s.steps = (int32_t)(meta & 0xFFFFFFu); s.idx = (int32_t)((meta >> 24) & 0x7Fu); s.just_found = (int32_t)(meta >> 31);
        s.prev_vel[0] = g[4][i].x; s.prev_vel[1] = g[4][i].y; s.prev_vel[2] = g[4][i].z; s.ep_ret = g[4][i].w;
        s.prev_ang_v[0] = g[5][i].x; s.prev_ang_v[1] = g[5][i].y; s.prev_ang_v[2] = g[5][i].z;
        int32_t lenword;
        memcpy(&lenword, &g[5][i].w, 4);
        s.ep_len = lenword & 0xFFFFFF;
        s.ep_ret_lo = ret_lo_decode(s.ep_ret, lenword >> 24);
        // _current_position equals pos once a post-step has run (steps > 0); the stored copy is the stale one
        if (s.steps > 0) { s.cur_pos[0] = s.pos[0]; s.cur_pos[1] = s.pos[1]; s.cur_pos[2] = s.pos[2]; }
        else { s.cur_pos[0] = g[6][i].x; s.cur_pos[1] = g[6][i].y; s.cur_pos[2] = g[6][i].z; }
        if (env->cfg.normalize_obs) {
            for (int k = 0; k < DN_OBS_DIM; ++k) { s.rms_mean[k] = mean[(size_t)k * n + i]; s.rms_var[k] = var[(size_t)k * n + i] / cnt[(size_t)i]; }      // the device holds the second moment var x count
            s.rms_count = cnt[(size_t)i];
        }
    }
    return DN_OK;
}

int32_t dn_set_state(dn_env *env, const dn_env_state *states, int64_t count)
{
    if (!env || !states) return fail(DN_ERR_INVALID_ARGUMENT, "env and states are required");
    const long long n = env->cfg.num_envs;
    if (count != n) return fail(DN_ERR_INVALID_ARGUMENT, "count (%lld) != num_envs (%lld)", (long long)count, n);
    std::vector<float4> g[7];
    for (int k = 0; k < 7; ++k) g[k].resize((size_t)n);
    std::vector<double> mean, var, cnt;
    if (env->cfg.normalize_obs) { mean.resize((size_t)n * DN_OBS_DIM); var.resize((size_t)n * DN_OBS_DIM); cnt.resize((size_t)n); }
    std::vector<double> rr;
    if (env->cfg.norm_rew) rr.resize((size_t)n * 4);
    std::vector<float4> g7;
    if (env->p.drag) g7.resize((size_t)n);
    std::vector<double> pid;
    if (env->p.pid_mode) pid.resize((size_t)n * 9);
    for (long long i = 0; i < n; ++i) {
        const dn_env_state &s = states[i];
        if (env->p.pid_mode) for (int k = 0; k < 9; ++k) pid[(size_t)k * n + i] = s.pid[k];
        if (env->p.drag) g7[(size_t)i] = make_float4(s.last_rpm[0], s.last_rpm[1], s.last_rpm[2], s.last_rpm[3]);
        if (env->cfg.norm_rew) { rr[(size_t)i] = s.rr_returns; rr[(size_t)n + i] = s.rr_mean; rr[(size_t)2 * n + i] = s.rr_var; rr[(size_t)3 * n + i] = s.rr_count; }
        if (s.idx < 0 || s.idx >= env->cfg.num_waypoints || s.steps < 0 || s.steps > (1 << 24) - 1)
            return fail(DN_ERR_INVALID_ARGUMENT, "state %lld: idx/steps out of range", i);
        if (s.steps > 0 && (s.cur_pos[0] != s.pos[0] || s.cur_pos[1] != s.pos[1] || s.cur_pos[2] != s.pos[2]))
            return fail(DN_ERR_BAD_STATE, "state %lld: _current_position must equal pos once _steps > 0", i);
        uint32_t meta = ((uint32_t)s.steps & 0xFFFFFFu) | (((uint32_t)s.idx & 0x7Fu) << 24) | ((uint32_t)(s.just_found != 0) << 31);
        float fmeta, flen;
        memcpy(&fmeta, &meta, 4);
        if (s.ep_len < 0 || s.ep_len > 0xFFFFFF) return fail(DN_ERR_INVALID_ARGUMENT, "state %lld: ep_len out of range", i);
        const int32_t lenword = s.ep_len | (int32_t)((uint32_t)ret_lo_encode(s.ep_ret, s.ep_ret_lo) << 24);
        memcpy(&flen, &lenword, 4);
        g[0][i] = make_float4(s.pos[0], s.pos[1], s.pos[2], s.d);
        g[1][i] = make_float4(s.quat[0], s.quat[1], s.quat[2], s.quat[3]);
        g[2][i] = make_float4(s.vel[0], s.vel[1], s.vel[2], s.d_prev);
        g[3][i] = make_float4(s.ang_v[0], s.ang_v[1], s.ang_v[2], fmeta);
        g[4][i] = make_float4(s.prev_vel[0], s.prev_vel[1], s.prev_vel[2], s.ep_ret);
        g[5][i] = make_float4(s.prev_ang_v[0], s.prev_ang_v[1], s.prev_ang_v[2], flen);
        g[6][i] = make_float4(s.cur_pos[0], s.cur_pos[1], s.cur_pos[2], 0.0f);
        if (env->cfg.normalize_obs) {
            if (!(s.rms_count > 0.0)) return fail(DN_ERR_INVALID_ARGUMENT, "state %lld: rms_count must be > 0 (RunningMeanStd starts at 1e-4)", i);
            for (int k = 0; k < DN_OBS_DIM; ++k) { mean[(size_t)k * n + i] = s.rms_mean[k]; var[(size_t)k * n + i] = second_moment(s.rms_var[k], s.rms_count); }
            cnt[(size_t)i] = s.rms_count;
        }
    }
    DN_HIP(hipSetDevice(env->cfg.device_id));
    DN_HIP(hipDeviceSynchronize());
    float4 *dst[7] = {env->p.st.g0, env->p.st.g1, env->p.st.g2, env->p.st.g3, env->p.st.g4, env->p.st.g5, env->p.st.g6};
    for (int k = 0; k < 7; ++k) DN_HIP(hipMemcpy(dst[k], g[k].data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice));
    if (env->cfg.normalize_obs) {
        DN_HIP(hipMemcpy(env->p.st.rms_mean, mean.data(), mean.size() * sizeof(double), hipMemcpyHostToDevice));
        DN_HIP(hipMemcpy(env->p.st.rms_m2, var.data(), var.size() * sizeof(double), hipMemcpyHostToDevice));
        DN_HIP(hipMemcpy(env->p.st.rms_count, cnt.data(), cnt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (env->cfg.norm_rew) DN_HIP(hipMemcpy(env->p.st.rr, rr.data(), rr.size() * sizeof(double), hipMemcpyHostToDevice));
    if (env->p.drag) DN_HIP(hipMemcpy(env->p.st.g7, g7.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice));
    if (env->p.pid_mode) DN_HIP(hipMemcpy(env->p.st.pid, pid.data(), pid.size() * sizeof(double), hipMemcpyHostToDevice));
    return DN_OK;
}

int32_t dn_get_stats(dn_env *env, dn_stats *out, void *stream)
{
    if (!env || !out) return fail(DN_ERR_INVALID_ARGUMENT, "env and out are required");
    std::vector<DnStatSlot> slots((size_t)env->blocks);
    DN_HIP(hipMemcpyAsync(slots.data(), env->p.st.stats, slots.size() * sizeof(DnStatSlot), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DN_HIP(hipStreamSynchronize((hipStream_t)stream));
    memset(out, 0, sizeof *out);
    long long fix = 0;
    for (const DnStatSlot &s : slots) {
        out->episodes += s.episodes; out->truncated += s.truncated; out->completed += s.completed;
        out->sum_ep_len += s.sum_len; out->sum_found_targets += s.sum_found; fix += s.sum_ret_fix;
    }
    out->sum_ep_return = (double)fix * 1e-6;
    out->env_steps = (int64_t)slots[0].step_count * env->cfg.num_envs;
    return DN_OK;
}

int32_t dn_reset_stats(dn_env *env, void *stream)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    // the slots also hold the vector-step counter, which a statistics reset must not rewind
    uint64_t steps = 0;
    int32_t rc = dn_get_step_count(env, &steps);
    if (rc != DN_OK) return rc;
    DN_HIP(hipMemsetAsync(env->p.st.stats, 0, (size_t)env->blocks * sizeof(DnStatSlot), (hipStream_t)stream));
    DN_HIP(dn_launch_set_step_count(env->p.st.stats, env->blocks, steps, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_get_kernel_waves(const dn_env *env, int32_t fused) { return env ? (fused ? env->waves_fused : env->waves_single) : 0; }

int32_t dn_get_step_count(const dn_env *env, uint64_t *out)
{
    if (!env || !out) return fail(DN_ERR_INVALID_ARGUMENT, "env and out are required");
    // the counter lives on the device (it advances under hipGraph replay too); every tile carries the same value
    DN_HIP(hipSetDevice(env->cfg.device_id));
    DN_HIP(hipDeviceSynchronize());
    unsigned long long v = 0;
    DN_HIP(hipMemcpy(&v, &env->p.st.stats[0].step_count, sizeof v, hipMemcpyDeviceToHost));
    *out = v;
    return DN_OK;
}

int32_t dn_set_step_count(dn_env *env, uint64_t value)
{
    if (!env) return fail(DN_ERR_INVALID_ARGUMENT, "env is NULL");
    DN_HIP(hipSetDevice(env->cfg.device_id));
    DN_HIP(hipDeviceSynchronize());
    DN_HIP(dn_launch_set_step_count(env->p.st.stats, env->blocks, value, nullptr));
    DN_HIP(hipStreamSynchronize(nullptr));
    return DN_OK;
}

int32_t dn_preprocess_action(const float *actions, int64_t num_envs, int32_t normalize_actions, float *rpm, float *forces,
                             float *z_torque, int32_t device_id, void *stream)
{
    if (!actions || num_envs < 1) return fail(DN_ERR_INVALID_ARGUMENT, "actions is required and num_envs >= 1");
    if (!rpm && !forces && !z_torque) return fail(DN_ERR_INVALID_ARGUMENT, "at least one output buffer is required");
    if (((uintptr_t)actions & 15u) || ((uintptr_t)rpm & 15u) || ((uintptr_t)forces & 15u))
        return fail(DN_ERR_INVALID_ARGUMENT, "actions, rpm and forces must be 16-byte aligned");
    DN_HIP(hipSetDevice(device_id));
    DN_HIP(dn_launch_action_chain(actions, num_envs, normalize_actions, rpm, forces, z_torque, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_policy_sample(dn_env *env, const float *mean, const float *log_std, uint64_t seed, int32_t deterministic, float *actions,
                         float *clipped, float *log_prob, void *stream)
{
    if (!env || !mean || !log_std || !actions || !clipped || !log_prob)
        return fail(DN_ERR_INVALID_ARGUMENT, "env, mean, log_std, actions, clipped and log_prob are required");
    if (((uintptr_t)mean | (uintptr_t)actions | (uintptr_t)clipped) & 15u)
        return fail(DN_ERR_INVALID_ARGUMENT, "mean, actions and clipped must be 16-byte aligned");
    DN_HIP(dn_launch_policy_sample(env->p, mean, log_std, seed, deterministic, actions, clipped, log_prob, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_squashed_sample(dn_env *env, const float *mu_log_std, uint64_t seed, int32_t deterministic, float *actions, float *log_prob,
                           void *stream)
{
    if (!env || !mu_log_std || !actions) return fail(DN_ERR_INVALID_ARGUMENT, "env, mu_log_std and actions are required");
    if (((uintptr_t)mu_log_std | (uintptr_t)actions) & 15u)
        return fail(DN_ERR_INVALID_ARGUMENT, "mu_log_std and actions must be 16-byte aligned");
    DN_HIP(dn_launch_squashed_sample(env->p, mu_log_std, seed, deterministic, actions, log_prob, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_add_bootstrap(float *reward, const float *terminal_value, const uint8_t *truncated, double gamma, int64_t num_envs,
                         int32_t device_id, void *stream)
{
    if (!reward || !terminal_value || !truncated || num_envs < 1)
        return fail(DN_ERR_INVALID_ARGUMENT, "reward, terminal_value, truncated are required and num_envs >= 1");
    DN_HIP(hipSetDevice(device_id));
    DN_HIP(dn_launch_add_bootstrap(reward, terminal_value, truncated, (float)gamma, num_envs, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_mlp_forward(const dn_mlp_net *nets, int32_t num_nets, const float *obs, const uint8_t *row_mask, int64_t num_envs,
                       int32_t obs_dim, int32_t device_id, void *stream)
{
    if (!nets || !obs) return fail(DN_ERR_INVALID_ARGUMENT, "nets and obs are required");
    if (num_nets < 1 || num_nets > 2) return fail(DN_ERR_INVALID_ARGUMENT, "num_nets must be 1 or 2 (got %d)", num_nets);
    if (num_envs < 1) return fail(DN_ERR_INVALID_ARGUMENT, "num_envs must be >= 1");
    if (obs_dim < 1 || obs_dim > 16) return fail(DN_ERR_INVALID_ARGUMENT, "obs_dim must be in 1..16 (got %d)", obs_dim);
    for (int k = 0; k < num_nets; ++k) {
        const dn_mlp_net &n = nets[k];
        if (n.arch != DN_MLP_ARCH_PPO && n.arch != DN_MLP_ARCH_SAC)
            return fail(DN_ERR_INVALID_ARGUMENT, "net %d: arch must be 0 (PPO 512-512-256 Tanh) or 1 (SAC actor 256-256 ReLU)", k);
        const bool three = n.arch == DN_MLP_ARCH_PPO;
        if (!n.w1 || !n.w2 || (three && !n.w3) || !n.wh || !n.b1 || !n.b2 || (three && !n.b3) || !n.bh || !n.out)
            return fail(DN_ERR_INVALID_ARGUMENT, "net %d: every weight, bias and output pointer is required", k);
        if (n.out_dim < 1 || n.out_dim > 32) return fail(DN_ERR_INVALID_ARGUMENT, "net %d: out_dim must be in 1..32", k);
        if (n.grade < 0 || n.grade > 2) return fail(DN_ERR_INVALID_ARGUMENT, "net %d: grade must be 0 (bf16), 1 (fp32 grade) or 2 (fp16)", k);
        if (n.grade != nets[0].grade || n.arch != nets[0].arch)
            return fail(DN_ERR_INVALID_ARGUMENT, "all networks of one call must share grade and arch");
        if (((uintptr_t)n.w1 | (uintptr_t)n.w2 | (three ? (uintptr_t)n.w3 : 0) | (uintptr_t)n.wh) & 15u)
            return fail(DN_ERR_INVALID_ARGUMENT, "net %d: packed weights must be 16-byte aligned", k);
    }
    DN_HIP(hipSetDevice(device_id));
    DN_HIP(dn_launch_mlp(nets, num_nets, obs, row_mask, num_envs, obs_dim, (hipStream_t)stream));
    return DN_OK;
}

int32_t dn_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
               const uint8_t *last_dones, int64_t n_steps, int64_t n_envs, double gamma, double gae_lambda,
               float *advantages, float *returns, int32_t device_id, void *stream)
{
    if (!rewards || !values || !dones || !last_values || !last_dones || !advantages || !returns)
        return fail(DN_ERR_INVALID_ARGUMENT, "all buffers are required");
    if (n_steps < 1 || n_envs < 1) return fail(DN_ERR_INVALID_ARGUMENT, "n_steps and n_envs must be >= 1");
    DN_HIP(hipSetDevice(device_id));
    // gamma and gamma*gae_lambda are Python floats multiplied into float32 tensors (cleanRLPPO.py:245-246)
    DN_HIP(dn_launch_gae(rewards, values, dones, last_values, last_dones, n_steps, n_envs, (float)gamma,
                         (float)(gamma * gae_lambda), advantages, returns, (hipStream_t)stream));
    return DN_OK;
}

}  // extern "C"
