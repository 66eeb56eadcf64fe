"""Stand-in pins for the two Bullet rows of SURVEY.md section 8(a) (A4 `p.stepSimulation`, BaseAviary.py:439-440 with the
world of :556-573; A5 `p.getEulerFromQuaternion`, :597).  CPU only; the same cases run on the HIP path in
tests/test_gpu_bullet.py.

`pybullet` is not in the image and not pinned by the reference, so the oracle's orc_bullet_step /
orc_euler_from_quat are "parity unpinned".  What CAN be checked is checked here:
  * closed-form one-step results of the published btMultiBody free-base step (hover, free fall, pure yaw, damping
    decay, Euler's equations / gyroscopic sign, the velocity clamp, the unit quaternion);
  * agreement to 1e-12 with an independent world-frame Newton-Euler integrator (tests/rigid_body_ref.py), which shares
    no algebra with the oracle's body-frame expansion;
  * Euler angles against scipy's intrinsic Z-Y-X decomposition and the two gimbal-lock branches by construction;
  * an optional one-step transition test against a real pybullet (skips here; settles the recall where the wheel exists).
"""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import rigid_body_ref as RB
from oracle import oracle as O

DP = C.POINTER(C.c_double)


def _dp(a):
    return a.ctypes.data_as(DP)


def bullet_step(pos, quat, vel, ang_v, forces, z_torque=0.0):
    p, q, v, w = (np.array(a, dtype=np.float64) for a in (pos, quat, vel, ang_v))
    f = np.array(forces, dtype=np.float64)
    O.lib().orc_bullet_step(_dp(p), _dp(q), _dp(v), _dp(w), _dp(f), float(z_torque))
    return p, q, v, w


def euler(quat):
    q = np.array(quat, dtype=np.float64)
    rpy = np.zeros(3)
    O.lib().orc_euler_from_quat(_dp(q), _dp(rpy))
    return rpy


REST = dict(pos=[0.3, -0.2, 1.0], quat=[0.0, 0.0, 0.0, 1.0], vel=[0.0, 0.0, 0.0], ang_v=[0.0, 0.0, 0.0])
HOVER_F = RB.M * RB.G / 4.0          # 0.06615 N per rotor


def test_bullet_hover_equilibrium():
    """F_i = M G / 4 from rest: zero acceleration, the state does not move."""
    assert abs(HOVER_F - 0.06615) < 1e-15
    p, q, v, w = bullet_step(**REST, forces=[HOVER_F] * 4)
    np.testing.assert_allclose(v, 0.0, atol=1e-15)
    np.testing.assert_allclose(w, 0.0, atol=1e-15)
    np.testing.assert_allclose(p, REST["pos"], atol=1e-16)
    np.testing.assert_array_equal(q, [0.0, 0.0, 0.0, 1.0])


def test_bullet_free_fall_one_step():
    """No thrust from rest: v_z = -9.8/240 and, semi-implicit Euler, z -= 9.8/240^2."""
    p, q, v, w = bullet_step(**REST, forces=[0.0] * 4)
    np.testing.assert_allclose(v, [0.0, 0.0, -9.8 / 240.0], rtol=1e-15, atol=1e-18)
    np.testing.assert_allclose(p, [0.3, -0.2, 1.0 - 9.8 / 240.0 ** 2], rtol=1e-15)
    np.testing.assert_array_equal(w, 0.0)


def test_bullet_pure_yaw_from_alternating_thrust():
    """(F, F', F, F') with F + F' = M G / 2: no net force, no roll / pitch torque, yaw rate = z_torque / Izz dt."""
    F, Fp = HOVER_F * 1.2, HOVER_F * 0.8
    tq = np.array([F, Fp, F, Fp]) * (RB.KM / RB.KF)
    zt = -tq[0] + tq[1] - tq[2] + tq[3]
    p, q, v, w = bullet_step(**REST, forces=[F, Fp, F, Fp], z_torque=zt)
    np.testing.assert_allclose(v, 0.0, atol=1e-15)
    np.testing.assert_allclose(w[:2], 0.0, atol=1e-12)
    assert zt < 0
    np.testing.assert_allclose(w[2], zt / 2.17e-5 / 240.0, rtol=1e-14)
    # the attitude turned about +z by w_z dt (exponential map): q = (0, 0, sin(a/2), cos(a/2))
    a = w[2] / 240.0
    np.testing.assert_allclose(q, [0.0, 0.0, np.sin(a / 2), np.cos(a / 2)], atol=1e-15)


@pytest.mark.parametrize("v0", [[1.5, 0.0, 0.0], [0.3, -0.4, 0.0], [0.0, 2.0, 0.0]])
def test_bullet_linear_damping_decay(v0):
    """Thrust = weight, level: v <- v (1 - (c + c |v|) dt), c = 0.04 (btMultiBody default, not removed: BaseAviary.py:571-573)."""
    v0 = np.array(v0)
    p, q, v, w = bullet_step(REST["pos"], REST["quat"], v0, [0, 0, 0], forces=[HOVER_F] * 4)
    k = 0.04 + 0.04 * np.linalg.norm(v0)
    np.testing.assert_allclose(v, v0 * (1.0 - k / 240.0), rtol=1e-14, atol=1e-17)
    np.testing.assert_allclose(p, np.array(REST["pos"]) + v / 240.0, rtol=1e-15)


def test_bullet_angular_damping_decay_about_principal_axis():
    """Spin about body z (a principal axis: no gyroscopic torque): w <- w (1 - (c + c |w|) dt)."""
    w0 = np.array([0.0, 0.0, 3.0])
    p, q, v, w = bullet_step(REST["pos"], REST["quat"], [0, 0, 0], w0, forces=[HOVER_F] * 4)
    np.testing.assert_allclose(w, w0 * (1.0 - (0.04 + 0.04 * 3.0) / 240.0), rtol=1e-14, atol=1e-17)


def test_bullet_gyroscopic_term_sign_and_size():
    """Euler's equations for Ixx = Iyy < Izz: Ixx w_x' = (Iyy - Izz) w_y w_z, Iyy w_y' = (Izz - Ixx) w_z w_x, w_z' = 0
    (plus the damping), body frame = world frame at identity attitude."""
    w0 = np.array([0.0, 4.0, 5.0])
    p, q, v, w = bullet_step(REST["pos"], REST["quat"], [0, 0, 0], w0, forces=[HOVER_F] * 4)
    k = 0.04 + 0.04 * np.linalg.norm(w0)
    Ix, Iy, Iz = RB.J
    exp = w0 + np.array([(Iy - Iz) * w0[1] * w0[2] / Ix, (Iz - Ix) * w0[2] * w0[0] / Iy, 0.0]) / 240.0 - w0 * k / 240.0
    np.testing.assert_allclose(w, exp, rtol=1e-13, atol=1e-15)
    assert w[0] < 0.0                      # the sign a wrong cross-product order would flip


def test_bullet_angular_momentum_direction_is_kept_by_a_tumbling_body():
    """Torque-free tumbling: the damping torque is -k L (body frame, elementwise I w), so the world angular momentum
    L = R I R^T w only shrinks; its direction drifts at the integrator's O(dt^2) per step."""
    rng = np.random.default_rng(5)
    q = Rotation.random(random_state=3).as_quat()
    p, v, w = np.array([0.0, 0.0, 50.0]), np.zeros(3), np.array([6.0, -9.0, 4.0])
    R0 = Rotation.from_quat(q).as_matrix()
    L0 = R0 @ np.diag(RB.J) @ R0.T @ w
    worst = 0.0
    for _ in range(240):
        p, q, v, w = bullet_step(p, q, v, w, forces=[HOVER_F] * 4)
        R = Rotation.from_quat(q).as_matrix()
        L = R @ np.diag(RB.J) @ R.T @ w
        cosang = L @ L0 / np.linalg.norm(L) / np.linalg.norm(L0)
        worst = max(worst, np.arccos(np.clip(cosang, -1, 1)))
        assert abs(np.linalg.norm(q) - 1.0) < 1e-15
    assert np.linalg.norm(L) < np.linalg.norm(L0)            # damped
    assert worst < 0.08, worst                                # one second of tumbling at ~11 rad/s, explicit Euler
    del rng


def test_bullet_velocity_clamp_fires():
    """applyDeltaVeeMultiDof clamps every velocity coordinate at m_maxCoordinateVelocity = 100."""
    p, q, v, w = bullet_step(REST["pos"], REST["quat"], [150.0, -170.0, 30.0], [-300.0, 20.0, 120.0], forces=[HOVER_F] * 4)
    assert v[0] == 100.0 and v[1] == -100.0 and abs(v[2]) < 100.0
    assert w[0] == -100.0 and abs(w[1]) < 100.0 and w[2] == 100.0
    np.testing.assert_allclose(p, np.array(REST["pos"]) + v / 240.0, rtol=1e-15)


def test_bullet_angle_clamp_is_unreachable_at_240hz():
    """ANGULAR_MOTION_THRESHOLD = pi/4 per step needs |w| > 188.5 rad/s, but the coordinate clamp above caps |w| at
    100 sqrt(3) = 173.2: the clamp branch of stepPositionsMultiDof never runs at dt = 1/240.  At the cap the attitude
    turns by exactly |w| dt."""
    assert 100.0 * np.sqrt(3.0) / 240.0 < np.pi / 4
    w0 = np.array([400.0, 400.0, 0.0])                        # Ixx = Iyy and w_z = 0: no gyroscopic torque
    p, q, v, w = bullet_step(REST["pos"], REST["quat"], [0, 0, 0], w0, forces=[HOVER_F] * 4)
    np.testing.assert_array_equal(w, [100.0, 100.0, 0.0])
    ang = 2.0 * np.arccos(np.clip(q[3], -1, 1))
    np.testing.assert_allclose(ang, 100.0 * np.sqrt(2.0) / 240.0, rtol=1e-12)
    np.testing.assert_allclose(q[:3] / np.linalg.norm(q[:3]), np.array([1.0, 1.0, 0.0]) / np.sqrt(2.0), rtol=1e-13, atol=1e-16)


def test_bullet_small_rate_taylor_branch_is_continuous():
    """Below |w| = 1e-3 Bullet uses a Taylor series of sin(|w| dt / 2) / |w|; the two branches meet to 1e-20."""
    for mag in (0.99e-3, 1.01e-3):
        w0 = np.array([0.6, 0.0, 0.8]) * mag
        # cancel the damping / gyro so that w stays w0: irrelevant here, only the map of w_new matters
        p, q, v, w = bullet_step(REST["pos"], REST["quat"], [0, 0, 0], w0, forces=[HOVER_F] * 4)
        h = np.linalg.norm(w) / 480.0
        np.testing.assert_allclose(q, np.append(w / np.linalg.norm(w) * np.sin(h), np.cos(h)), rtol=0, atol=1e-19)


def random_states(rng, n):
    quat = Rotation.random(n, random_state=int(rng.integers(1 << 30))).as_quat()
    pos = rng.uniform(-2, 2, (n, 3)) + [0, 0, 3]
    vel = rng.normal(0, 2.0, (n, 3))
    ang_v = rng.normal(0, 15.0, (n, 3))
    forces = rng.uniform(0.028, 0.148, (n, 4))
    zt = ((forces * (RB.KM / RB.KF)) * RB.YAW_SIGN).sum(1)
    return pos, quat, vel, ang_v, forces, zt


def test_bullet_step_matches_independent_world_frame_integrator():
    """orc_bullet_step (body frame, hand-expanded) against tests/rigid_body_ref.py (world frame, numpy / scipy) on
    random tumbling states over the whole thrust range: 1e-12."""
    rng = np.random.default_rng(11)
    pos, quat, vel, ang_v, forces, zt = random_states(rng, 2000)
    pos[:5], vel[:5], ang_v[:5] = REST["pos"], 0.0, 0.0              # a few at rest, a few past the clamp
    vel[5:8] *= 60.0
    ang_v[8:11] *= 10.0
    for k in range(len(pos)):
        got = bullet_step(pos[k], quat[k], vel[k], ang_v[k], forces[k], zt[k])
        ref = RB.step(pos[k], quat[k], vel[k], ang_v[k], forces[k], zt[k])
        for g, r, name in zip(got, ref, ("pos", "quat", "vel", "ang_v")):
            np.testing.assert_allclose(g, r, rtol=1e-12, atol=1e-12, err_msg=f"state {k}: {name}")


def test_bullet_multi_step_trajectory_matches_independent_integrator():
    """240 free-running steps (one second) of a kicked drone: the two integrators stay within 1e-9."""
    rng = np.random.default_rng(2)
    s1 = (np.array([0.0, 0.0, 5.0]), np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3), np.array([0.5, -0.3, 0.2]))
    s2 = tuple(a.copy() for a in s1)
    for _ in range(240):
        f = HOVER_F * (1.0 + 0.2 * rng.standard_normal(4))
        zt = ((f * (RB.KM / RB.KF)) * RB.YAW_SIGN).sum()
        s1 = bullet_step(*s1, f, zt)
        s2 = RB.step(*s2, f, zt)
    for a, b in zip(s1, s2):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)


def test_independent_integrator_reproduces_the_references_own_explicit_dynamics(golden):
    """The one statement of a rigid-body step the reference itself owns: BaseAviary._dynamics + _integrateQ
    (BaseAviary.py:899-973; dead code -- Physics.DYN is never selected -- executed for the fixture dead_dynamics.npz with
    TIMESTEP set).  With its three declared differences from what Bullet simulates switched in (no damping, arm
    L / sqrt(2), the safegym URDF's prop y signs) the independent world-frame integrator reproduces it to 1e-13:
    semi-implicit order (v, w first, then x with the NEW v), thrust along body z, yaw torque -t0 + t1 - t2 + t3, the
    gyroscopic term's sign, the exponential-map attitude update.  Chain of evidence for row A4's structure:
    reference _dynamics == rigid_body_ref(damping 0, dead layout) here; rigid_body_ref(Bullet damping, loaded URDF layout)
    == oracle to 1e-12 (above) == HIP path to 2e-6 (tests/test_gpu_bullet.py).  What stays unpinned is Bullet-specific: the
    damping law and the velocity clamp."""
    g = golden("dead_dynamics")
    assert abs(float(g["L"]) / np.sqrt(2.0) - RB.L_DEAD) < 1e-18 and float(g["KF"]) == RB.KF and float(g["KM"]) == RB.KM
    assert (np.abs(g["rates"]).sum(1) == 0).sum() >= 3                  # _integrateQ's |omega| = 0 branch is in the fixture
    for k in range(len(g["pos"])):
        q = g["quat"][k]
        R = Rotation.from_quat(q).as_matrix()
        f, tq = g["rpm"][k] ** 2 * RB.KF, g["rpm"][k] ** 2 * RB.KM
        zt = -tq[0] + tq[1] - tq[2] + tq[3]
        pos, quat, vel, w = RB.step(g["pos"][k], q, g["vel"][k], R @ g["rates"][k], f, zt, damping=0.0, prop_xy=RB.PROP_XY_DEAD)
        np.testing.assert_allclose(pos, g["out_pos"][k], rtol=0, atol=1e-13)
        np.testing.assert_allclose(vel, g["out_vel"][k], rtol=0, atol=1e-13)
        qq = g["out_quat"][k] / np.linalg.norm(g["out_quat"][k])
        np.testing.assert_allclose(quat, qq if np.dot(qq, quat) > 0 else -qq, rtol=0, atol=1e-13)
        np.testing.assert_allclose(R.T @ w, g["out_rates"][k], rtol=0, atol=1e-12)      # new body rates: w' = R w_b' to first order
        np.testing.assert_allclose(R @ g["out_rates"][k], g["out_ang_v_world"][k], rtol=0, atol=1e-12)


def test_oracle_without_damping_matches_the_references_explicit_dynamics_through_its_action_chain(golden):
    """A DIRECT reference-generated target for the rigid-body step: thrust commands (a, b, a, b) through the reference's own
    PBDroneEnv._preprocessAction (float32 rpm) into its own BaseAviary._dynamics (dead_dynamics.npz, chain_* arrays).  That
    pattern has no roll / pitch torque in either prop layout, so the reference's explicit model and the body Bullet
    simulates coincide once Bullet's damping is switched off -- which is the changeDynamics line the reference keeps
    commented out (BaseAviary.py:571-573) and dn_config.zero_damping here.  The oracle step with damp = 0 against it:
    tumbling states, gyroscopic coupling, yaw torque, semi-implicit order, attitude update.  (1e-7: _dynamics sums the
    four float32 rotor forces in float32, Bullet in double.)"""
    g = golden("dead_dynamics")
    L = O.lib()
    n = len(g["chain_pos"])
    assert n >= 200 and np.abs(g["chain_ang_v"]).max() > 10.0
    for k in range(n):
        rpm = g["chain_rpm"][k]
        assert rpm.dtype == np.float32
        f = np.zeros(4, np.float32)
        z = C.c_float()
        L.orc_rotor_forces(rpm.ctypes.data_as(C.POINTER(C.c_float)), f.ctypes.data_as(C.POINTER(C.c_float)), C.byref(z))
        p, q, v, w = (np.array(g["chain_" + name][k], dtype=np.float64) for name in ("pos", "quat", "vel", "ang_v"))
        fd, none = f.astype(np.float64), np.zeros(3)
        L.orc_bullet_step_damp(_dp(p), _dp(q), _dp(v), _dp(w), _dp(fd), float(z.value), _dp(none), 0.0)
        qq = g["chain_out_quat"][k] / np.linalg.norm(g["chain_out_quat"][k])
        np.testing.assert_allclose(p, g["chain_out_pos"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(v, g["chain_out_vel"][k], rtol=0, atol=1e-7)
        np.testing.assert_allclose(q, qq if np.dot(qq, q) > 0 else -qq, rtol=0, atol=1e-8)
        np.testing.assert_allclose(w, g["chain_out_ang_v_world"][k], rtol=0, atol=1e-6)


def test_euler_matches_scipy_zyx_away_from_gimbal_lock():
    """p.getEulerFromQuaternion = intrinsic yaw-pitch-roll (Z-Y-X); scipy is the independent statement."""
    quats = Rotation.random(5000, random_state=7).as_quat()
    n = 0
    for q in quats:
        sarg = -2.0 * (q[0] * q[2] - q[3] * q[1])
        if abs(sarg) >= 0.9999:
            continue
        np.testing.assert_allclose(euler(q), RB.euler_from_quat(q), rtol=0, atol=1e-10)
        n += 1
    assert n > 4900


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_euler_gimbal_lock_branches(sign):
    """|sarg| >= 0.99999: roll := 0, pitch := +-pi/2 exactly, yaw := 2 atan2(-+x, +-y) carries roll + yaw.  The rotation
    rebuilt from the returned angles is the input rotation to the branch's own resolution (sqrt(2e-5) rad)."""
    hit = 0
    for yaw in np.linspace(-3.0, 3.0, 25):
        for eps in (0.0, 1e-4, 2e-3):
            rot = Rotation.from_euler("ZYX", [yaw, sign * (np.pi / 2 - eps), 0.3])
            q = rot.as_quat()
            if q[3] < 0:
                q = -q
            sarg = -2.0 * (q[0] * q[2] - q[3] * q[1])
            rpy = euler(q)
            if abs(sarg) >= 0.99999:
                hit += 1
                assert rpy[0] == 0.0 and rpy[1] == sign * 0.5 * np.pi
                back = Rotation.from_euler("ZYX", [rpy[2], rpy[1], rpy[0]])
                assert (back.inv() * rot).magnitude() < 5e-3
            else:
                np.testing.assert_allclose(rpy, RB.euler_from_quat(q), atol=1e-9)
    assert hit >= 25


def test_oracle_env_hover_holds_position_for_a_second():
    """The env-level chain (float32 action -> thrust -> PWM -> RPM -> force, A1-A3) feeding the integrator: commanding
    the hover thrust keeps the drone within float32 rounding of the rotor force of where it spawned."""
    wp = np.array([[0.0, 0.0, 1.5]])
    cfg = O.make_config(wp, [0.0, 0.0, 1.0], [-5, -5, 0, 5, 5, 5], cylinder=False, ground_contact=False,
                        normalize_actions=False, max_steps=10000)
    env = O.OracleVecEnv(cfg, 1)
    env.reset()
    a = np.full((1, 4), HOVER_F, np.float32)
    for _ in range(240):
        assert not env.step(a)["done"][0]
    f32, _ = RB.thrust_to_force(np.float32(HOVER_F))
    acc = 4 * f32 / RB.M - RB.G                               # residual of the float32 chain: ~1e-7 m/s^2
    assert abs(acc) < 1e-5
    np.testing.assert_allclose(env.envs["pos"][0], [0.0, 0.0, 1.0 + 0.5 * acc], atol=1e-6)
    np.testing.assert_allclose(env.envs["quat"][0], [0, 0, 0, 1], atol=1e-12)


URDF = """<?xml version="1.0" ?>
<robot name="cf2">
  <link name="base_link">
    <inertial><origin rpy="0 0 0" xyz="0 0 0"/><mass value="0.027"/>
      <inertia ixx="1.4e-5" ixy="0.0" ixz="0.0" iyy="1.4e-5" iyz="0.0" izz="2.17e-5"/></inertial>
    <collision><origin rpy="0 0 0" xyz="0 0 0"/><geometry><cylinder radius=".06" length=".025"/></geometry></collision>
  </link>
  {props}
  <link name="center_of_mass_link"><inertial><origin rpy="0 0 0" xyz="0 0 0"/><mass value="0"/>
      <inertia ixx="0" ixy="0" ixz="0" iyy="0" iyz="0" izz="0"/></inertial></link>
  <joint name="center_of_mass_joint" type="fixed"><parent link="base_link"/><child link="center_of_mass_link"/></joint>
</robot>
"""
PROP = """<link name="prop{i}_link"><inertial><origin rpy="0 0 0" xyz="{x} {y} 0"/><mass value="0"/>
      <inertia ixx="0" ixy="0" ixz="0" iyy="0" iyz="0" izz="0"/></inertial></link>
  <joint name="prop{i}_joint" type="fixed"><parent link="base_link"/><child link="prop{i}_link"/></joint>"""


def test_bullet_one_step_transitions_against_real_pybullet():
    """Settles the [3P-recall] where a pybullet wheel exists (not in this image: skips).  The body is rebuilt from the
    constants of Sol/resources/cf2x.urdf (mass, inertia, massless fixed prop links at +-0.028) and stepped the way
    BaseAviary._housekeeping / _physics drive it (BaseAviary.py:556-573, :776-794)."""
    p = pytest.importorskip("pybullet")
    rng = np.random.default_rng(3)
    props = "\n  ".join(PROP.format(i=i, x=x, y=y) for i, (x, y) in enumerate(RB.PROP_XY))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cf2x.urdf")
        with open(path, "w") as fh:
            fh.write(URDF.format(props=props))
        cid = p.connect(p.DIRECT)
        try:
            p.setGravity(0, 0, -9.8, physicsClientId=cid)
            p.setRealTimeSimulation(0, physicsClientId=cid)
            p.setTimeStep(1.0 / 240.0, physicsClientId=cid)
            body = p.loadURDF(path, [0, 0, 1], [0, 0, 0, 1], flags=p.URDF_USE_INERTIA_FROM_FILE, physicsClientId=cid)
            pos, quat, vel, ang_v, forces, zt = random_states(rng, 200)
            for k in range(len(pos)):
                p.resetBasePositionAndOrientation(body, pos[k], quat[k], physicsClientId=cid)
                p.resetBaseVelocity(body, vel[k], ang_v[k], physicsClientId=cid)
                for i in range(4):
                    p.applyExternalForce(body, i, forceObj=[0, 0, forces[k, i]], posObj=[0, 0, 0], flags=p.LINK_FRAME,
                                         physicsClientId=cid)
                p.applyExternalTorque(body, 4, torqueObj=[0, 0, zt[k]], flags=p.LINK_FRAME, physicsClientId=cid)
                p.stepSimulation(physicsClientId=cid)
                got_p, got_q = p.getBasePositionAndOrientation(body, physicsClientId=cid)
                got_v, got_w = p.getBaseVelocity(body, physicsClientId=cid)
                ref = bullet_step(pos[k], quat[k], vel[k], ang_v[k], forces[k], zt[k])
                for g, r, name in zip((got_p, got_q, got_v, got_w), ref, ("pos", "quat", "vel", "ang_v")):
                    np.testing.assert_allclose(g, r, rtol=1e-9, atol=1e-9, err_msg=f"pybullet vs oracle, state {k}: {name}")
                np.testing.assert_allclose(p.getEulerFromQuaternion(got_q), euler(np.array(got_q)), atol=1e-12)
        finally:
            p.disconnect(cid)
