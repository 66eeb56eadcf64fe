"""Independent float64 rigid-body integrator for the Bullet rows.  TEST HELPER.

A third statement of one `p.stepSimulation` of the reference's drone (BaseAviary.py:439-440, world set-up
:556-573), written in the WORLD frame with numpy / scipy so that it shares no algebra with the oracle
(oracle/dn_oracle.c::orc_bullet_step, body frame, hand-expanded) or with the HIP kernel (physics_phase):

    I_w   = R diag(Ixx, Iyy, Izz) R^T                           world inertia tensor
    m a   = R (0, 0, sum F_i) + m g - m v (c + c |v|)           btMultiBody linear damping 0.04 (1 + |v|)
    I_w w' = R tau_b - w x (I_w w) - I_w w (c + c |w|)          Euler's equation + angular damping + gyroscopic term
    v += a dt;  w += w' dt;  clamp every coordinate to +-100     applyDeltaVeeMultiDof
    x += v dt;  q <- rotvec(w dt) (x) q                          stepPositionsMultiDof (exponential map, world increment)

A body-frame slip shared by two restatements of the same author (a transposed rotation, a sign in r x F or in the
gyroscopic term, damping applied in the wrong frame) shows up against this form.  Everything here is [3P-recall] of
Bullet3 too: it checks consistency of the algebra, not Bullet itself (tests/test_bullet_invariants.py holds the
optional real-pybullet test).
"""
import numpy as np
from scipy.spatial.transform import Rotation

M = 0.027
J = np.array([1.4e-5, 1.4e-5, 2.17e-5])
G = 9.8
DT = 1.0 / 240.0
KF = 3.16e-10
KM = 7.94e-12
C_DAMP = 0.04
MAX_COORD_VEL = 100.0
# prop-link offsets of the URDF Bullet loads (Sol/resources/cf2x.urdf:42,54,66,78)
PROP_XY = np.array([[0.028, -0.028], [-0.028, -0.028], [-0.028, 0.028], [0.028, 0.028]])
YAW_SIGN = np.array([-1.0, 1.0, -1.0, 1.0])          # z_torque = -t0 + t1 - t2 + t3, BaseAviary.py:780


# the layout of the reference's own (dead) explicit model, BaseAviary._dynamics (BaseAviary.py:927-931): arm L / sqrt(2) and
# the y signs of Sol/resources/safegym/cf2x.urdf -- x_torque = (F0 + F1 - F2 - F3) l, y_torque = (-F0 + F1 + F2 - F3) l
L_DEAD = 0.0397 / np.sqrt(2.0)
PROP_XY_DEAD = np.array([[L_DEAD, L_DEAD], [-L_DEAD, L_DEAD], [-L_DEAD, -L_DEAD], [L_DEAD, -L_DEAD]])


def body_wrench(forces, z_torque, prop_xy=None):
    """Resultant of four +z forces at the prop offsets and the yaw torque, body frame."""
    f = np.asarray(forces, dtype=np.float64)
    force = np.array([0.0, 0.0, f.sum()])
    tau = np.zeros(3)
    for (x, y), fi in zip(PROP_XY if prop_xy is None else prop_xy, f):
        tau += np.cross([x, y, 0.0], [0.0, 0.0, fi])
    tau[2] += z_torque
    return force, tau


def step(pos, quat, vel, ang_v, forces, z_torque, extra_world_force=None, damping=C_DAMP, prop_xy=None):
    """One 1/240 s step.  quat is (x, y, z, w), base -> world.  Returns new (pos, quat, vel, ang_v).
    damping / prop_xy: Bullet's default damping and the loaded URDF's prop layout unless overridden."""
    pos, vel, w = (np.asarray(a, dtype=np.float64).copy() for a in (pos, vel, ang_v))
    rot = Rotation.from_quat(np.asarray(quat, dtype=np.float64))
    R = rot.as_matrix()
    f_b, tau_b = body_wrench(forces, z_torque, prop_xy)
    f_w = R @ f_b + np.array([0.0, 0.0, -M * G]) - M * vel * (damping + damping * np.linalg.norm(vel))
    if extra_world_force is not None:
        f_w = f_w + np.asarray(extra_world_force, dtype=np.float64)
    a = f_w / M
    I_w = R @ np.diag(J) @ R.T
    L = I_w @ w
    rhs = R @ tau_b - np.cross(w, L) - L * (damping + damping * np.linalg.norm(w))
    w_dot = np.linalg.solve(I_w, rhs)
    w = np.clip(w + w_dot * DT, -MAX_COORD_VEL, MAX_COORD_VEL)
    vel = np.clip(vel + a * DT, -MAX_COORD_VEL, MAX_COORD_VEL)
    pos = pos + vel * DT
    angle = np.linalg.norm(w) * DT
    rv = w * DT
    if angle > 0.25 * np.pi:                          # ANGULAR_MOTION_THRESHOLD (unreachable: |w| <= 100 sqrt 3)
        rv = rv * (0.25 * np.pi / angle)
    q = (Rotation.from_rotvec(rv) * rot).as_quat()
    if np.dot(q, quat) < 0.0:                         # scipy may return the antipode; keep the continuous branch
        q = -q
    return pos, q / np.linalg.norm(q), vel, w


def euler_from_quat(quat):
    """roll, pitch, yaw of p.getEulerFromQuaternion away from gimbal lock = intrinsic Z-Y-X (yaw, pitch, roll)."""
    yaw, pitch, roll = Rotation.from_quat(quat).as_euler("ZYX")
    return np.array([roll, pitch, yaw])


def thrust_to_force(thrust):
    """What the float32 action chain turns a commanded thrust (newton, normalize_actions=False) into, evaluated in
    float64 from the published formulas (PBDroneEnv.py:889, env_utils.py:29-58, BaseAviary.py:776-780)."""
    t = np.clip(np.asarray(thrust, dtype=np.float64), KF * (0.2685 * 20000 + 4070.3) ** 2, KF * (0.2685 * 65535 + 4070.3) ** 2)
    pwm = np.clip((np.sqrt(t / KF) - 4070.3) / 0.2685, 20000, 65535)
    rpm = 0.2685 * pwm + 4070.3
    return rpm ** 2 * KF, rpm ** 2 * KM
