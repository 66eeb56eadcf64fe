"""Stand-in modules that let the reference's Python env logic be IMPORTED in the build container.

Used only by tests/golden/gen_golden.py (fixture generation, this container only).  The
reference needs pybullet, gymnasium, gym, stable_baselines3, pkg_resources, torchviz, graphviz,
hiddenlayer -- none installed, none installable (no network).  These stubs provide just enough
surface for `Sol.Model.Environments.PBDroneEnv`, `Sol.Model.Environments.normalize` and
`Sol.Utilities.Waypoints` to import and run unmodified.

The fake `pybullet` keeps one rigid body per client.  `stepSimulation` advances it with the
oracle's restatement of Bullet (oracle.orc_bullet_step) -- so closed-loop fixtures pin every
line of the reference's own Python (rows A1-A3, A6-A12) while the integrator itself (row A4)
stays "parity unpinned": Bullet is not in the tree.
"""
import ctypes as C
import sys
import types

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()


class _Env:
    metadata = {}

    def reset(self, seed=None, options=None):
        return None


class _Wrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)


class FakeBullet(types.ModuleType):
    """Records the C-API calls the reference makes and integrates with the oracle."""
    GUI, DIRECT, LINK_FRAME, URDF_USE_INERTIA_FROM_FILE = 1, 2, 1, 2
    COV_ENABLE_RGB_BUFFER_PREVIEW = COV_ENABLE_DEPTH_BUFFER_PREVIEW = COV_ENABLE_SEGMENTATION_MARK_PREVIEW = 0

    def __init__(self, oracle_lib):
        super().__init__("pybullet")
        self._L = oracle_lib
        self._clients = {}
        self._next = 0
        self.calls = []
        self.current = None

    # -- world management ---------------------------------------------------
    def connect(self, mode, **kw):
        cid = self._next
        self._next += 1
        self._clients[cid] = dict(pos=np.zeros(3), quat=np.array([0, 0, 0, 1.0]), vel=np.zeros(3), ang_v=np.zeros(3),
                                  forces=np.zeros(4), z_torque=0.0, nbodies=0, gravity=None, dt=None,
                                  applied=[], contacts=())
        self.current = cid
        return cid

    def _c(self, kw):
        cid = kw.get("physicsClientId", self.current)
        self.current = cid
        return self._clients[cid]

    def resetSimulation(self, **kw):
        c = self._c(kw)
        c["nbodies"] = 0
        c["forces"][:] = 0
        c["z_torque"] = 0.0

    def setGravity(self, x, y, z, **kw):
        self._c(kw)["gravity"] = (x, y, z)

    def setTimeStep(self, dt, **kw):
        self._c(kw)["dt"] = dt

    def setRealTimeSimulation(self, *a, **kw):
        pass

    def setAdditionalSearchPath(self, *a, **kw):
        pass

    def loadURDF(self, fileName, basePosition=None, baseOrientation=None, **kw):
        c = self._c(kw)
        bid = c["nbodies"]
        c["nbodies"] += 1
        if "plane" not in str(fileName) and "target" not in str(fileName):
            c["urdf"] = str(fileName)
            c["flags"] = kw.get("flags")
            c["pos"] = np.array(basePosition, dtype=np.float64).copy()
            c["quat"] = np.array(baseOrientation, dtype=np.float64).copy()
            c["vel"] = np.zeros(3)
            c["ang_v"] = np.zeros(3)
        return bid

    def getQuaternionFromEuler(self, rpy, **kw):
        r, p, y = [float(v) for v in rpy]
        cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
        return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                cr * cp * cy + sr * sp * sy)

    # -- the five force calls + the step ----------------------------------------
    def applyExternalForce(self, objectUniqueId, linkIndex, forceObj, posObj, flags, **kw):
        c = self._c(kw)
        assert flags == self.LINK_FRAME and list(posObj) == [0, 0, 0]
        if linkIndex == 4:                                   # BaseAviary._drag: a general vector on the centre-of-mass link
            c["body_force"] = np.array([float(v) for v in forceObj])
            c["applied"].append(("D", linkIndex, type(forceObj[2]).__name__, tuple(float(v) for v in forceObj)))
            return
        assert forceObj[0] == 0 and forceObj[1] == 0
        c["forces"][linkIndex] += float(forceObj[2])         # PyFloat_AsDouble widening; forces on one link add up
        c["applied"].append(("F", linkIndex, type(forceObj[2]).__name__, float(forceObj[2])))

    def applyExternalTorque(self, objectUniqueId, linkIndex, torqueObj, flags, **kw):
        c = self._c(kw)
        assert flags == self.LINK_FRAME and linkIndex == 4 and torqueObj[0] == 0 and torqueObj[1] == 0
        c["z_torque"] = float(torqueObj[2])
        c["applied"].append(("T", linkIndex, type(torqueObj[2]).__name__, float(torqueObj[2])))

    def stepSimulation(self, **kw):
        c = self._c(kw)
        assert c["gravity"] == (0, 0, -9.8) and abs(c["dt"] - 1 / 240) < 1e-18
        if c.get("frozen"):
            c["forces"][:] = 0
            c["z_torque"] = 0.0
            return
        dp = C.POINTER(C.c_double)
        f = np.ascontiguousarray(c["forces"], dtype=np.float64)
        self._L.orc_bullet_step(c["pos"].ctypes.data_as(dp), c["quat"].ctypes.data_as(dp),
                                c["vel"].ctypes.data_as(dp), c["ang_v"].ctypes.data_as(dp),
                                f.ctypes.data_as(dp), C.c_double(c["z_torque"]))
        c["forces"][:] = 0                                   # Bullet clears external forces after a step
        c["z_torque"] = 0.0

    # -- getters --------------------------------------------------------------
    def getBasePositionAndOrientation(self, bid, **kw):
        c = self._c(kw)
        return tuple(c["pos"].tolist()), tuple(c["quat"].tolist())

    def getBaseVelocity(self, bid, **kw):
        c = self._c(kw)
        return tuple(c["vel"].tolist()), tuple(c["ang_v"].tolist())

    def getEulerFromQuaternion(self, quat, **kw):
        q = np.array(quat, dtype=np.float64)
        out = np.zeros(3)
        dp = C.POINTER(C.c_double)
        self._L.orc_euler_from_quat(q.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return tuple(out.tolist())

    # -- used only by the dead Physics.PYB_GND / PYB_DRAG terms (fixture extra_physics.npz) [3P-recall] --
    @staticmethod
    def _matrix(q):
        x, y, z, w = [float(v) for v in q]
        s = 2.0 / (x * x + y * y + z * z + w * w)           # btMatrix3x3::setRotation
        xs, ys, zs = x * s, y * s, z * s
        wx, wy, wz = w * xs, w * ys, w * zs
        xx, xy, xz, yy, yz, zz = x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
        return np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx],
                         [xz - wy, yz + wx, 1.0 - (xx + yy)]])

    def getMatrixFromQuaternion(self, quat, **kw):
        return tuple(self._matrix(quat).ravel().tolist())

    def getLinkStates(self, bid, linkIndices, **kw):
        c = self._c(kw)
        R = self._matrix(c["quat"])
        offs = {0: (0.028, -0.028, 0.0), 1: (-0.028, -0.028, 0.0), 2: (-0.028, 0.028, 0.0), 3: (0.028, 0.028, 0.0),
                4: (0.0, 0.0, 0.0)}                           # Sol/resources/cf2x.urdf:42,54,66,78,90
        return [(tuple((c["pos"] + R @ np.array(offs[i])).tolist()),) for i in linkIndices]

    def getContactPoints(self, *a, **kw):
        return self._clients[self.current]["contacts"]

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _noop(*a, **kw):
            self.calls.append(name)
            return 0
        return _noop


def install(oracle_lib):
    """Insert the stub modules into sys.modules; returns the fake pybullet module."""
    import torch  # noqa: F401  (real; PBDroneEnv imports it)
    import pandas  # noqa: F401

    pb = FakeBullet(oracle_lib)
    sys.modules["pybullet"] = pb

    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: "/nonexistent"
    sys.modules["pybullet_data"] = pbd

    gymn = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    space_mod = types.ModuleType("gymnasium.spaces.space")
    space_mod.Space = object
    spaces.space = space_mod
    spaces.Space = object
    gymn.spaces = spaces
    gymn.Env = _Env
    sys.modules["gymnasium"] = gymn
    sys.modules["gymnasium.spaces"] = spaces
    sys.modules["gymnasium.spaces.space"] = space_mod

    gym = types.ModuleType("gym")
    core = types.ModuleType("gym.core")
    core.Wrapper = _Wrapper
    gym.core = core
    gym.Env = _Env
    gym.Wrapper = _Wrapper
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.core"] = core
    sys.modules["gym.spaces"] = spaces

    pkgr = types.ModuleType("pkg_resources")

    def resource_filename(package, rel):
        # BaseControl._getURDFParameter asks the (absent) gym_pybullet_drones package for assets/cf2x.urdf; the reference
        # vendors that file as Sol/resources/cf2x.urdf (cwd is the reference root while fixtures are generated)
        import os
        return os.path.join("Sol", "resources", os.path.basename(rel))
    pkgr.resource_filename = resource_filename
    sys.modules["pkg_resources"] = pkgr

    sb3 = types.ModuleType("stable_baselines3")
    sb3c = types.ModuleType("stable_baselines3.common")
    sb3r = types.ModuleType("stable_baselines3.common.running_mean_std")
    sb3r.RunningMeanStd = object
    sb3.common = sb3c
    sb3c.running_mean_std = sb3r
    sys.modules["stable_baselines3"] = sb3
    sys.modules["stable_baselines3.common"] = sb3c
    sys.modules["stable_baselines3.common.running_mean_std"] = sb3r

    for name in ("torchviz", "graphviz", "hiddenlayer"):
        m = types.ModuleType(name)
        m.make_dot = lambda *a, **k: None
        sys.modules[name] = m
    import matplotlib
    matplotlib.use("Agg")
    return pb
