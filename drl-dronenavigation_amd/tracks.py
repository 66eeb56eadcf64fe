"""Waypoint tracks: the data the reference's Sol/Utilities/Waypoints.py generators return, restated.

Each generator returns a `Track(waypoints, initial_xyzs, aviary_dim, is_circle)` like the
reference's `Track(track_fn(), circle=...)` (Waypoints.py:9-20).  `targets()` applies what
PBDroneSimulator.__init__ does before handing the points to the env (PBDroneSimulator.py:127-130):
optional dilation and, for circle tracks, dropping the first point.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Track:
    waypoints: np.ndarray          # (W, 3) float64
    initial_xyzs: np.ndarray       # (1, 3) float64
    aviary_dim: np.ndarray         # x_low y_low z_low x_high y_high z_high
    is_circle: bool = False

    def __post_init__(self):
        self.waypoints = np.asarray(self.waypoints, dtype=np.float64).reshape(-1, 3)
        self.initial_xyzs = np.asarray(self.initial_xyzs, dtype=np.float64).reshape(1, 3)
        self.aviary_dim = np.asarray(self.aviary_dim, dtype=np.float64).reshape(6)

    def targets(self, target_factor=0):
        pts = dilate_targets(self.waypoints, target_factor)
        if self.is_circle:
            pts = pts[1:]
        return np.asarray(pts, dtype=np.float64)

    def __str__(self):
        return (f"Track with {len(self.waypoints)} waypoints, initial position of: {self.initial_xyzs}, "
                f"and aviary dimensions of: {self.aviary_dim}.")


def dilate_targets(targets, factor):
    """Insert `factor` evenly spaced points between consecutive targets (PBDroneSimulator.py:89-105)."""
    targets = np.asarray(targets, dtype=np.float64)
    out = []
    for a, b in zip(targets[:-1], targets[1:]):
        out.extend(np.linspace(a, b, num=factor + 2)[:-1])
    out.append(targets[-1])
    return np.asarray(out)


_BOX2 = (-2, -2, 0, 2, 2, 2)


def circle(radius=1.0, num_points=6, height=1.0, center=(0.0, 0.0, 0.0)):
    """XY circle of num_points+1 points (first == last), spawn (r, 0, cz + r) (Waypoints.py:108-139)."""
    ang = np.linspace(0, 2 * np.pi, num_points + 1, endpoint=True)
    pts = np.zeros((num_points + 1, 3))
    pts[:, 0] = center[0] + radius * np.cos(ang)
    pts[:, 1] = center[1] + radius * np.sin(ang)
    pts[:, 2] = center[2] + height
    return Track(pts, [[radius, 0, center[2] + radius]], _BOX2, True)


def reaching():
    """The 8-gate race track (Waypoints.py:172-197): gates (g + [0,0,3]) / 5, spawn on gate 0, box +-4."""
    arr = np.array([[-2.5, 4.5, 3], [10, 3.5, 1], [8, -4.5, 1], [-4.5, -6, 2], [-5, -5, 2], [5, -1, 3],
                    [2.5, 6, 3], [-2.5, 4.5, 3]], dtype=np.float64)
    for i in range(len(arr)):
        arr[i][2] += 3
        arr[i] /= 5
    return Track(arr, [arr[0]], (-4, -4, 0, 4, 4, 4), False)


def up():
    return Track([[0, 0, .1], [0, 0, .2], [0, 0, .5], [0, 0, .7], [0, 0, 1]], [[0, 0, .1]], _BOX2)


def half_up_forward():
    return Track([[0, 0, .5], [0, 0, 1], [0, 1, 1.5]], [[0, 0, .1]], _BOX2)


def up_circle():
    return Track([[0, 0, .2], [.1, 0, .3], [.1, .2, .7], [.3, .5, 1.5], [.5, 1, 1.5], [1, 1, 1.5], [1.5, 1, 1.5],
                  [1.5, 1.5, 1], [1.5, .5, 1], [1, .5, .5], [.5, .2, .2], [0, 0, .2]], [[0, 0, .1]], _BOX2)


def up_sharp_back_turn():
    return Track([[0, 0, .5], [-.5, .2, .7], [.3, .5, .7], [1, .5, 1], [1.5, 1, 1.2]], [[0, 0, .1]], _BOX2)


def default_track():
    """simulation_controller.py:97 -- Track(circle(radius=1, num_points=6, height=1), circle=True)."""
    return circle(1, 6, 1)


REGISTRY = {"circle": circle, "circle4": lambda: circle(1, 4, 1), "circle6": lambda: circle(1, 6, 1),
            "reaching": reaching, "race": reaching, "up": up, "half_up_forward": half_up_forward,
            "up_circle": up_circle, "up_sharp_back_turn": up_sharp_back_turn}
