"""Episode/metrics stream and on-disk formats around the environment.

Host-side writers for what the reference's tooling reads:
  * the rollout text dump `obs x 13, reward\\n` of PBDroneEnv.collect_rollout (Sol/Model/Environments/PBDroneEnv.py:
    811-821; read back by Sol/Model/Policies/alt_methods.py:67-140),
  * Monitor-style episode records `r, l, t` (SB3 `Monitor` wraps every env in make_env, PBDroneSimulator.py:196) as a
    `monitor.csv` [3P-recall of the SB3 format: a `#{json}` header line, then `r,l,t` rows],
  * `evaluations.npz` with `timesteps, results, ep_lengths` as SB3's EvalCallback writes it
    (PBDroneSimulator.py:718-729) [3P-recall],
  * the `found_targets` histogram behind FoundTargetsCallback (Sol/Utilities/Callbacks.py:42-75).
All of it consumes the device outputs of dn_step / dn_step_many; nothing here touches the step itself.
"""
import json
import time

import numpy as np


def format_rollout_lines(obs, rewards):
    """Text of PBDroneEnv.collect_rollout for observation rows `obs` [M, 13] and rewards [M]: every observation
    value as a 32-decimal positional float32, comma separated, then str(reward).

    Same layout as the reference's file, byte-identical when `rewards` carries the reference's own float64 values
    (tests/test_metrics.py).  Fed from the device, the reward is the step's float32 output: the reference prints
    str() of a float64 on the ordinary reward branch (17 significant digits, PBDroneEnv.py:555-571), of a float32 on
    the gate-pass branch and the literal -10.0 on a crash, so device-fed lines equal the reference's on the last two
    branches and agree to float32 precision (1e-7 relative) on the first."""
    obs = np.asarray(obs, dtype=np.float32).reshape(len(rewards), -1)
    out = []
    for row, r in zip(obs, rewards):
        cells = [np.format_float_positional(np.float32(x), unique=False, precision=32) for x in row.tolist()]
        out.append(",".join(cells) + "," + str(float(r)) + "\n")
    return "".join(out)


def write_rollout_dump(path, obs, rewards, mode="a+"):
    """Append (obs, reward) pairs to a rollout file in the reference's format.  `obs` [K, N, 13] / [M, 13] and
    `rewards` [K, N] / [M] may be torch tensors on the GPU (copied to the host here)."""
    o = obs.detach().cpu().numpy() if hasattr(obs, "detach") else np.asarray(obs)
    r = rewards.detach().cpu().numpy() if hasattr(rewards, "detach") else np.asarray(rewards)
    o = o.reshape(-1, o.shape[-1])
    r = r.reshape(-1).astype(np.float64)
    with open(path, mode) as f:
        f.write(format_rollout_lines(o, r))
    return len(r)


class EpisodeLog:
    """Monitor-style episode records gathered from step outputs (`done`, `ep_return`, `ep_length`, `found_targets`,
    `truncated`), one row per finished episode, plus the found_targets histogram."""

    def __init__(self, num_waypoints, env_id="DroneVecEnv"):
        self.t_start = time.time()
        self.env_id = env_id
        self.rows = []                                  # (r, l, t, found_targets, truncated, drone)
        self.found_hist = np.zeros(int(num_waypoints) + 1, dtype=np.int64)

    def add_step(self, done, ep_return, ep_length, found_targets, truncated=None):
        """Arrays/tensors of one vector step [N] (or step-major [K, N]); only rows where `done` is set are read."""
        def host(x):
            return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        d = host(done).astype(bool)
        if not d.any():
            return 0
        r, l, f = host(ep_return)[d], host(ep_length)[d], host(found_targets)[d]
        tr = host(truncated)[d].astype(bool) if truncated is not None else np.zeros(len(r), bool)
        who = np.nonzero(d.reshape(-1))[0] % d.shape[-1]
        t = round(time.time() - self.t_start, 6)
        for k in range(len(r)):
            self.rows.append((round(float(r[k]), 6), int(l[k]), t, int(f[k]), bool(tr[k]), int(who[k])))
        np.add.at(self.found_hist, np.clip(f, 0, len(self.found_hist) - 1), 1)
        return len(r)

    def write_monitor_csv(self, path):
        with open(path, "w") as fh:
            fh.write("#" + json.dumps({"t_start": self.t_start, "env_id": self.env_id}) + "\n")
            fh.write("r,l,t\n")
            for r, l, t, *_ in self.rows:
                fh.write(f"{r},{l},{t}\n")
        return len(self.rows)

    def summary(self):
        if not self.rows:
            return {"episodes": 0}
        r = np.array([x[0] for x in self.rows])
        l = np.array([x[1] for x in self.rows])
        return {"episodes": len(self.rows), "ep_rew_mean": float(r.mean()), "ep_len_mean": float(l.mean()),
                "found_targets_mean": float(np.mean([x[3] for x in self.rows])),
                "truncated_frac": float(np.mean([x[4] for x in self.rows])), "found_targets_hist": self.found_hist.tolist()}


def found_targets_series(found_targets, log_freq, first_call=1):
    """What FoundTargetsCallback._on_step logs (Sol/Utilities/Callbacks.py:55-61): every `log_freq`-th callback call the
    scalar infos[0]["found_targets"], i.e. drone 0's running gate count.  `found_targets` is the step-major [K, N]
    (or [K]) output of dn_step / dn_step_many; call number of row k is first_call + k.  Returns (n_calls, values)."""
    f = found_targets.detach().cpu().numpy() if hasattr(found_targets, "detach") else np.asarray(found_targets)
    f0 = f.reshape(f.shape[0], -1)[:, 0]
    calls = np.arange(first_call, first_call + len(f0))
    keep = calls % int(log_freq) == 0
    return calls[keep], f0[keep].astype(np.int64)


def save_evaluations(path, timesteps, results, ep_lengths):
    """`evaluations.npz` as SB3's EvalCallback accumulates it: timesteps [E], results [E, n_eval_episodes],
    ep_lengths [E, n_eval_episodes]."""
    np.savez(path, timesteps=np.asarray(timesteps), results=np.asarray(results, dtype=np.float64),
             ep_lengths=np.asarray(ep_lengths))
