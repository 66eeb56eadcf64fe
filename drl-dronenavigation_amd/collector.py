"""Rollout collection over sharded drones: the host loop around the step kernel for BASELINE configs 3-5.

What it stands in for in the reference: SB3 `OnPolicyAlgorithm.collect_rollouts` driving the
`SubprocVecEnv` built at Sol/Model/PBDroneSimulator.py:653-666 (policy -> clip -> env.step -> buffer.add,
then `compute_returns_and_advantage`), whose only in-tree statement of the advantage recursion is
Sol/Model/Algorithms/cleanRLPPO.py:207-248.  Here every buffer stays on the GPU: the policy's output tensor is
handed to `dn_step` by pointer, the rollout is laid out step-major [n_steps, N_local, ...] and the advantages
come from the `dn_gae` kernel.

Sharding: every drone is an independent world, so rank r of R owns the contiguous range
[r*N_local, (r+1)*N_local) and stepping needs NO collective.  The one exchange is per rollout: where the
learner draws its minibatches from the global batch (BASELINE config 4) the packed [2, n_steps, N_local]
advantages/returns are all-gathered (RCCL `ncclAllGather` over xGMI when the process group's backend is
"nccl"; the same call runs on gloo for the CPU tests).

`ShardPlan` and `all_gather_rollout` are device-agnostic (plain torch.distributed); `RolloutCollector` needs
the HIP environment and has no CPU path.
"""
from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous split of `global_num_envs` drones over `world_size` ranks."""
    global_num_envs: int
    world_size: int
    rank: int

    def __post_init__(self):
        if self.world_size < 1 or not 0 <= self.rank < self.world_size:
            raise ValueError(f"bad rank/world_size {self.rank}/{self.world_size}")
        if self.global_num_envs < self.world_size or self.global_num_envs % self.world_size:
            raise ValueError(f"global_num_envs ({self.global_num_envs}) must be a positive multiple of the world size "
                             f"({self.world_size}): all_gather_into_tensor needs equal shards")

    @property
    def num_envs(self):
        """Drones owned by this rank."""
        return self.global_num_envs // self.world_size

    @property
    def env_id_offset(self):
        """Global id of this rank's drone 0 (keys the Philox noise streams; dn_config.env_id_offset)."""
        return self.rank * self.num_envs

    def local_slice(self):
        return slice(self.env_id_offset, self.env_id_offset + self.num_envs)

    @classmethod
    def from_env(cls, global_num_envs):
        import os
        return cls(int(global_num_envs), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))


def all_gather_rollout(advantages, returns, group=None):
    """All-gather the per-rank [n_steps, N_local] advantages and returns into [n_steps, N_global] tensors whose
    columns are in global drone order.  One collective for both arrays (packed [2, T, N_local] send buffer ->
    [R*2, T, N_local] receive buffer), none at all without a process group (a one-rank group still goes through the
    collective, which keeps that code path testable on one GPU).  Convenience form: it allocates the receive buffer and a
    permuted copy on every call; a collector that gathers every rollout uses RolloutGather below, which does neither."""
    import torch.distributed as dist
    if advantages.shape != returns.shape or advantages.dim() != 2:
        raise ValueError("advantages and returns must both be [n_steps, N_local]")
    if not (dist.is_available() and dist.is_initialized()):
        return advantages, returns
    world = dist.get_world_size(group)
    T, n = advantages.shape
    send = torch.stack((advantages, returns)).contiguous()                       # [2, T, n]
    recv = torch.empty((world * 2, T, n), dtype=send.dtype, device=send.device)     # concatenated along dim 0
    dist.all_gather_into_tensor(recv, send, group=group)
    out = recv.view(world, 2, T, n).permute(1, 2, 0, 3).reshape(2, T, world * n)   # rank-major columns = global order
    return out[0], out[1]


class RolloutGather:
    """The per-rollout exchange of BASELINE config 4 with every buffer allocated ONCE: `send` [2, T, N_local] is the storage the
    collector's advantages / returns live in (`advantages = send[0]`, `returns = send[1]`: dn_gae writes straight into the send
    buffer, no packing copy), `recv` [R, 2, T, N_local] is what `all_gather_into_tensor` fills, and the global arrays are
    returned as strided VIEWS of it, shaped [T, R, N_local]: element [t, r, i] belongs to global drone r * N_local + i (rank-major
    = global drone order), so `view.reshape(T, R * N_local)` is the [n_steps, N_global] array where a consumer needs it flat.
    67 MB per rollout at R = 8, n_steps = 32, 32 768 drones per rank: neither allocated nor copied per rollout.
    Without a process group gather() returns the local arrays as [T, 1, N_local] views (no collective)."""

    def __init__(self, n_steps, num_envs, device, group=None, dtype=torch.float32):
        import torch.distributed as dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.send = torch.zeros((2, int(n_steps), int(num_envs)), dtype=dtype, device=device)
        self.recv = torch.zeros((self.world, 2, int(n_steps), int(num_envs)), dtype=dtype, device=device) if self.active else None

    advantages = property(lambda self: self.send[0])
    returns = property(lambda self: self.send[1])

    def gather(self):
        if not self.active:
            return self.send[0].unsqueeze(1), self.send[1].unsqueeze(1)
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.recv.view(self.world * 2, *self.send.shape[1:]), self.send, group=self.group)
        return self.recv[:, 0].permute(1, 0, 2), self.recv[:, 1].permute(1, 0, 2)      # [T, R, n] views, no copy


class RolloutCollector:
    """n_steps x (policy -> dn_step) on one GPU's shard, then GAE (+ optional all-gather).

    policy(obs[N,13] f32) -> (actions[N,4] f32, values[N] f32, log_probs[N] f32), all on the env's device;
    value_fn(obs) -> values[N] (defaults to the policy's second output).  Actions are clipped to the action
    space before the step, as SB3 does.  `bootstrap_truncated` adds gamma * V(terminal_observation) to the reward
    of drones whose episode hit the time limit (SB3's TimeLimit handling).

    `use_graph`: the whole rollout -- n_steps x (policy kernels, clip, dn_step, bootstrap, buffer writes) and the
    GAE kernel -- is captured into one hipGraph on the second call to collect() (the first call runs eagerly and
    doubles as the warm-up) and replayed afterwards: the loop is launch-bound (a 13->512->512->256 MLP step is a
    dozen small kernels), and a graph replay takes the host out of it.  Every tensor the loop touches is a static
    buffer; the environment's vector-step counter lives on the device, so noise streams keep advancing under replay.
    The policy must be capture-safe (no host synchronisation, no data-dependent shapes)."""

    def __init__(self, env, policy, n_steps, *, value_fn=None, gamma=0.99, gae_lambda=0.95, bootstrap_truncated=True,
                 gather=False, group=None, use_graph=False):
        from .vec_env import ACT_DIM, DroneVecEnv
        if not isinstance(env, DroneVecEnv):
            raise TypeError("RolloutCollector drives a DroneVecEnv (HIP); there is no CPU path")
        self.env, self.policy, self.value_fn = env, policy, value_fn
        self.n_steps, self.gamma, self.gae_lambda = int(n_steps), float(gamma), float(gae_lambda)
        self.bootstrap_truncated, self.gather, self.group = bool(bootstrap_truncated), bool(gather), group
        self.use_graph = bool(use_graph)
        import inspect
        try:        # a value function that can skip unflagged tiles (FusedMlpPolicy.predict_values) gets the truncation mask
            self._value_fn_takes_mask = value_fn is not None and "row_mask" in inspect.signature(value_fn).parameters
        except (TypeError, ValueError):
            self._value_fn_takes_mask = False
        n, T, dev, f32 = env.num_envs, self.n_steps, env.device, torch.float32
        self.buf = dict(
            obs=torch.empty((T, n, env.obs_dim), dtype=f32, device=dev),
            actions=torch.empty((T, n, ACT_DIM), dtype=f32, device=dev),
            values=torch.empty((T, n), dtype=f32, device=dev), log_probs=torch.empty((T, n), dtype=f32, device=dev),
            rewards=torch.empty((T, n), dtype=f32, device=dev),
            episode_starts=torch.empty((T, n), dtype=torch.uint8, device=dev),
            advantages=torch.empty((T, n), dtype=f32, device=dev), returns=torch.empty((T, n), dtype=f32, device=dev),
            last_values=torch.empty(n, dtype=f32, device=dev), last_dones=torch.empty(n, dtype=torch.uint8, device=dev))
        self._clipped = torch.empty((n, ACT_DIM), dtype=f32, device=dev)
        self._last_obs = env.reset_tensor().clone()
        self._last_done = torch.ones(n, dtype=torch.uint8, device=dev)       # SB3: _last_episode_starts = True
        self.num_timesteps = 0
        self._graph = None
        self._calls = 0

    def _values(self, obs, row_mask=None):
        if self.value_fn is not None:
            if row_mask is not None and self._value_fn_takes_mask:
                return self.value_fn(obs, row_mask=row_mask).reshape(-1)
            return self.value_fn(obs).reshape(-1)
        return self.policy(obs)[1].reshape(-1)

    def _rollout(self):
        """One rollout on the current stream, in place in the static buffers (eager or under graph capture)."""
        from . import _capi
        import ctypes as C
        env, b = self.env, self.buf
        obs, done = self._last_obs, self._last_done
        for t in range(self.n_steps):
            actions, values, log_probs = self.policy(obs)
            b["obs"][t].copy_(obs)
            b["episode_starts"][t].copy_(done)
            b["values"][t].copy_(values.reshape(-1))
            b["log_probs"][t].copy_(log_probs.reshape(-1))
            b["actions"][t].copy_(actions)
            torch.clamp(b["actions"][t], -1.0, 1.0, out=self._clipped)
            next_obs, reward, next_done, info = env.step_tensor(self._clipped, want_terminal=self.bootstrap_truncated)
            if self.bootstrap_truncated:
                # rows of terminal_obs are valid only where done; `truncated` is zero elsewhere
                tv = self._values(torch.where(next_done.bool()[:, None], info["terminal_obs"], next_obs),
                                  row_mask=info["truncated"])
                reward = reward + self.gamma * tv * info["truncated"].to(reward.dtype)
            b["rewards"][t].copy_(reward)
            obs.copy_(next_obs)
            done.copy_(next_done)
        b["last_values"].copy_(self._values(obs))
        b["last_dones"].copy_(done)
        dev = env.device
        _capi.check(_capi.load().dn_gae(
            b["rewards"].data_ptr(), b["values"].data_ptr(), b["episode_starts"].data_ptr(), b["last_values"].data_ptr(),
            b["last_dones"].data_ptr(), self.n_steps, env.num_envs, self.gamma, self.gae_lambda,
            b["advantages"].data_ptr(), b["returns"].data_ptr(), dev.index,
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    @torch.no_grad()
    def collect(self):
        """One rollout.  Returns the buffer dict (static tensors, overwritten by the next call) with `advantages`,
        `returns` ([n_steps, N_local]) and, with `gather`, `advantages_global` / `returns_global` ([n_steps, N_global])."""
        env = self.env
        with torch.cuda.device(env.device):
            if self.use_graph and self._calls >= 1:
                if self._graph is None:
                    torch.cuda.synchronize(env.device)
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._rollout()
                self._graph.replay()
            else:
                self._rollout()
        self._calls += 1
        self.num_timesteps += self.n_steps * env.num_envs
        out = dict(self.buf)
        if self.gather:
            out["advantages_global"], out["returns_global"] = all_gather_rollout(out["advantages"], out["returns"], self.group)
        return out


class ReplayBuffer:
    """Device-resident ring buffer with the layout and semantics of SB3's `ReplayBuffer` as the reference's SAC/DDPG
    agents use it (PBDroneSimulator.py:297-338; `SaveReplayBufferCallback`, Sol/Utilities/Callbacks.py:13-39)
    [3P-recall]: per slot and drone `obs`, `next_obs` (the TERMINAL observation where the episode ended, not the reset
    one), `action`, `reward`, `done` (with `handle_timeout_termination`: a TimeLimit truncation is stored as not done)
    and `timeout`.  Step-major [buffer_size, N, ...]; `add` is a handful of in-place copies, `sample` a gather."""

    def __init__(self, buffer_size, num_envs, obs_dim, act_dim, device):
        f32, dev = torch.float32, torch.device(device)
        self.buffer_size, self.num_envs = int(buffer_size), int(num_envs)
        self.obs = torch.empty((self.buffer_size, num_envs, obs_dim), dtype=f32, device=dev)
        self.next_obs = torch.empty_like(self.obs)
        self.actions = torch.empty((self.buffer_size, num_envs, act_dim), dtype=f32, device=dev)
        self.rewards = torch.empty((self.buffer_size, num_envs), dtype=f32, device=dev)
        self.dones = torch.empty((self.buffer_size, num_envs), dtype=f32, device=dev)
        self.timeouts = torch.empty((self.buffer_size, num_envs), dtype=f32, device=dev)
        self.pos, self.full = 0, False

    def add(self, obs, next_obs, action, reward, done, timeout):
        p = self.pos
        self.obs[p].copy_(obs)
        self.next_obs[p].copy_(next_obs)
        self.actions[p].copy_(action)
        self.rewards[p].copy_(reward)
        self.dones[p].copy_(done)
        self.timeouts[p].copy_(timeout)
        self.pos = (p + 1) % self.buffer_size
        self.full = self.full or self.pos == 0

    def __len__(self):
        return (self.buffer_size if self.full else self.pos) * self.num_envs

    def sample(self, batch_size, generator=None):
        upper = self.buffer_size if self.full else self.pos
        if upper == 0:
            raise RuntimeError("the replay buffer is empty")
        dev = self.obs.device
        t = torch.randint(0, upper, (batch_size,), device=dev, generator=generator)
        e = torch.randint(0, self.num_envs, (batch_size,), device=dev, generator=generator)
        return dict(obs=self.obs[t, e], next_obs=self.next_obs[t, e], actions=self.actions[t, e], rewards=self.rewards[t, e],
                    dones=self.dones[t, e] * (1.0 - self.timeouts[t, e]))


class RingReplayBuffer:
    """The same transitions as ReplayBuffer, laid out so that the kernels write them in place: `obs_ring[t + 1]` is both the
    observation the step at slot t produced and the input of the step at slot t + 1 (SB3's `optimize_memory_usage` idea, with one
    spare row, so that every slot is whole at a cycle boundary; mid-cycle the slot being replaced is excluded), `terminal_obs[t]` holds the terminal observations dn_step reports, `dones` /
    `timeouts` stay the uint8 flags the kernel writes.  SB3's view of a transition is assembled on demand: `next_obs` = the
    TERMINAL observation where the episode ended, the next one elsewhere; `dones`, `timeouts` as float32."""

    def __init__(self, buffer_size, num_envs, obs_dim, act_dim, device):
        f32, u8, dev = torch.float32, torch.uint8, torch.device(device)
        self.buffer_size, self.num_envs = int(buffer_size), int(num_envs)
        self.obs_ring = torch.zeros((self.buffer_size + 1, num_envs, obs_dim), dtype=f32, device=dev)
        self.terminal_obs = torch.zeros((self.buffer_size, num_envs, obs_dim), dtype=f32, device=dev)
        self.actions = torch.zeros((self.buffer_size, num_envs, act_dim), dtype=f32, device=dev)
        self.rewards = torch.zeros((self.buffer_size, num_envs), dtype=f32, device=dev)
        self.done_flags = torch.zeros((self.buffer_size, num_envs), dtype=u8, device=dev)
        self.timeout_flags = torch.zeros((self.buffer_size, num_envs), dtype=u8, device=dev)
        self.pos, self.full = 0, False

    obs = property(lambda self: self.obs_ring[:self.buffer_size])
    next_obs = property(lambda self: torch.where(self.done_flags.bool()[..., None], self.terminal_obs, self.obs_ring[1:]))
    dones = property(lambda self: self.done_flags.float())
    timeouts = property(lambda self: self.timeout_flags.float())

    def valid_slots(self):
        """Slots whose transition is whole.  Mid-cycle on a full ring the OLDEST slot (= pos, the next to be written) has lost
        its observation to the newest transition's next observation (they share a row); at a cycle boundary (pos = 0) the
        spare row keeps all buffer_size slots whole."""
        if not self.full:
            return list(range(self.pos))
        return [t for t in range(self.buffer_size) if t != self.pos or self.pos == 0]

    def __len__(self):
        return len(self.valid_slots()) * self.num_envs

    def _sample_slots(self, batch_size, generator=None):
        dev = self.obs_ring.device
        upper = (self.buffer_size - (1 if self.pos else 0)) if self.full else self.pos
        if upper == 0:
            raise RuntimeError("the replay buffer is empty")
        t = torch.randint(0, upper, (batch_size,), device=dev, generator=generator)
        if self.full and self.pos:
            t = t + (t >= self.pos).long()                                   # skip the slot that is being replaced
        return t, torch.randint(0, self.num_envs, (batch_size,), device=dev, generator=generator)

    def sample(self, batch_size, generator=None):
        t, e = self._sample_slots(batch_size, generator)
        d = self.done_flags[t, e].bool()
        return dict(obs=self.obs_ring[t, e], next_obs=torch.where(d[:, None], self.terminal_obs[t, e], self.obs_ring[t + 1, e]),
                    actions=self.actions[t, e], rewards=self.rewards[t, e],
                    dones=d.float() * (1.0 - self.timeout_flags[t, e].float()))


class OffPolicyCollector:
    """BASELINE config 5's collection loop (SAC: one environment step per policy step, every transition into the
    replay buffer): actor(obs) -> actions in [-1, 1] -> dn_step -> replay buffer, with SB3's terminal-observation
    handling.  Sharded like the on-policy collector: a rank's drones feed the rank's buffer, no collective.

    With a policy_mfma.FusedSacActor the loop is two launches per step and no copies: dn_mlp_forward (mu | log_std from the
    ring's current observation row) and dn_step_squashed (clamp, Philox draw, tanh inside the step kernel: the action, the next
    observation, reward, flags and terminal observation land in their buffer slots) -- RingReplayBuffer; three launches
    (dn_squashed_sample + dn_step, the same bits) with `fused_sample=False` or with the options dn_step_squashed is not built for.  With any other
    torch callable the actions come from the callable and the transitions are copied into a ReplayBuffer.

    `collect_cycle()`: one whole pass over the ring buffer (buffer_size steps, slot 0 .. buffer_size - 1) captured into
    a hipGraph on its second call and replayed afterwards -- every tensor of the loop is static and the ring position is
    back where it started, so the host leaves the loop as it does for RolloutCollector(use_graph=True)."""

    def __init__(self, env, actor, buffer_size, *, seed=0, deterministic=False, fused_sample=True):
        from .policy_mfma import FusedSacActor
        from .vec_env import ACT_DIM, DroneVecEnv
        if not isinstance(env, DroneVecEnv):
            raise TypeError("OffPolicyCollector drives a DroneVecEnv (HIP); there is no CPU path")
        self.env, self.actor = env, actor
        self.direct = isinstance(actor, FusedSacActor)
        self.seed, self.deterministic = int(seed), bool(deterministic)
        if self.direct:
            if env.obs_dim != 13 or env.num_envs != actor._out.shape[0]:
                raise ValueError("the direct loop needs the full 13-column observation and an actor built for env.num_envs drones")
            self.buffer = RingReplayBuffer(buffer_size, env.num_envs, env.obs_dim, ACT_DIM, env.device)
            self.buffer.obs_ring[0].copy_(env.reset_tensor())
            self._found = torch.zeros(env.num_envs, dtype=torch.int32, device=env.device)
            cfg = env.cfg
            self._fused_sample = fused_sample and not (cfg.clip_rew or cfg.norm_rew or cfg.physics or cfg.action_type or cfg.random_spawn
                                                       or cfg.zero_damping)               # dn_step_squashed's scope
        else:
            self.buffer = ReplayBuffer(buffer_size, env.num_envs, env.obs_dim, ACT_DIM, env.device)
            self._obs = env.reset_tensor().clone()
        self.num_timesteps = 0
        self._graph, self._cycles, self._carry = None, 0, False

    def _steps_direct(self, n_steps):
        import ctypes as C
        from . import _capi
        from .policy_mfma import mlp_forward
        env, buf, lib = self.env, self.buffer, _capi.load()
        h, T = env._handle, buf.buffer_size
        sptr = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
        out8 = self.actor._out
        for _ in range(int(n_steps)):
            p = buf.pos
            if p == 0 and self._carry:                      # wrapped: the newest observation (row T) becomes row 0 only now, when
                buf.obs_ring[0].copy_(buf.obs_ring[T])      # slot 0's old transition is about to be replaced as a whole
                self._carry = False
            mlp_forward([self.actor.pack], buf.obs_ring[p], [out8])
            outs = (buf.obs_ring[p + 1].data_ptr(), buf.rewards[p].data_ptr(), buf.done_flags[p].data_ptr(), buf.timeout_flags[p].data_ptr(),
                    self._found.data_ptr(), buf.terminal_obs[p].data_ptr(), None, None, None, sptr)
            if self._fused_sample:                          # the draw inside the step kernel: two launches per step
                _capi.check(lib.dn_step_squashed(h, out8.data_ptr(), self.seed, int(self.deterministic), buf.actions[p].data_ptr(), None, *outs))
            else:
                _capi.check(lib.dn_squashed_sample(h, out8.data_ptr(), self.seed, int(self.deterministic), buf.actions[p].data_ptr(), None, sptr))
                _capi.check(lib.dn_step(h, buf.actions[p].data_ptr(), *outs))
            buf.pos = p + 1
            if buf.pos == T:
                buf.pos, buf.full, self._carry = 0, True, True

    def _steps(self, n_steps):
        if self.direct:
            return self._steps_direct(n_steps)
        env = self.env
        for _ in range(int(n_steps)):
            actions = self.actor(self._obs).clamp(-1.0, 1.0)
            next_obs, reward, done, info = env.step_tensor(actions, want_terminal=True)
            d = done.bool()[:, None]
            self.buffer.add(self._obs, torch.where(d, info["terminal_obs"], next_obs), actions, reward, done.float(),
                            info["truncated"].float())
            self._obs.copy_(next_obs)

    @torch.no_grad()
    def collect(self, n_steps=1):
        with torch.cuda.device(self.env.device):
            self._steps(n_steps)
        self.num_timesteps += int(n_steps) * self.env.num_envs
        return self.buffer

    @torch.no_grad()
    def collect_cycle(self):
        if self.buffer.pos != 0:
            raise RuntimeError("collect_cycle() fills slots 0 .. buffer_size - 1: call it with the ring at position 0")
        T = self.buffer.buffer_size
        with torch.cuda.device(self.env.device):
            if self._cycles >= 1:
                if self._graph is None:
                    torch.cuda.synchronize(self.env.device)
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._steps(T)
                    self.buffer.pos = 0                     # the capture pass advanced the host-side ring position only
                self._graph.replay()
                self.buffer.full, self._carry = True, True
            else:
                self._steps(T)
        self._cycles += 1
        self.num_timesteps += T * self.env.num_envs
        return self.buffer


class FusedRolloutCollector:
    """The on-policy rollout with every per-step operation on a purpose-built kernel and no intermediate copies:

        dn_mlp_forward (actor + critic, one launch; the value lands in the rollout buffer)
        dn_step_sampled (Gaussian sample from the environment's Philox streams, clip, log-probability -> buffer, and the
                         step itself: next observation, reward and episode-start flag straight into the buffer slots)
    two launches per step instead of ~25 small torch kernels -- or ONE with `one_launch=True` (dn_mlp_step_sampled: the policy kernel's
    workgroups step the drones they evaluated; the same bits).  Off by default because it measured SLOWER on MI355X at 32 768 drones
    (80.7 against 69.9 us per step with bf16 networks, 179.9 against 174.1 in the float32 grade): the single step is latency bound, inside
    the policy kernel it runs on the critical path of every actor workgroup with nothing to overlap it (LDS keeps one workgroup per CU),
    and the step's registers push spills into the MFMA loops; the kernel boundary it removes costs 1.7 us (profiles/r03_notes.md); then, once per rollout, SB3's TimeLimit bootstrap
    (rewards[t] += gamma V(terminal_observation) where TimeLimit.truncated): the step kernel leaves the terminal
    observations and truncation flags of all n_steps in the buffer, ONE dn_mlp_forward masked by those flags evaluates
    the critic on the tiles that hold a truncated drone (the weights do not change inside a rollout, so this is the
    value SB3 computes on the spot) and ONE dn_add_bootstrap adds it; then dn_gae.  The whole rollout is captured into one
    hipGraph on the second call (`use_graph`).  Same results contract as RolloutCollector (SB3 collect_rollouts
    semantics); the action noise comes from Philox (seed, global drone id, step counter), so a rollout is reproducible
    and independent of how the drones are sharded.  Call `policy.refresh()` after optimiser steps (it re-packs in place, so
    the captured launches see the new weights); a changed log_std (baked into the captured launches by value) is noticed at the
    next collect(), which captures again (`recapture()` forces it).
    With `gather`, `advantages_global` / `returns_global` are [n_steps, R, N_local] views of a static receive buffer
    (RolloutGather: one all_gather_into_tensor per rollout, nothing allocated or copied per rollout)."""

    def __init__(self, env, policy, n_steps, *, gamma=0.99, gae_lambda=0.95, bootstrap_truncated=True, gather=False,
                 group=None, use_graph=True, seed=0, one_launch=False):
        from .policy_mfma import FusedMlpPolicy
        from .vec_env import ACT_DIM, OBS_DIM, DroneVecEnv
        if not isinstance(env, DroneVecEnv) or not isinstance(policy, FusedMlpPolicy):
            raise TypeError("FusedRolloutCollector needs a DroneVecEnv and a FusedMlpPolicy (HIP); there is no CPU path")
        if env.num_envs % 4 or env.obs_dim != OBS_DIM:
            raise ValueError("num_envs must be a multiple of 4 and the observation the full 13 columns")
        self.env, self.policy = env, policy
        self.n_steps, self.gamma, self.gae_lambda = int(n_steps), float(gamma), float(gae_lambda)
        self.bootstrap_truncated, self.gather, self.group, self.use_graph = bool(bootstrap_truncated), bool(gather), group, bool(use_graph)
        self.seed = int(seed)
        n, T, dev, f32, u8 = env.num_envs, self.n_steps, env.device, torch.float32, torch.uint8
        self.buf = dict(
            obs=torch.empty((T + 1, n, OBS_DIM), dtype=f32, device=dev),
            actions=torch.empty((T, n, ACT_DIM), dtype=f32, device=dev),
            values=torch.empty((T, n), dtype=f32, device=dev), log_probs=torch.empty((T, n), dtype=f32, device=dev),
            rewards=torch.empty((T, n), dtype=f32, device=dev),
            episode_starts=torch.ones((T + 1, n), dtype=u8, device=dev),          # SB3: _last_episode_starts = True
            last_values=torch.empty((n, 1), dtype=f32, device=dev))
        # advantages / returns live in the all-gather's send buffer (RolloutGather): dn_gae writes them where RCCL reads them
        self._gather = RolloutGather(T, n, dev, group) if self.gather else None
        if self._gather is not None:
            self.buf["advantages"], self.buf["returns"] = self._gather.advantages, self._gather.returns
        else:
            self.buf["advantages"], self.buf["returns"] = torch.empty((T, n), dtype=f32, device=dev), torch.empty((T, n), dtype=f32, device=dev)
        self._mean = torch.empty((n, ACT_DIM), dtype=f32, device=dev)
        cfg = env.cfg
        self._sampled_step = not (cfg.clip_rew or cfg.norm_rew or cfg.physics or cfg.action_type or cfg.random_spawn or cfg.zero_damping)   # dn_step_sampled's scope
        self._clipped = None if self._sampled_step else torch.empty((n, ACT_DIM), dtype=f32, device=dev)
        # ONE launch per step (dn_mlp_step_sampled: the actor's workgroups step the drones they evaluated) where that kernel is built:
        # the float64 reference configuration without noise / ground contact, fleets on the three-wave single step, whole workgroups
        self._one_launch = bool(one_launch) and self._sampled_step and not (cfg.ground_contact or cfg.act_noise_sigma > 0 or cfg.obs_noise_sigma > 0
                                                                          or cfg.compute_f32) \
            and env.kernel_waves(fused=False) == 3 and n % (64 if policy.grade == "fp32" else 128) == 0
        self._trunc = torch.zeros((T, n), dtype=u8, device=dev)
        self._found = torch.zeros(n, dtype=torch.int32, device=dev)
        # terminal observations of every step of the rollout (rows are written where a drone finished; the rest is stale
        # and never used: the masked forward and the bootstrap select by `truncated`)
        self._term_obs = torch.zeros((T, n, OBS_DIM), dtype=f32, device=dev) if self.bootstrap_truncated else None
        self._tv = torch.zeros((T * n, 1), dtype=f32, device=dev)
        self.buf["obs"][0].copy_(env.reset_tensor())
        self.num_timesteps = 0
        self._graph, self._calls, self._captured_log_std = None, 0, None

    def recapture(self):
        self._graph = None

    def _rollout(self):
        import ctypes as C
        from . import _capi
        from .policy_mfma import mlp_forward
        env, pol, b, T = self.env, self.policy, self.buf, self.n_steps
        lib, h, n, dev = _capi.load(), env._handle, env.num_envs, env.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        log_std = (C.c_float * 4)(*[float(x) for x in pol.log_std_host])
        from .policy_mfma import _net_struct
        for t in range(T):
            obs_t = b["obs"][t]
            term = self._term_obs[t].data_ptr() if self.bootstrap_truncated else None
            if self._one_launch:
                # policy forward (actor + critic), Gaussian sample, clip, log-probability and the environment step in ONE launch
                nets = (_capi.DnMlpNet * 2)(_net_struct(pol.pi, self._mean), _net_struct(pol.vf, b["values"][t].view(n, 1)))
                _capi.check(lib.dn_mlp_step_sampled(h, C.cast(nets, C.c_void_p), 2, obs_t.data_ptr(), obs_t.shape[1], log_std, self.seed, 0,
                                                    b["actions"][t].data_ptr(), b["log_probs"][t].data_ptr(), b["obs"][t + 1].data_ptr(),
                                                    b["rewards"][t].data_ptr(), b["episode_starts"][t + 1].data_ptr(), self._trunc[t].data_ptr(),
                                                    self._found.data_ptr(), term, None, None, None, stream))
                continue
            mlp_forward([pol.pi, pol.vf], obs_t, [self._mean, b["values"][t].view(n, 1)])
            if self._sampled_step:
                # Gaussian sample (Philox), clip, log-probability and the environment step in ONE launch (dn_step_sampled)
                _capi.check(lib.dn_step_sampled(h, self._mean.data_ptr(), log_std, self.seed, 0, b["actions"][t].data_ptr(),
                                                b["log_probs"][t].data_ptr(), b["obs"][t + 1].data_ptr(), b["rewards"][t].data_ptr(),
                                                b["episode_starts"][t + 1].data_ptr(), self._trunc[t].data_ptr(),
                                                self._found.data_ptr(), term, None, None, None, stream))
            else:                                              # reward wrappers / extra physics terms / RPM actions: two launches
                _capi.check(lib.dn_policy_sample(h, self._mean.data_ptr(), log_std, self.seed, 0, b["actions"][t].data_ptr(),
                                                 self._clipped.data_ptr(), b["log_probs"][t].data_ptr(), stream))
                _capi.check(lib.dn_step(h, self._clipped.data_ptr(), b["obs"][t + 1].data_ptr(), b["rewards"][t].data_ptr(),
                                        b["episode_starts"][t + 1].data_ptr(), self._trunc[t].data_ptr(), self._found.data_ptr(),
                                        term, None, None, None, stream))
        if self.bootstrap_truncated:
            mlp_forward([pol.vf], self._term_obs.view(T * n, -1), [self._tv], row_mask=self._trunc.view(T * n))
            _capi.check(lib.dn_add_bootstrap(b["rewards"].data_ptr(), self._tv.data_ptr(), self._trunc.data_ptr(),
                                             self.gamma, T * n, dev.index, stream))
        mlp_forward([pol.vf], b["obs"][T], [b["last_values"]])
        _capi.check(lib.dn_gae(b["rewards"].data_ptr(), b["values"].data_ptr(), b["episode_starts"].data_ptr(),
                               b["last_values"].data_ptr(), b["episode_starts"][T].data_ptr(), T, n, self.gamma,
                               self.gae_lambda, b["advantages"].data_ptr(), b["returns"].data_ptr(), dev.index, stream))

    @torch.no_grad()
    def collect(self):
        """One rollout; returns views of the static buffers (obs / episode_starts hold n_steps + 1 slots: the last one is
        the observation / start flag the next rollout begins with)."""
        env, b, T = self.env, self.buf, self.n_steps
        with torch.cuda.device(env.device):
            if self._calls > 0:                            # carry the last observation / start flags over
                b["obs"][0].copy_(b["obs"][T])
                b["episode_starts"][0].copy_(b["episode_starts"][T])
            if self.use_graph and self._calls >= 1:
                # dn_step_sampled takes log_std by value (a HOST float[4]): it is baked into the captured launches, so a
                # refresh() that moved it (PPO trains log_std) makes the graph stale -- capture again
                if self._graph is not None and tuple(self.policy.log_std_host) != self._captured_log_std:
                    self._graph = None
                if self._graph is None:
                    torch.cuda.synchronize(env.device)
                    self._graph = torch.cuda.CUDAGraph()
                    self._captured_log_std = tuple(self.policy.log_std_host)
                    with torch.cuda.graph(self._graph):
                        self._rollout()
                self._graph.replay()
            else:
                self._rollout()
        self._calls += 1
        self.num_timesteps += T * env.num_envs
        out = dict(obs=b["obs"][:T], actions=b["actions"], values=b["values"], log_probs=b["log_probs"], rewards=b["rewards"],
                   episode_starts=b["episode_starts"][:T], advantages=b["advantages"], returns=b["returns"],
                   last_values=b["last_values"].view(-1), last_dones=b["episode_starts"][T], next_obs=b["obs"][T])
        if self.gather:
            # [n_steps, R, N_local] strided views of the static receive buffer ([t, r, i] = global drone r * N_local + i)
            out["advantages_global"], out["returns_global"] = self._gather.gather()
        return out
