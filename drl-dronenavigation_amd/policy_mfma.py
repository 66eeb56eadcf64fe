"""The reference's policy/value networks (PPO actor + critic, SAC actor) on the fused MFMA kernels (csrc/dn_mlp.hip,
dn_mlp_forward).

`pack_mlp` turns the float32 [out, in] matrices of one 13-512-512-256-out network into the bfloat16 fragment order
the kernel loads (see dn_mlp.hip for why the K index of every layer but the first is permuted), `FusedMlpPolicy`
wraps an `MlpActorCritic` so that the rollout collector's policy(obs) -> (actions, values, log_probs) runs both
trunks and both heads in ONE kernel launch; the Gaussian sample and its log-probability (a few element-wise ops on
[N, 4]) stay in torch, float32.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _capi

HIDDEN = (512, 512, 256)


def _k_order(in_features, first):
    """Input-feature index read by (K-step kk, lane group g, slot s) -> [KS, 2, 8]."""
    if first:
        ks = 1
        f = 16 * np.arange(ks)[:, None, None] + 8 * np.arange(2)[None, :, None] + np.arange(8)[None, None, :]
        return np.where(f < in_features, f, -1)
    ks = in_features // 16
    kk = np.arange(ks)[:, None, None]
    g = np.arange(2)[None, :, None]
    r = 8 * (kk & 1) + np.arange(8)[None, None, :]
    return 32 * (kk >> 1) + 4 * g + (r & 3) + 8 * (r >> 2)


TANH_PRESCALE = 2.0 / math.log(2.0)          # 2 log2(e): the kernel's tanh is 1 - 2 / (1 + 2^x) on pre-scaled x


def pack_layer(weight, bias, first, scale=1.0, grade="bf16"):
    """weight [out, in] float32, bias [out] -> (packed bf16 [MT, KS, 64, 8] as a torch tensor, bias padded to 32*MT).
    `scale` multiplies both before the bf16 rounding (TANH_PRESCALE for the layers that feed a tanh).
    grade="fp32": the weights are split into two bf16 words, w = w_hi + w_lo (16 mantissa bits), and packed per M-tile as
    the KS hi fragments followed by the KS lo fragments, [MT, 2, KS, 64, 8] -- the stream of dn_mlp_x3_kernel."""
    w = np.asarray(weight.detach().cpu().float().numpy() if hasattr(weight, "detach") else weight, dtype=np.float32)
    b = np.asarray(bias.detach().cpu().float().numpy() if hasattr(bias, "detach") else bias, dtype=np.float32)
    w, b = (w * np.float32(scale)).astype(np.float32), (b * np.float32(scale)).astype(np.float32)
    out_f, in_f = w.shape
    if not first and in_f % 32:
        raise ValueError("hidden widths must be multiples of 32")
    mt = (out_f + 31) // 32
    korder = _k_order(in_f, first)                               # [KS, 2, 8]
    ks = korder.shape[0]
    wp = np.zeros((mt * 32, in_f + 1), np.float32)               # extra zero column for the -1 (padding) index
    wp[:out_f, :in_f] = w
    rows = (32 * np.arange(mt)[:, None] + np.arange(32)[None, :])            # [MT, 32]
    lane_row = np.concatenate([rows, rows], axis=1)                          # lane l -> row 32 mo + (l & 31)   [MT, 64]
    lane_g = np.repeat(np.arange(2), 32)                                     # lane l -> group l >> 5           [64]
    cols = korder[:, lane_g, :]                                              # [KS, 64, 8]
    packed = wp[lane_row[:, None, :, None], cols[None, :, :, :]]             # [MT, KS, 64, 8]
    bp = np.zeros(mt * 32, np.float32)
    bp[:out_f] = b
    pt = torch.from_numpy(packed)
    if grade == "fp32":
        hi = pt.to(torch.bfloat16)
        lo = (pt - hi.float()).to(torch.bfloat16)
        return torch.stack((hi, lo), dim=1).contiguous(), torch.from_numpy(bp)          # [MT, 2, KS, 64, 8]
    if grade == "fp16":
        return pt.to(torch.float16).contiguous(), torch.from_numpy(bp)
    return pt.to(torch.bfloat16).contiguous(), torch.from_numpy(bp)


GRADES = {"bf16": 0, "fp32": 1, "fp16": 2}


def pack_mlp(layers, device, grade="bf16"):
    """layers: [(W1, b1), (W2, b2), (W3, b3), (Wh, bh)] of one network -> dict of device tensors for dn_mlp_net.
    grade: "bf16" (bf16 operands, ~1e-2 on the outputs), "fp16" (float16 operands: the same speed, ~1e-3) or "fp32" (the
    reference's precision through split-bf16 operands and three MFMAs per product, <= 1e-4; about 3x the time)."""
    if grade not in GRADES:
        raise ValueError(f"grade must be one of {sorted(GRADES)}")
    if [tuple(w.shape) for w, _ in layers[1:3]] != [(HIDDEN[1], HIDDEN[0]), (HIDDEN[2], HIDDEN[1])] or \
            layers[0][0].shape[0] != HIDDEN[0] or layers[3][0].shape[1] != HIDDEN[2]:
        raise ValueError("the fused kernel is built for obs -> 512 -> 512 -> 256 -> out (PBDroneSimulator.py:251-258)")
    out = {}
    for name, (w, b), first in zip(("1", "2", "3", "h"), layers, (True, False, False, False)):
        pw, pb = pack_layer(w, b, first, scale=1.0 if name == "h" else TANH_PRESCALE, grade=grade)
        out["w" + name], out["b" + name] = pw.to(device), pb.to(device)
    out["grade"] = GRADES[grade]
    out["out_dim"] = int(layers[3][0].shape[0])
    out["obs_dim"] = int(layers[0][0].shape[1])
    return out


ARCH_PPO, ARCH_SAC = 0, 1
SAC_HIDDEN = (256, 256)


def pack_sac_actor(layers, device, grade="bf16"):
    """layers: [(W1, b1), (W2, b2), (Wmu, bmu), (Wls, bls)] of the reference's SAC actor (obs -> 256 -> 256 -> mu | log_std,
    ReLU; PBDroneSimulator.py:297-303) -> dict of device tensors for dn_mlp_net with arch = DN_MLP_ARCH_SAC: the two heads
    are stacked into one [2 act_dim, 256] matrix (rows [0, act_dim) mu, [act_dim, 2 act_dim) log_std)."""
    if grade not in GRADES:
        raise ValueError(f"grade must be one of {sorted(GRADES)}")
    (w1, b1), (w2, b2), (wm, bm), (ws, bs) = layers
    if w1.shape[0] != SAC_HIDDEN[0] or tuple(w2.shape) != (SAC_HIDDEN[1], SAC_HIDDEN[0]) or wm.shape[1] != SAC_HIDDEN[1] \
            or tuple(ws.shape) != tuple(wm.shape):
        raise ValueError("the SAC actor kernel is built for obs -> 256 -> 256 -> (mu, log_std) (PBDroneSimulator.py:297-303)")
    f = lambda t: t.detach().cpu().float()          # noqa: E731
    wh, bh = torch.cat((f(wm), f(ws)), 0), torch.cat((f(bm), f(bs)), 0)
    out = {}
    for name, (w, b), first in zip(("1", "2", "h"), ((w1, b1), (w2, b2), (wh, bh)), (True, False, False)):
        pw, pb = pack_layer(w, b, first, scale=1.0, grade=grade)
        out["w" + name], out["b" + name] = pw.to(device), pb.to(device)
    out["grade"], out["arch"] = GRADES[grade], ARCH_SAC
    out["out_dim"], out["obs_dim"] = int(wh.shape[0]), int(w1.shape[1])
    return out


def _repack_into(dst, src):
    """Refresh a packed network IN PLACE: a hipGraph that captured dn_mlp_forward holds the device addresses of the packed
    weights, so a refresh after an optimiser step must land in the same allocations (a re-pack into fresh tensors would leave
    every captured launch reading the old, freed buffers).  `dst` is the pack the kernels were given, `src` a new pack of the
    same network shape and grade."""
    if dst is None:
        return src
    for k, v in src.items():
        if torch.is_tensor(v):
            if k not in dst or dst[k].shape != v.shape or dst[k].dtype != v.dtype:
                raise ValueError(f"refresh() changed the packed shape of {k!r}: build a new policy object (and recapture)")
            dst[k].copy_(v)
        elif dst.get(k) != v:
            raise ValueError(f"refresh() changed {k!r} ({dst.get(k)} -> {v}): build a new policy object (and recapture)")
    return dst


def _net_struct(pk, out_tensor):
    n = _capi.DnMlpNet()
    for k in ("w1", "w2", "w3", "wh", "b1", "b2", "b3", "bh"):
        if k in pk:
            setattr(n, k, pk[k].data_ptr())
    n.out, n.out_dim, n.grade, n.arch = out_tensor.data_ptr(), pk["out_dim"], pk.get("grade", 0), pk.get("arch", ARCH_PPO)
    return n


def mlp_forward(packs, obs, outs=None, row_mask=None):
    """Run one or two packed networks over obs [N, obs_dim] (float32, CUDA) in one launch; returns the list of
    float32 outputs [N, out_dim].  row_mask (uint8 [N], optional): 32-drone tiles without a flagged drone are skipped
    and their outputs zeroed."""
    if obs.device.type != "cuda" or obs.dtype != torch.float32 or obs.dim() != 2:
        raise ValueError("obs must be a float32 CUDA tensor [N, obs_dim]; there is no CPU fallback")
    obs = obs.contiguous()
    n, dev = obs.shape[0], obs.device
    if outs is None:
        outs = [torch.empty((n, p["out_dim"]), dtype=torch.float32, device=dev) for p in packs]
    arr = (_capi.DnMlpNet * len(packs))(*[_net_struct(p, o) for p, o in zip(packs, outs)])
    with torch.cuda.device(dev):
        _capi.check(_capi.load().dn_mlp_forward(C.cast(arr, C.c_void_p), len(packs), obs.data_ptr(),
                                                row_mask.data_ptr() if row_mask is not None else None, n, obs.shape[1],
                                                dev.index, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return outs


class FusedMlpPolicy:
    """policy(obs) -> (actions, values, log_probs) for RolloutCollector on the fused kernel, from an MlpActorCritic's
    weights (re-pack with `refresh()` after every optimiser step).  Static output buffers: hipGraph-capture safe."""

    def __init__(self, module, num_envs, device, grade="bf16"):
        self.module, self.device, self.grade = module, torch.device(device), grade
        self._mean = torch.empty((num_envs, module.action_net.out_features), dtype=torch.float32, device=self.device)
        self._value = torch.empty((num_envs, 1), dtype=torch.float32, device=self.device)
        self.pi = self.vf = None
        self.refresh()

    def refresh(self):
        """Re-pack the module's current weights into the SAME device tensors (captured graphs keep working)."""
        m = self.module
        lin = lambda seq: [l for l in seq if isinstance(l, torch.nn.Linear)]           # noqa: E731
        pi = [(l.weight, l.bias) for l in lin(m.pi)] + [(m.action_net.weight, m.action_net.bias)]
        vf = [(l.weight, l.bias) for l in lin(m.vf)] + [(m.value_net.weight, m.value_net.bias)]
        self.pi = _repack_into(self.pi, pack_mlp(pi, self.device, self.grade))
        self.vf = _repack_into(self.vf, pack_mlp(vf, self.device, self.grade))
        self.log_std = m.log_std.detach().to(self.device).float()
        self.log_std_host = [float(x) for x in m.log_std.detach().cpu().float()]     # dn_policy_sample takes it by value

    def __call__(self, obs, deterministic=False):
        mlp_forward([self.pi, self.vf], obs, [self._mean, self._value])
        mean = self._mean
        std = torch.exp(self.log_std)
        actions = mean if deterministic else mean + std * torch.randn_like(mean)
        z = (actions - mean) / std
        log_prob = (-0.5 * z * z - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        return actions, self._value.squeeze(-1), log_prob

    def predict_values(self, obs, out=None, row_mask=None):
        """V(obs); with row_mask (uint8 [N]) only tiles that contain a flagged drone are evaluated, the rest read 0."""
        out = self._value if out is None else out
        mlp_forward([self.vf], obs, [out], row_mask=row_mask)
        return out.squeeze(-1)


class FusedSacActor:
    """actor(obs) -> actions in (-1, 1) for OffPolicyCollector (BASELINE config 5) on the fused kernel, from a policy.SacActor's
    weights (re-pack with `refresh()` after optimiser steps): ONE launch gives mu and log_std of every drone; the clamp, the
    Gaussian draw and the tanh squash (a few element-wise ops on [N, 4]) stay in torch, float32."""

    def __init__(self, module, num_envs, device, grade="bf16"):
        self.module, self.device, self.grade = module, torch.device(device), grade
        self.act_dim = module.mu.out_features
        self._out = torch.empty((num_envs, 2 * self.act_dim), dtype=torch.float32, device=self.device)
        self.pack = None
        self.refresh()

    def refresh(self):
        """Re-pack the module's current weights into the SAME device tensors: OffPolicyCollector.collect_cycle's hipGraph
        holds their addresses, and SAC refreshes after every cycle."""
        m = self.module
        lin = [l for l in m.latent_pi if isinstance(l, torch.nn.Linear)]
        layers = [(l.weight, l.bias) for l in lin] + [(m.mu.weight, m.mu.bias), (m.log_std.weight, m.log_std.bias)]
        self.pack = _repack_into(self.pack, pack_sac_actor(layers, self.device, self.grade))

    def mean_log_std(self, obs):
        from .policy import LOG_STD_MAX, LOG_STD_MIN
        mlp_forward([self.pack], obs, [self._out])
        return self._out[:, :self.act_dim], torch.clamp(self._out[:, self.act_dim:], LOG_STD_MIN, LOG_STD_MAX)

    def __call__(self, obs, deterministic=False):
        mean, log_std = self.mean_log_std(obs)
        pre = mean if deterministic else mean + torch.exp(log_std) * torch.randn_like(mean)
        return torch.tanh(pre)
