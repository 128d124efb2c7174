"""Sampling-based gait search for the quadruped with all rollouts on the GPU -- the loop of the reference's
examples/learning/quadruped_sampling.jl (PD controller on a parametrised leg trajectory :21-55, rollout :69-80, random
search over the five gait parameters :83-127), batched: every iteration evaluates `candidates` perturbed parameter sets at
once, one environment each, instead of one after the other; observation -> controller -> step stay on the device.

    python examples/quadruped_sampling_device.py --iterations 5 --candidates 512 --horizon 2000
"""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))

KP = (100.0, 80.0, 60.0)
KD = (5.0, 4.0, 3.0)


def controller(x, k, params):
    """controller!(x, k) (quadruped_sampling.jl:25-55) for B environments with their own parameters [B, 5] = [freq, amp2, off2, amp3, off3]:
    PD control of every leg joint towards legmovement(k, a, b, c, offset) = a cos(k b 0.01 2π + offset) + c."""
    import torch
    B = x.shape[0]
    th = x[:, 12:36].reshape(B, 4, 3, 2)                         # legs FR, FL, RR, RL x (hip, thigh, calf) x (θ, dθ)
    phase = (k * 0.01 * 2 * math.pi) * params[:, 0:1]             # [B, 1]
    off2 = torch.tensor([0.0, math.pi, math.pi, 0.0], dtype=x.dtype, device=x.device)              # legs 1, 4 | 2, 3
    off3 = torch.tensor([-math.pi / 2, math.pi / 2, math.pi / 2, -math.pi / 2], dtype=x.dtype, device=x.device)
    target = torch.stack([torch.zeros(B, 4, dtype=x.dtype, device=x.device),
                          params[:, 1:2] * torch.cos(phase + off2) + params[:, 2:3],
                          params[:, 3:4] * torch.cos(phase + off3) + params[:, 4:5]], dim=2)       # [B, 4, 3]
    kp = torch.tensor(KP, dtype=x.dtype, device=x.device); kd = torch.tensor(KD, dtype=x.dtype, device=x.device)
    return (kp * (target - th[..., 0]) - kd * th[..., 1]).reshape(B, 12)


def standing_states(spec, params):
    """reset_state! (quadruped_sampling.jl:58-67): legs at the parameters' offsets, trunk height such that the feet touch the ground."""
    from dojo_amd import coords
    from dojo_amd.quat import vrot
    X = []
    feet = [c for c in spec.contacts if c.name.endswith("_calf_contact")]
    for p in params:
        x = coords.nominal_minimal(spec, body_position=[0, 0, -0.43], thigh_angle=float(p[2]), calf_angle=float(p[4]))
        z = coords.minimal_to_maximal(spec, x)
        low = min(z[13 * c.body + 2] + vrot(c.origin, z[13 * c.body + 6:13 * c.body + 10])[2] - c.radius for c in feet)
        x[2] -= low
        X.append(x)
    return np.stack(X)


def rollout(env, params, horizon):
    """rollout(env) (quadruped_sampling.jl:69-80) for all candidates: distance travelled along x; a candidate that falls
    (trunk below the ground plane), produces non-finite states or runs away counts as failed (-inf)."""
    import torch
    x0 = standing_states(env.spec, params.cpu().numpy())
    env.initialize(x0=x0)
    x = env.get_state()
    start = x[:, 0].clone()
    failed = torch.zeros(env.batch, dtype=torch.bool, device=env.device)
    for k in range(1, horizon + 1):
        failed |= (x[:, 2] < 0) | ~torch.isfinite(x).all(dim=1) | (x[:, 0].abs() > 1000)
        u = controller(torch.nan_to_num(x), float(k), params)
        env.step(torch.nan_to_num(x), u)
        x = env.get_state()
    dist = x[:, 0] - start
    return torch.where(failed | ~torch.isfinite(dist), torch.full_like(dist, -float("inf")), dist)


def search(iterations=5, candidates=512, horizon=2000, dtype="f32", seed=1, log=print):
    import torch
    from dojo_amd.envs import BatchedEnvironment
    env = BatchedEnvironment("quadruped_sampling", candidates, dtype=dtype, timestep=0.001, joint_limits={}, gravity=-9.81, contact_body=False)
    gen = torch.Generator(device=env.device); gen.manual_seed(seed)
    best = torch.tensor([0.1, 0.0, 1.0, 0.0, -1.5], dtype=env.torch_dtype, device=env.device)      # paramcontainer, quadruped_sampling.jl:12
    best_dist, explore = 0.0, 0.1
    for it in range(iterations):
        t0 = time.time()
        params = best + explore * torch.randn(candidates, 5, dtype=env.torch_dtype, device=env.device, generator=gen)
        params[0] = best
        dist = rollout(env, params, horizon)
        i = int(torch.argmax(dist))
        torch.cuda.synchronize()
        secs = time.time() - t0
        if float(dist[i]) > best_dist:
            best, best_dist, explore = params[i].clone(), float(dist[i]), 0.1
        else:
            explore *= 0.9
        log("iteration %d: best distance %.3f m (this batch %.3f, %d of %d candidates upright); %d env-steps in %.1f s = %.0f steps/s"
            % (it, best_dist, float(dist[i]), int(torch.isfinite(dist).sum()), candidates, candidates * horizon, secs, candidates * horizon / secs))
    env.close()
    return best.cpu().numpy(), best_dist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--candidates", type=int, default=512)
    ap.add_argument("--horizon", type=int, default=2000)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    print(search(a.iterations, a.candidates, a.horizon, a.dtype))
