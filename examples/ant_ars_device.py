"""Augmented random search on the Ant with every rollout on the GPU -- the loop of the reference's
examples/learning/ant_ars.jl (rollout_policy :78-115, training :118-190), batched: the 2 x n_directions perturbed
policies of one ARS iteration are the B environments of one `BatchedEnvironment`, and observation -> normalisation ->
linear policy -> step -> reward stay on the device (dojo_step_minimal_dev + dojo_observe_dev, torch for the policy).

    python examples/ant_ars_device.py --iterations 5 --directions 256 --horizon 100
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))


class Normalizer:
    """Running mean / variance of the observations (examples/learning/ars.jl `Normalizer`, `observe!`, `normalize`),
    updated with one batch of observations at a time (Chan et al. parallel update)."""

    def __init__(self, n, dtype, device):
        import torch
        self.n = torch.zeros((), dtype=dtype, device=device)
        self.mean = torch.zeros(n, dtype=dtype, device=device)
        self.m2 = torch.zeros(n, dtype=dtype, device=device)

    def observe(self, obs):
        b = obs.shape[0]
        bm = obs.mean(dim=0); bm2 = ((obs - bm) ** 2).sum(dim=0)
        tot = self.n + b
        delta = bm - self.mean
        self.mean = self.mean + delta * (b / tot)
        self.m2 = self.m2 + bm2 + delta ** 2 * (self.n * b / tot)
        self.n = tot

    def normalize(self, obs):
        var = (self.m2 / self.n.clamp(min=1.0)).clamp(min=1e-2)
        return (obs - self.mean) / var.sqrt()


def rollout_policy(theta, env, normalizer, horizon, observe=True):
    """rollout_policy (ant_ars.jl:78-115) for B policies at once: theta [B, n_actions, n_obs] -> rewards [B].
    An environment that fails the reference's health check (finite state, 0.2 <= z <= 1) stops collecting reward."""
    import torch
    env.initialize()
    B = env.batch
    rewards = torch.zeros(B, dtype=env.torch_dtype, device=env.device)
    alive = torch.ones(B, dtype=torch.bool, device=env.device)
    dt = env.spec.timestep
    nx = env.nx
    for k in range(horizon):
        state = env.get_state()
        x = state[:, :nx]
        if observe:
            normalizer.observe(state[alive] if k else state)
        action = torch.bmm(theta, normalizer.normalize(state).unsqueeze(2)).squeeze(2)
        env.step(x, action)
        after = env.get_state()
        forward_reward = 100.0 * (after[:, 0] - x[:, 0]) / dt
        control_cost = 0.05 / 10.0 * (action * action).sum(dim=1)
        contact_cost = 0.5e-3 * (after[:, nx:] ** 2).sum(dim=1)           # the observation already holds clamp(impulse, -1, 1)
        reward = forward_reward - control_cost - contact_cost + 0.05
        rewards = rewards + torch.where(alive, reward, torch.zeros_like(reward))
        ok = torch.isfinite(after).all(dim=1) & (after[:, 2] >= 0.2) & (after[:, 2] <= 1.0)
        alive = alive & ok
    return rewards


def train(iterations=5, directions=256, top=32, horizon=100, step_size=0.02, noise=0.03, dtype="f32", seed=0, log=print):
    """ARS-V2 update (examples/learning/ars.jl `train`): theta += step / (top * sigma_R) * sum_top (R+ - R-) delta."""
    import torch
    from dojo_amd.envs import BatchedEnvironment
    B = 2 * directions
    env = BatchedEnvironment("ant_ars", B, dtype=dtype)
    gen = torch.Generator(device=env.device); gen.manual_seed(seed)
    na, nobs = env.spec.nu - env.n_unactuated, env.nobs
    theta = torch.zeros(na, nobs, dtype=env.torch_dtype, device=env.device)
    normalizer = Normalizer(nobs, env.torch_dtype, env.device)
    history = []
    for it in range(iterations):
        t0 = time.time()
        delta = torch.randn(directions, na, nobs, dtype=env.torch_dtype, device=env.device, generator=gen)
        thetas = torch.cat([theta + noise * delta, theta - noise * delta], dim=0)
        R = rollout_policy(thetas, env, normalizer, horizon)
        Rp, Rm = R[:directions], R[directions:]
        order = torch.argsort(torch.maximum(Rp, Rm), descending=True)[:top]
        sigma = R.std().clamp(min=1e-6)                               # σ_r = std(rewards) over all rollouts (ant_ars.jl:152)
        theta = theta + step_size / (top * sigma) * ((Rp[order] - Rm[order]).view(-1, 1, 1) * delta[order]).sum(dim=0)
        torch.cuda.synchronize()
        secs = time.time() - t0
        history.append(float(R.mean()))
        log("iteration %d: mean reward %.2f  best %.2f  (%d env-steps in %.2f s = %.0f steps/s)" % (it, R.mean(), R.max(), B * horizon, secs, B * horizon / secs))
    env.close()
    return theta, history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--directions", type=int, default=256)
    ap.add_argument("--top", type=int, default=32)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    train(a.iterations, a.directions, a.top, a.horizon, dtype=a.dtype)
