"""Random sweep of the cut elements on the CPU: random tree mechanisms (tests/random_mechanisms.py) with a loop-closing
Spherical joint between two bodies, a free sphere that falls on a body, or a body-body contact between two bodies of the tree that are no
neighbours -- the device program under the emulator (lane mapping, general build) against the oracle: status, Newton iteration counts, states,
and for the loops the IFT Jacobians.  Usage: python tools/random_cut_sweep.py [first_seed] [count]"""
import os, sys, copy
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np
import dojo_amd as d
from oracle import Oracle
from emu_wrap import emu_step
from random_mechanisms import random_mechanism
from dojo_amd.mechanisms import BodySpec, Floating, Spherical, sphere_inertia, sphere_sphere_contact
from dojo_amd.quat import vrot, qconj


def with_cut(seed):
    rng = np.random.default_rng(1000 + seed)
    spec, z, u = random_mechanism(seed, nb=int(rng.integers(3, 8)), translational=bool(rng.random() < 0.4))
    spec = copy.deepcopy(spec)
    Z = z.reshape(-1, 13)
    parent = {j.child: j.parent for j in spec.joints}
    kind = ["loop", "free", "inner"][seed % 3]
    pairs = [(a, b) for a in range(spec.Nb) for b in range(spec.Nb) if a != b and parent[b] != a and parent[a] != b]
    if kind != "free" and not pairs:
        kind = "free"
    if kind == "loop":
        a, b = pairs[int(rng.integers(len(pairs)))]
        P = 0.5 * (Z[a, :3] + Z[b, :3])                      # one world point, seen from both bodies: the loop is closed at the start
        va = vrot(P - Z[a, :3], qconj(Z[a, 6:10])); vb = vrot(P - Z[b, :3], qconj(Z[b, 6:10]))
        j = Spherical("loop", a, b, va, vb, damper=float(rng.choice([0.0, 0.4])))
        j.loop = True
        spec.joints.append(j)
        u = np.concatenate([u, 0.3 * rng.normal(size=3)])
    elif kind == "free":
        a = int(rng.integers(spec.Nb)); r = 0.15
        spec.bodies.append(BodySpec("ball", 0.4, sphere_inertia(r, 0.4)))
        spec.joints.append(Floating("ball_free", -1, spec.Nb - 1))
        spec.contacts.append(sphere_sphere_contact("ball_on_body", a, spec.Nb - 1, r, r, 0.5, "nonlinear"))
        dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        zb = np.zeros(13); zb[6] = 1.0; zb[0:3] = Z[a, :3] + dirn * (2 * r + 0.02); zb[3:6] = Z[a, 3:6] - 1.0 * dirn; zb[10:13] = rng.normal(size=3)
        z = np.concatenate([z, zb]); u = np.concatenate([u, np.zeros(6)])
    else:
        a, b = pairs[int(rng.integers(len(pairs)))]
        dist = np.linalg.norm(Z[a, :3] - Z[b, :3])
        if dist < 0.05:
            return None
        r = 0.5 * (dist - 0.01)
        spec.contacts.append(sphere_sphere_contact("inner", a, b, r, r, 0.5, "nonlinear"))
    return kind, spec, z, u


def run(seed, steps=6, verbose=False):
    got = with_cut(seed)
    if got is None:
        return "skipped", 0.0
    kind, spec, z, u = got
    try:
        o = Oracle(spec)
    except Exception as e:
        return "refused by the oracle: %s" % e, 0.0
    worst = 0.0
    for k in range(steps):
        zo, info = o.step(z, u)
        try:
            r = emu_step(spec, z[None], u[None], quad=False, grad=(kind == "loop" and k == steps - 1))
        except RuntimeError as e:
            return "%s refused: %s" % (kind, e), 0.0
        if r["status"][0] != info["status"] or r["iters"][0] != info["iters"]:
            return "%s step %d: status %d/%d iterations %d/%d" % (kind, k, r["status"][0], info["status"], r["iters"][0], info["iters"]), worst
        if info["status"] == 0:
            e = float(np.abs(r["z_next"][0] - zo).max()); worst = max(worst, e)
            if e > 1e-7:
                return "%s step %d: state error %.2e (%d iterations)" % (kind, k, e, info["iters"]), worst
            if "dz" in r:
                _, _, _, dz_o, du_o = o.step_batch(z[None], u[None], with_grad=True, nthreads=1)
                eg = max(np.abs(r["dz"][0] - dz_o[0]).max() / max(1.0, np.abs(dz_o[0]).max()), np.abs(r["du"][0] - du_o[0]).max() / max(1.0, np.abs(du_o[0]).max()))
                worst = max(worst, eg)
                if eg > 1e-6:
                    return "%s step %d: Jacobian error %.2e" % (kind, k, eg), worst
        z = zo
    return "%s ok" % kind, worst


if __name__ == "__main__":
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    bad = 0
    for seed in range(s0, s0 + n):
        msg, w = run(seed)
        flag = "" if msg.endswith("ok") or msg == "skipped" else "   <<<<"
        bad += bool(flag)
        print("seed %4d: %-60s worst %.2e%s" % (seed, msg, w, flag), flush=True)
    print("%d of %d seeds off" % (bad, n))
