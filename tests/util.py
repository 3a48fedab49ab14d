"""Shared helpers for the tests (scene builders, tolerances)."""
import math

import numpy as np
import torch

from splatco_amd.cameras import look_at_camera, make_camera
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians


def oracle_settings(orc, cam, bg, scale_modifier=1.0, sh_degree=1):
    return orc.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5),
                        math.tan(cam.FoVy * 0.5), np.asarray(bg, np.float32), scale_modifier,
                        cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), sh_degree,
                        cam.camera_center.numpy())


def small_scene(P=96, W=64, H=48, seed=3, spread=1.0):
    """A small random scene seen by an off-axis camera (exercises every matrix entry)."""
    rng = np.random.default_rng(seed)
    cam = look_at_camera(eye=(0.6, -0.4, -4.0), target=(0.1, 0.05, 0.0), up=(0.05, -1.0, 0.1),
                         FoVx=math.radians(55.0), width=W, height=H)
    means = rng.uniform(-1.2, 1.2, (P, 3)) * spread
    scales = np.exp(rng.uniform(math.log(0.03), math.log(0.35), (P, 3)))
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.8, 1.2, (P, 1))       # the operator does not normalise quaternions
    op = rng.uniform(0.05, 0.95, (P, 1))
    col = rng.uniform(0, 1, (P, 3))
    f = np.float32
    return cam, dict(means3D=means.astype(f), scales=scales.astype(f), rotations=q.astype(f),
                     opacities=op.astype(f), colors=col.astype(f),
                     bg=np.array([0.2, 0.5, 0.9], f))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def psnr(a, b):
    """utils/image_utils.py:17-19 (mean over channels of per-channel PSNR)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return float(np.mean(20 * np.log10(1.0 / np.sqrt(np.maximum(mse, 1e-300)))))


def stress_scene(rng):
    """One randomised scene of the stress set: image sizes 17..700 px, 1..30 k Gaussians, random camera, anisotropy
    up to 300:1, optional dense clump (thousands of instances in one tile -> workgroup sort / merge passes), skewed
    opacity distributions, scale modifiers.  Returns (camera, gaussians, scale_modifier)."""
    W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
    P = int(rng.integers(1, 30000))
    eye = rng.uniform(-1, 1, 3) + np.array([0, 0, -rng.uniform(2.5, 7)])
    cam = look_at_camera(eye, rng.uniform(-0.3, 0.3, 3), (rng.uniform(-0.2, 0.2), -1.0, rng.uniform(-0.2, 0.2)),
                         math.radians(rng.uniform(30, 100)), W, H)
    spread = rng.uniform(0.3, 3.0)
    means = rng.normal(0, spread, (P, 3))
    if rng.random() < 0.3:                       # a dense clump -> large tiles / merge passes
        k = P // 2
        means[:k] = rng.normal(0, 0.03, (k, 3)) + rng.uniform(-0.5, 0.5, 3)
    smax = rng.choice([0.02, 0.1, 0.6, 3.0])
    scales = np.exp(rng.uniform(math.log(smax / 300), math.log(smax), (P, 3)))   # extreme anisotropy included
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.7, 1.3, (P, 1))
    op = rng.uniform(0.0, 1.0, (P, 1)) ** rng.choice([0.3, 1.0, 3.0])
    col = rng.uniform(0, 1, (P, 3))
    f = np.float32
    g = dict(means3D=means.astype(f), scales=scales.astype(f), rotations=q.astype(f), opacities=op.astype(f),
             colors=col.astype(f), bg=rng.uniform(0, 1, 3).astype(f))
    return cam, g, float(rng.choice([0.5, 1.0, 1.7]))


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(script, argv, world, tmp_path, timeout=1500, env_extra=None):
    """Start `world` ranks of `script` as plain child processes (RANK / WORLD_SIZE / MASTER_* in the environment, a free
    rendezvous port, no elastic launcher in between), each with its own stdout / stderr file under tmp_path.  Returns
    (ok, message): on failure the message ENDS with the first failing rank's own stderr tail -- the one thing a truncated
    log must still show."""
    import os
    import signal
    import subprocess
    import sys
    import time
    port = free_port()
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra or {})
        out, err = open(tmp_path / f"rank{r}.out", "w"), open(tmp_path / f"rank{r}.err", "w")
        files += [out, err]
        procs.append(subprocess.Popen([sys.executable, str(script)] + [str(a) for a in argv], stdout=out, stderr=err, env=env,
                                      start_new_session=True))
    deadline, first_bad, timed_out = time.time() + timeout, None, False
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and first_bad is None:
            first_bad = bad[0]
            deadline = min(deadline, time.time() + 20)      # the peers of a dead rank block in their next collective
        if all(c is not None for c in codes):
            break
        if time.time() > deadline:
            timed_out = first_bad is None
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)     # exactly the session started above
                    except ProcessLookupError:
                        pass
            for p in procs:
                p.wait()
            break
        time.sleep(0.2)
    for f in files:
        f.close()
    outs = [(tmp_path / f"rank{r}.out").read_text() for r in range(world)]
    if first_bad is None and not timed_out:
        return True, "\n".join(outs)
    errs = [(tmp_path / f"rank{r}.err").read_text() for r in range(world)]
    who = 0 if first_bad is None else first_bad
    msg = [f"exit codes {[p.returncode for p in procs]}" + (f", timed out after {timeout} s" if timed_out else "")]
    msg += [f"--- rank {r} stdout tail ---\n{o[-(4000 if r == who else 300):]}" for r, o in enumerate(outs)]
    msg += [f"--- rank {who} (first to fail) stderr tail ---\n{errs[who][-2500:]}"]
    return False, "\n".join(msg)
