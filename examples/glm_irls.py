"""A glum-style GLM fit on tabmat_amd blocks: Poisson regression (log link, ridge penalty) by iteratively
reweighted least squares.  Every pass over the design is one of the three products this package accelerates --

    eta = X @ beta                         SplitMatrix.matvec
    H   = X' diag(w) X                     SplitMatrix.sandwich            (the per-iteration hot spot)
    g   = X' (w * z)                       SplitMatrix.transpose_matvec

-- with device vectors in and device results out (no host traffic inside the loop); the p x p solve is the only
thing left to torch.  The same code runs on the reference by swapping the import and dropping the torch tensors.

    python examples/glm_irls.py [rows]     # BASELINE configs[3] design: dense 128 + sparse 512 @ 5 % + 3 categoricals
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fit_poisson(X, y, alpha: float = 1.0, iters: int = 8, beta0=None, callback=None):
    """X: any tabmat_amd matrix (n, p); y: device tensor of counts (n,).  Minimises the Poisson deviance
    + alpha / 2 * |beta|^2 with IRLS; returns beta as a device tensor of X's dtype."""
    n, p = X.shape
    dt = y.dtype
    beta = torch.zeros(p, dtype=dt, device=y.device) if beta0 is None else beta0.clone()
    eye = torch.eye(p, dtype=torch.float64, device=y.device)
    for it in range(iters):
        eta = X.matvec(beta)                              # (n,) on the device
        mu = torch.exp(eta.clamp(max=30.0))
        w = mu                                            # IRLS weights of the log link
        z = eta + (y - mu) / mu                           # working response
        H = X.sandwich(w)                                 # (p, p) float64 on the device
        g = X.transpose_matvec(w * z)                     # (p,)
        beta_new = torch.linalg.solve(H + alpha * eye, g.to(torch.float64)).to(dt)
        step = float((beta_new - beta).abs().max())
        beta = beta_new
        if callback is not None:
            callback(it, beta, step)
        if step < 1e-10:
            break
    return beta


def _numpy_reference(E, y, alpha, iters):
    beta = np.zeros(E.shape[1])
    for _ in range(iters):
        eta = E @ beta
        mu = np.exp(np.minimum(eta, 30.0))
        z = eta + (y - mu) / mu
        H = E.T @ (mu[:, None] * E)
        beta_new = np.linalg.solve(H + alpha * np.eye(E.shape[1]), E.T @ (mu * z))
        done = np.abs(beta_new - beta).max() < 1e-10
        beta = beta_new
        if done:
            break
    return beta


def main():
    from tabmat_amd import synth

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
    t0 = time.perf_counter()
    X.to_device()
    torch.cuda.synchronize()
    print(f"design {X.shape}: twins built in {(time.perf_counter() - t0) * 1e3:.0f} ms", flush=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    truth = torch.randn(X.shape[1], dtype=torch.float64, device="cuda", generator=g) * 0.02
    y = torch.poisson(torch.exp(X.matvec(truth)), generator=g)
    ts = []

    def cb(it, beta, step):
        torch.cuda.synchronize()
        ts.append(time.perf_counter())
        print(f"  iteration {it}: max |delta beta| = {step:.3e}", flush=True)

    torch.cuda.synchronize()
    ts.append(time.perf_counter())
    beta = fit_poisson(X, y, alpha=1.0, iters=8, callback=cb)
    per = np.diff(ts) * 1e3
    print(f"IRLS: {len(per)} iterations, {per[1:].mean() if len(per) > 1 else per[0]:.1f} ms per iteration "
          f"(matvec + sandwich + transpose_matvec + {X.shape[1]} x {X.shape[1]} solve); "
          f"max |beta - truth| = {float((beta - truth).abs().max()):.3e}")


if __name__ == "__main__":
    main()
