#!/usr/bin/env python3
"""Train a Simplex-GP on a synthetic regression task on one MI355X, the way
experiments/train_simplexgp.py trains on UCI data (which is not redistributable
and not in this repo): standardised inputs, 64/16/20 split, Adam(lr=0.1) on the
CG/SLQ marginal likelihood, validation-RMSE early stopping, best state saved.

  python examples/train_synthetic.py --n 100000 --d 8 --epochs 30 [--nu 1.5 --order 3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx  # noqa: E402
from simplex_gp_amd import solvers, training  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--nu", type=float, default=None, help="Matern smoothness (1.5 / 2.5); default RBF")
    ap.add_argument("--min-noise", type=float, default=1e-4)
    ap.add_argument("--pre-size", type=int, default=100, help="rank of the pivoted-Cholesky preconditioner (train_simplexgp.py:88 default; 0 = plain CG)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="model.pt")
    args = ap.parse_args()

    g = torch.Generator().manual_seed(args.seed)
    x = torch.randn(args.n, args.d, generator=g)
    w = torch.randn(args.d, generator=g) / args.d ** 0.5
    y = torch.sin(x @ w * 2) + 0.5 * torch.cos(x[:, 0]) + 0.1 * torch.randn(args.n, generator=g)
    x = (x - x.mean(0)) / x.std(0)                      # experiments/utils.py:34-44 standardises too
    y = (y - y.mean()) / y.std()
    n_tr, n_va = int(0.64 * args.n), int(0.16 * args.n)
    dev = torch.device("cuda", 0)
    tr = (x[:n_tr].to(dev), y[:n_tr].to(dev))
    va = (x[n_tr:n_tr + n_va].to(dev), y[n_tr:n_tr + n_va].to(dev))
    te = (x[n_tr + n_va:].to(dev), y[n_tr + n_va:].to(dev))

    kernel = (plx.MaternLattice(nu=args.nu, order=args.order, ard_num_dims=args.d) if args.nu is not None
              else plx.RBFLattice(order=args.order, ard_num_dims=args.d))
    model = solvers.LatticeGP(kernel, min_noise=args.min_noise).to(dev)
    t0 = time.perf_counter()
    history, best = training.fit(model, tr, val=va, test=te, epochs=args.epochs, lr=args.lr, pre_size=args.pre_size, checkpoint=args.out, cap_host_threads=True,
                                 log=lambda row: print(json.dumps({k: round(v, 4) if isinstance(v, float) else v
                                                                   for k, v in row.items()}), flush=True))
    torch.cuda.synchronize()
    print(json.dumps({"seconds": round(time.perf_counter() - t0, 2), "best": best["summary"], "saved": args.out}))


if __name__ == "__main__":
    main()
