"""Launches per CG iteration of solvers.khat_solve, counted from a HIP-graph capture (bench.cg_launch_leg), and the device still
works afterwards.  python tests/checks/launch_count_check.py"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
class Ctx: dev=torch.device('cuda:0')
print(bench.cg_launch_leg(Ctx, n=200000, d=8))
x=torch.randn(10,device='cuda'); print(float(x.sum()))   # the device still works
