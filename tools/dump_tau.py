#!/usr/bin/env python3
"""diagnostic: python3 tools/dump_tau.py --lib X.so --out f.npy kind:cfg:n   -> tau, metrics, status of one launch"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--lib"); ap.add_argument("--out"); ap.add_argument("case")
a = ap.parse_args()
os.environ["WBC_HIP_LIB"] = os.path.abspath(a.lib)
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
kind, cfg, n = a.case.split(":"); cfg = int(cfg); n = int(n)
b = workloads.make_batch(cfg, n=n)
ctrl = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind](model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
tau, met, st = ctrl.step(*[up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")])
ctrl.sync()
np.save(a.out, np.concatenate([tau.cpu().numpy(), met.cpu().numpy(), st.cpu().numpy()[None].astype(float)]))
