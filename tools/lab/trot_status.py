"""diagnostic (GPU box): step the trot closed loop launch-per-stage and capture every tick whose status is non-zero"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import MPTCController, workloads
from quadruped_drake_amd.trajectory import TrunkTrajectory
n, steps = 4096, 520
K = 4000
ts = np.arange(K) * 1e-3
st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
tg = np.tile(st_t, (K, 1))
tg[:, 0] += 0.01 * np.sin(2 * np.pi * ts / 0.3); tg[:, 3] = 0.01 * 2 * np.pi / 0.3 * np.cos(2 * np.pi * ts / 0.3)
masks = np.where((np.arange(K) // 150) % 2 == 0, 0b1001, 0b0110).astype(np.uint8)
for f in range(4):
    sw = ((masks >> f) & 1) == 0
    tg[sw, 18 + 9 * f + 2] += 0.02
traj = TrunkTrajectory(ts, tg, masks, wait_time=0.0, device=0, standing_targets=st_t, standing_mask=0b1111)
q0, v0 = workloads.nominal_state("mini_cheetah", n)
rng = np.random.default_rng(1)
q0[7:] += rng.uniform(-0.03, 0.03, (12, n))
ctrl = MPTCController(max_batch=n, device=0)
dev = "cuda:0"
q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(rng.uniform(0.0, 0.6, n), device=dev)
vd = torch.zeros((18, n), dtype=torch.float64, device=dev)
ctrl.set_vdot_output(vd)
caps = []
for s in range(steps):
    tgt, mk = traj.lookup(t)
    tau, met, st = ctrl.step(q, v, tgt, mk); ctrl.sync()
    bad = torch.nonzero(st != 0).flatten()
    if bad.numel():
        b = bad.cpu().numpy()
        caps.append(dict(step=s, idx=b, st=st[bad].cpu().numpy(), q=q[:, bad].cpu().numpy(), v=v[:, bad].cpu().numpy(),
                         tg=tgt[:, bad].cpu().numpy(), mk=mk[bad].cpu().numpy(), tau=tau[:, bad].cpu().numpy()))
    ctrl.integrate(q, v, vd, 1e-3); t += 1e-3
print("captured", sum(len(c["idx"]) for c in caps), "ticks with non-zero status; statuses", np.bincount(np.concatenate([c["st"] for c in caps])) if caps else None)
if caps:
    np.savez("gpurun_out/trot_status.npz", q=np.concatenate([c["q"] for c in caps], 1), v=np.concatenate([c["v"] for c in caps], 1),
             tg=np.concatenate([c["tg"] for c in caps], 1), mk=np.concatenate([c["mk"] for c in caps]), st=np.concatenate([c["st"] for c in caps]),
             step=np.concatenate([np.full(len(c["idx"]), c["step"]) for c in caps]), idx=np.concatenate([c["idx"] for c in caps]))
    for c in caps[:6]:
        print(c["step"], c["idx"], c["st"], "knees", c["q"][[9, 12, 15, 18]].T)
