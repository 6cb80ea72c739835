import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import pyoracle as oracle
from shifu_amd import _abi
from shifu_amd.backend import Sim, default_sim_params
from shifu_amd.gym.abb_fused import FusedAbbEnv
n = 512
env = FusedAbbEnv(num_envs=n, seed=5, link_shapes="hull", solver="pgs", face_manifold=True)
torch.cuda.synchronize()
e = 286
rng = np.random.default_rng(4)
raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
root0 = env.sim.tensors[_abi.T_ROOT_STATE].cpu().numpy().reshape(n, 4, 13)[e].copy()
dof0 = env.sim.tensors[_abi.T_DOF_STATE].cpu().numpy().reshape(n, -1, 2)[e].copy()
env.task.step(torch.from_numpy(raw).cuda())
torch.cuda.synchronize()
tgt = env.task.tensors[_abi.ABB_DOF_TARGETS].cpu().numpy().reshape(n, -1)[e].copy()
m = 8
# first the true trajectory (oracle, 8 contacts) up to sub-step 2
spo = env.sim_params
dof = np.tile(dof0.reshape(1, -1, 2), (m, 1, 1)).reshape(-1, 2).astype(np.float32)
root = np.tile(root0.reshape(1, 4, 13), (m, 1, 1)).reshape(-1, 13).astype(np.float32)
pt = np.tile(tgt, m).astype(np.float32)
with oracle.scene_extras(hulls=env.cm.hulls, flags=1):
    for it in range(2):
        oracle.scene_step(env.cm.blob, spo, env.boxes, m, dof, root, pos_target=pt, friction=np.ones(m, np.float32))
print("state before sub-step 2: cube", root.reshape(m, 4, 13)[0, 2])
for flags in (0, 1):
    for kmax in (1,):
        sp = default_sim_params(dt=0.02, solver="pgs", max_contacts=kmax)
        sim = Sim(sp, "cuda:0")
        sim.set_plane(1.0)
        sim.set_articulation(env.cm.blob)
        sim.set_hulls(env.cm.hulls)
        if flags:
            sim.set_scene_flags(flags)
        for b in env.boxes:
            sim.add_box(b)
        sim.finalize(m, 0, group=32)
        sim.tensors[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
        sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
        sim.set_dof_command(_abi.T_POS_TARGET, torch.from_numpy(pt).cuda())
        sim.step()
        torch.cuda.synchronize()
        d2, r2 = dof.copy(), root.copy()
        oracle.dropped(reset=True)
        with oracle.scene_extras(hulls=env.cm.hulls, flags=flags):
            oracle.scene_step(env.cm.blob, sp, env.boxes, m, d2, r2, pos_target=pt, friction=np.ones(m, np.float32))
        print("flags", flags, "kmax", kmax, "candidates gpu", sim.tensors[_abi.T_DROPPED].cpu().numpy()[:2] + kmax, "oracle", oracle.dropped() // m + kmax)
print("BITS", repr(root.reshape(m, 4, 13)[0].view(np.uint32).tolist()))
