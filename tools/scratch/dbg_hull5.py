import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import pyoracle as oracle
from shifu_amd import _abi
from shifu_amd.backend import Sim
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
print("targets", tgt)
# the same env alone through the hook path, sub-step by sub-step
m = 8
sim = Sim(env.sim_params, "cuda:0")
sim.set_plane(1.0)
sim.set_articulation(env.cm.blob)
sim.set_hulls(env.cm.hulls)
sim.set_scene_flags(_abi.SCENE_FACE_MANIFOLD)
for b in env.boxes:
    sim.add_box(b)
sim.finalize(m, 0, group=32)
dof = np.tile(dof0.reshape(1, -1, 2), (m, 1, 1)).reshape(-1, 2).astype(np.float32)
root = np.tile(root0.reshape(1, 4, 13), (m, 1, 1)).reshape(-1, 13).astype(np.float32)
sim.tensors[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
pt = np.tile(tgt, m).astype(np.float32)
with oracle.scene_extras(hulls=env.cm.hulls, flags=_abi.SCENE_FACE_MANIFOLD):
    for it in range(6):
        sim.set_dof_command(_abi.T_POS_TARGET, torch.from_numpy(pt).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        oracle.dropped(reset=True)
        contact, bstate, _ = oracle.scene_step(env.cm.blob, env.sim_params, env.boxes, m, dof, root, pos_target=pt, friction=np.ones(m, np.float32))
        torch.cuda.synchronize()
        gc = sim.tensors[_abi.T_CONTACT].cpu().numpy().reshape(m, -1, 3)[0]
        oc = contact.reshape(m, -1, 3)[0]
        eq = np.array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root)
        print("substep", it, "root equal", eq, "dof equal", np.array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof), "drops gpu", sim.tensors[_abi.T_DROPPED].cpu().numpy()[:2], "oracle", oracle.dropped() // m)
        if not np.array_equal(gc, oc):
            print("contact gpu\n", gc, "\noracle\n", oc)
            print("cube root gpu", sim.tensors[_abi.T_ROOT_STATE].cpu().numpy().reshape(m, 4, 13)[0, 2], "\noracle", root.reshape(m, 4, 13)[0, 2])
            break
