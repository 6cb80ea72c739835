import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import pyoracle as oracle
from shifu_amd import _abi
from shifu_amd.gym.abb_fused import FusedAbbEnv
from tests.test_gpu_parity import _ABB_SIM_T, _ABB_T
n = 512
for fm in (True, False):
    env = FusedAbbEnv(num_envs=n, seed=5, link_shapes="hull", solver="pgs", face_manifold=fm)
    dof = env.sim.tensors[_abi.T_DOF_STATE].view(n, -1, 2)
    g = torch.Generator().manual_seed(1)
    bend = torch.rand(n, generator=g)
    dof[::3, 1, 0] += (0.5 + 0.4 * bend[::3]).cuda()
    dof[::3, 2, 0] += (0.3 * bend[::3]).cuda()
    env.sim.tensors[_abi.T_SIM_DOF].copy_(env.sim.tensors[_abi.T_DOF_STATE])
    env.task.tensors[_abi.ABB_EP_LEN].copy_(torch.randint(150, 201, (n,)))
    torch.cuda.synchronize()
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(4)
    raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
    env.task.step(torch.from_numpy(raw).cuda())
    torch.cuda.synchronize()
    gd = env.sim.tensors[_abi.T_DROPPED].cpu().numpy()
    od = np.zeros(n, np.int64)
    shapes = {"rew_sums": (2, n), "done_sums": (4, n)}
    with oracle.scene_extras(hulls=env.cm.hulls, flags=_abi.SCENE_FACE_MANIFOLD if fm else 0):
        for e in range(n):
            be = {}
            for k, v in bufs.items():
                if k in shapes:
                    be[k] = np.ascontiguousarray(v.reshape(shapes[k][0], n)[:, e:e + 1])
                else:
                    per = v.shape[0] // n
                    be[k] = np.ascontiguousarray(v[e * per:(e + 1) * per])
            oracle.dropped(reset=True)
            oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, 1, e, be, raw[e:e + 1])
            od[e] = oracle.dropped(reset=True)
            got = env.sim.tensors[_abi.T_ROOT_STATE].cpu().numpy().reshape(n, -1)[e]
            if not np.array_equal(got, be["root_state"].reshape(-1)) or od[e] != gd[e]:
                print("fm", fm, "env", e, "drops gpu/oracle", gd[e], od[e], "root equal", np.array_equal(got, be["root_state"].reshape(-1)),
                      "dof", bufs["dof_state"].reshape(n, -1, 2)[e, :, 0])
    print("fm", fm, "total drops", gd.sum(), od.sum())
    env.destroy()
