import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import pyoracle as oracle
from shifu_amd import _abi
from shifu_amd.gym.abb_fused import FusedAbbEnv
from tests.test_gpu_parity import _ABB_SIM_T, _ABB_T
n = 512
env = FusedAbbEnv(num_envs=n, seed=5, link_shapes="hull", solver="pgs")
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
oracle.dropped(reset=True)
with oracle.scene_extras(hulls=env.cm.hulls, flags=_abi.SCENE_FACE_MANIFOLD):
    for it in range(40):
        raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
        env.task.step(torch.from_numpy(raw).cuda())
        oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
        torch.cuda.synchronize()
        bad = set()
        for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
            got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
            if not np.array_equal(got, bufs[k]):
                d = (got != bufs[k]).reshape(n, -1).any(1) if got.size % n == 0 else None
                envs = np.nonzero(d)[0] if d is not None else []
                print("step", it, "tensor", k, "envs", list(envs)[:10])
                bad |= set(envs)
        if bad:
            e = sorted(bad)[0]
            print("dropped gpu", env.sim.tensors[_abi.T_DROPPED].cpu().numpy()[sorted(bad)], "oracle total", oracle.dropped(reset=False))
            print("contact gpu", env.sim.tensors[_abi.T_CONTACT].cpu().numpy().reshape(n, -1, 3)[e])
            print("contact ora", bufs["contact"].reshape(n, -1, 3)[e])
            print("dof gpu", env.sim.tensors[_abi.T_DOF_STATE].cpu().numpy().reshape(n, -1, 2)[e])
            print("dof ora", bufs["dof_state"].reshape(n, -1, 2)[e])
            break
print("dropped total gpu", int(env.sim.tensors[_abi.T_DROPPED].sum()), "oracle", oracle.dropped())
