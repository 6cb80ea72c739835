import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# a debug build of the oracle with the fallback trace
subprocess.check_call("cd %s/oracle && gcc -O2 -fPIC -std=c11 -mavx2 -mfma -ffp-contract=off -fno-fast-math -fopenmp -DSHF_MANIFOLD_DEBUG -DSHF_REAL_DOUBLE=0 -c shf_oracle.c -o /tmp/o32.o && gcc -O2 -fPIC -std=c11 -mavx2 -mfma -ffp-contract=off -fno-fast-math -fopenmp -DSHF_REAL_DOUBLE=1 -c shf_oracle.c -o /tmp/o64.o && gcc -shared -fopenmp -o /tmp/liborc_dbg.so /tmp/o32.o /tmp/o64.o -lm" % ROOT, shell=True)
os.environ["SHF_ORACLE_LIB"] = "/tmp/liborc_dbg.so"
import numpy as np, torch
from oracle import pyoracle as oracle
from shifu_amd import _abi
from shifu_amd.gym.abb_fused import FusedAbbEnv
from tests.test_gpu_parity import _ABB_SIM_T, _ABB_T
n = 512
env = FusedAbbEnv(num_envs=n, seed=5, link_shapes="hull", solver="pgs", face_manifold=True)
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
e = 286
shapes = {"rew_sums": (2, n), "done_sums": (4, n)}
be = {}
for k, v in bufs.items():
    if k in shapes:
        be[k] = np.ascontiguousarray(v.reshape(shapes[k][0], n)[:, e:e + 1])
    else:
        per = v.shape[0] // n
        be[k] = np.ascontiguousarray(v[e * per:(e + 1) * per])
with oracle.scene_extras(hulls=env.cm.hulls, flags=_abi.SCENE_FACE_MANIFOLD):
    oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, 1, e, be, raw[e:e + 1])
print("root after", be["root_state"].reshape(4, 13)[2])
