import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shifu_amd import _abi
from shifu_amd.gym.abb_fused import FusedAbbEnv
n = 512
env = FusedAbbEnv(num_envs=n, seed=5, link_shapes="hull", solver="pgs", face_manifold=True)
torch.cuda.synchronize()
r = env.sim.tensors[_abi.T_ROOT_STATE].cpu().numpy().reshape(n, 4, 13)[286]
print(repr(r.view(np.uint32).tolist()))
