"""Quaternion / IK helpers (reference shifu/utils/torch_utils.py:4-58), xyzw order."""
import torch

from shifu_amd.isaacgym.torch_utils import quat_conjugate, quat_mul  # same formulas (:12-40)


def free_tensor_attrs(obj):
    """Drop tensor attributes so device memory can be reclaimed (:4-9)."""
    for name in list(vars(obj).keys()):
        if isinstance(getattr(obj, name, None), torch.Tensor):
            delattr(obj, name)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def orientation_error(desired, current):
    q_r = quat_mul(desired, quat_conjugate(current))
    return q_r[:, 0:3] * torch.sign(q_r[:, 3]).unsqueeze(-1)


def inverse_kinematics(dof_pos, ee_pos, ee_quat, tar_pos, tar_quat, j_ee, device, damping=0.05):
    """Damped least squares: dq = J^T (J J^T + lambda^2 I)^-1 dpose (:43-58)."""
    dpose = torch.cat([tar_pos - ee_pos, orientation_error(tar_quat, ee_quat)], -1).unsqueeze(-1)
    jt = torch.transpose(j_ee, 1, 2)
    lam = torch.eye(6, device=device) * (damping ** 2)
    u = (jt @ torch.inverse(j_ee @ jt + lam) @ dpose).view(dof_pos.shape[0], dof_pos.shape[1])
    return dof_pos + u
