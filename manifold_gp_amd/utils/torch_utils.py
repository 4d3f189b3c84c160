"""bump_function (manifold_gp/utils/torch_utils.py:38-41): compact-support blend used by the
out-of-sample features and the hybrid posterior.  Pure elementwise torch on the caller's device
(the fused HIP kernel mgp_features_oos evaluates the same formula in place)."""
import torch


def bump_function(x, alpha, beta):
    alpha = torch.as_tensor(alpha, dtype=x.dtype, device=x.device)
    inside = x.abs() < alpha
    a2 = alpha.square()
    safe = torch.where(inside, x.square() - a2, -torch.ones_like(x))
    val = torch.exp(beta / safe) / torch.exp(-beta / a2)
    return torch.where(inside, val, torch.zeros_like(x))
