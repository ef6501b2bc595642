"""Mirror of the quaternion helpers the hot path and its caller use
(dpc/util/quaternion.py:22-117), on torch tensors, (w,x,y,z) order.

On the hot path the rotation is NOT executed by these functions: it is folded
into the HIP point kernels (k_points_fwd / k_points_bwd).  They exist so that
a caller written against the reference finds the same names."""
import torch
import torch.nn.functional as F


def validate_shape(x):
    if x.shape[-1] != 4:
        raise ValueError("Can't create a quaternion from a tensor with shape {}."
                         "The last dimension must be 4.".format(tuple(x.shape)))


def vector3d_to_quaternion(x):
    x = torch.as_tensor(x)
    if x.shape[-1] != 3:
        raise ValueError("The last dimension of x must be 3.")
    return F.pad(x, (1, 0))


def _prepare(x):
    x = torch.as_tensor(x)
    if x.shape[-1] == 3:
        x = vector3d_to_quaternion(x)
    validate_shape(x)
    return x


def quaternion_multiply(a, b):
    a, b = _prepare(a), _prepare(b)
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2), dim=-1)


def quaternion_conjugate(q):
    # (no constant tensor from a Python list here: that is a synchronous host-to-device copy on every call,
    #  and not capturable into a HIP graph)
    return torch.cat([q[..., :1], -q[..., 1:]], dim=-1)


def quaternion_normalise(q):
    return q / q.norm(dim=-1, keepdim=True)


def quaternion_rotate(pc, q, inverse=False):
    """pc [B,N,3], q [B,4] (normalised here) -> q * pc * q' [B,N,3]."""
    q = q / q.norm(dim=-1, keepdim=True)
    q = q.unsqueeze(1)
    q_ = quaternion_conjugate(q)
    if not inverse:
        wxyz = quaternion_multiply(quaternion_multiply(q, pc), q_)
    else:
        wxyz = quaternion_multiply(quaternion_multiply(q_, pc), q)
    return wxyz[:, :, 1:4]
