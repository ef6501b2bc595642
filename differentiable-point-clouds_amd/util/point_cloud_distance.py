"""Mirror of dpc/util/point_cloud_distance.py:26-39 (the nearest-neighbour kernel of the
Chamfer evaluation, dpc/run/eval_chamfer.py:18-34)."""
import torch

from .. import ops


def point_cloud_distance(Vs, Vt):  # noqa: N803 (reference argument names)
    """For each point in Vs [VsN,3] the closest point in Vt [VtN,3]:
    returns (proj [VsN,3], minDist [VsN], idx [VsN] int32).  Runs in the tensors' own
    precision (float64 in the reference's evaluation); no source chunking is needed."""
    if Vs.dim() != 2 or Vt.dim() != 2 or Vs.shape[1] != 3 or Vt.shape[1] != 3:
        raise ValueError("point_cloud_distance expects [N,3] tensors, got %s and %s" % (tuple(Vs.shape), tuple(Vt.shape)))
    if Vs.shape[0] == 0 or Vt.shape[0] == 0:
        raise ValueError("point_cloud_distance needs non-empty point sets")
    return ops.NNDistance.apply(Vs, Vt)


def chamfer_distance(pred, gt):
    """eval_chamfer.py:118-127: (mean pred->gt distance, mean gt->pred distance)."""
    _, d_pred_to_gt, _ = point_cloud_distance(pred, gt)
    _, d_gt_to_pred, _ = point_cloud_distance(gt, pred)
    return torch.mean(d_pred_to_gt), torch.mean(d_gt_to_pred)
