"""Stock PyTorch restatement of the reference's encoder / decoder / pose
networks (dpc/nets/img_encoder.py:11-50, pc_decoder.py:5-42, pose_net.py:20-56,
model_pc.py:90-107) for the full training step of BASELINE configs[2]/[3].
These are dense conv / GEMM layers (MIOpen / rocBLAS territory): deliberately
NOT hand-written kernels -- the hand-written part of this repo is the projector
they feed."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class ImgEncoder(nn.Module):
    """5x5 s2 conv, then log2(S/4)-1 x [3x3 s2, 3x3 s1], 3 FC (leaky ReLU)."""

    def __init__(self, image_size=128, f_dim=16, fc_dim=1024, z_dim=1024, predict_pose=True):
        super().__init__()
        act = nn.LeakyReLU(0.2)
        layers = [nn.Conv2d(3, f_dim, 5, stride=2, padding=2), act]
        num_blocks = int(math.log2(image_size / 4) - 1)
        c = f_dim
        for _ in range(num_blocks):
            layers += [nn.Conv2d(c, 2 * c, 3, stride=2, padding=1), act, nn.Conv2d(2 * c, 2 * c, 3, padding=1), act]
            c *= 2
        # the images arrive NHWC (forward permutes a view): keep the filters in the matching layout, otherwise every
        # convolution converts its weights on every call
        self.conv = nn.Sequential(*layers).to(memory_format=torch.channels_last)
        self.fc1 = nn.Linear(c * 4 * 4, fc_dim)
        self.fc2 = nn.Linear(fc_dim, fc_dim)
        self.fc3 = nn.Linear(fc_dim, z_dim)
        self.pose = nn.Linear(fc_dim, z_dim) if predict_pose else None

    def forward(self, images):                       # [B,S,S,3] in [0,1] (reference layout NHWC)
        x = images.permute(0, 3, 1, 2) * 2 - 1
        h = self.conv(x).flatten(1)
        fc1 = F.leaky_relu(self.fc1(h), 0.2)
        fc2 = F.leaky_relu(self.fc2(fc1), 0.2)
        out = {"conv_features": h, "z_latent": fc1, "ids": F.leaky_relu(self.fc3(fc2), 0.2)}
        if self.pose is not None:
            out["poses"] = F.relu(self.pose(fc2))    # slim.fully_connected default activation
        return out


class PcDecoder(nn.Module):
    """FC z -> N*3, tanh / 2 (unit cube); occupancy-scaling head FC z -> 1, sigmoid."""

    def __init__(self, z_dim=1024, num_points=8000, init_stddev=0.025):
        super().__init__()
        self.num_points = num_points
        self.pts = nn.Linear(z_dim, num_points * 3)
        nn.init.trunc_normal_(self.pts.weight, std=init_stddev)
        self.scale = nn.Linear(z_dim, 1)
        nn.init.trunc_normal_(self.scale.weight, std=0.025)

    def forward(self, ids_1):
        pts = torch.tanh(self.pts(ids_1).reshape(-1, self.num_points, 3)) / 2.0
        return {"points_1": pts, "scaling_factor": torch.sigmoid(self.scale(ids_1))}


def _pose_branch(z_dim, num_layers=3, f_dim=32):
    layers, d = [], z_dim
    for k in range(num_layers):
        last = k == num_layers - 1
        layers.append(nn.Linear(d, 4 if last else f_dim))
        if not last:
            layers.append(nn.LeakyReLU(0.2))
        d = f_dim
    return nn.Sequential(*layers)


class PoseNet(nn.Module):
    """Ensemble of pose-candidate MLP branches (+ student branch)."""

    def __init__(self, z_dim=1024, num_candidates=4, student=True):
        super().__init__()
        self.branches = nn.ModuleList([_pose_branch(z_dim) for _ in range(num_candidates)])
        self.student = _pose_branch(z_dim) if student else None

    def forward(self, z):                            # [B*V, z] -> poses [B*V*C, 4]
        q = torch.cat([b(z) for b in self.branches], dim=1).reshape(-1, 4)
        out = {"poses": q}
        if self.student is not None:
            out["pose_student"] = self.student(z)
        return out


class Im2PointCloud(nn.Module):
    """encoder -> (decoder on view 0 of each model, pose net on every view)
    = model_predict of dpc/models/model_pc.py:176-214 for predict_pose=true."""

    def __init__(self, cfg, image_size=128, f_dim=16, fc_dim=1024, z_dim=1024):
        super().__init__()
        self.cfg = cfg
        self.encoder = ImgEncoder(image_size, f_dim, fc_dim, z_dim, predict_pose=True)
        self.decoder = PcDecoder(z_dim, cfg.pc_num_points)
        self.posenet = PoseNet(z_dim, cfg.pose_predict_num_candidates, cfg.pose_predictor_student)

    def forward(self, images):                       # [B*V,S,S,3], model-major then view
        enc = self.encoder(images)
        ids_1 = enc["ids"][::self.cfg.step_size]     # pool_single_view(cfg, ids, 0), model_base.py:7-10
        out = {"ids": enc["ids"], "focal_length": None}
        out.update(self.decoder(ids_1))
        out.update(self.posenet(enc["poses"]))
        return out
