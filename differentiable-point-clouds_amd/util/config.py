"""Attribute-style config with the reference defaults that the hot path reads
(dpc/resources/default_config.yaml).  The functions in
this package accept ANY object with these attributes (e.g. the reference's
EasyDict); this class is a convenience for callers that have none."""


class Config(dict):
    DEFAULTS = dict(
        vox_size=64,                       # default_config.yaml:77
        vox_size_z=-1,                     # :78
        camera_distance=2.0,               # :81
        focal_length=1.875,                # :79
        pose_quaternion=True,              # :35
        pc_gauss_kernel_size=11,           # :55 (experiments use 21)
        pc_separable_gauss_filter=True,    # :56
        pc_relative_sigma=1.0,             # :49
        pc_relative_sigma_end=0.2,         # :50
        max_number_of_steps=600000,        # :122
        ptn_max_projection=False,          # :83
        drc_logsum=True,                   # :88
        drc_logsum_clip_val=1e-5,          # :89
        drc_tf_cumulative=True,            # :90
        max_depth=10.0,                    # :85
        pc_rgb=False,                      # :61
        pc_rgb_stop_points_gradient=False,
        pc_rgb_clip_after_conv=False,
        pc_rgb_divide_by_occupancies=False,
        pc_rgb_divide_by_occupancies_epsilon=0.01,
        # caller side (dpc/models/model_pc.py:225-299)
        pc_normalise_gauss=False, pc_normalise_gauss_analytical=True,     # :51-52 (slow path only)
        pc_fast=True, predict_pose=False, predict_translation=False,
        pc_point_dropout=1.0, pc_learn_occupancy_scaling=True,
        pose_predict_num_candidates=1, step_size=4, batch_size=8,
        pc_num_points=8000, learn_focal_length=False,
        pc_point_dropout_scheduled=True, pc_point_dropout_exponential_schedule=False,
        pc_point_dropout_start_step=0.0, pc_point_dropout_end_step=1.0,
        pose_predictor_student=True, pose_predictor_student_loss_weight=1.0,
        pose_student_align_loss=False, variable_num_views=False,
        # loss side (model_pc.py:383-445)
        bicubic_gt_downsampling=False, pc_gauss_filter_gt=False, pc_gauss_filter_gt_rgb=False,
        pc_gauss_filter_gt_switch_off=False, proj_rgb_weight=0.0, max_dataset_depth=10.0,
        proj_weight=1.0, drc_weight=0.0, proj_depth_weight=0.0,
        # (this build, not reference keys) in-kernel point dropout / replication over views and candidates
        pc_fused_dropout=True, pc_replicate_in_kernel=True, pc_fused_proj_loss=True, pc_trim_gauss_taps=True,
    )

    def __init__(self, **kw):
        super().__init__(self.DEFAULTS)
        unknown = set(kw) - set(self.DEFAULTS)
        if unknown:
            raise KeyError("unknown config keys: %s" % sorted(unknown))  # config.py:7-41 is strict too
        self.update(kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    # every mutation bumps a counter, so that per-step callers (util.point_cloud._meta) can reuse what they derived from
    # the config instead of re-reading a dozen keys per projector call
    def __setitem__(self, k, v):
        dict.__setitem__(self, k, v)
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1

    def update(self, *a, **kw):
        dict.update(self, *a, **kw)
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1

    def __delitem__(self, k):
        dict.__delitem__(self, k)
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1

    def pop(self, *a):
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1
        return dict.pop(self, *a)

    def setdefault(self, k, d=None):
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1
        return dict.setdefault(self, k, d)

    def clear(self):
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1
        dict.clear(self)

    def popitem(self):
        self.__dict__["_edits"] = self.__dict__.get("_edits", 0) + 1
        return dict.popitem(self)

    def __ior__(self, other):            # cfg |= {...}: dict.__ior__ writes without going through update()
        self.update(other)
        return self


def default_config(**kw):
    return Config(**kw)
