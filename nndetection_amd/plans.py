"""Synthetic plans in the reference's own plan format.

The reference derives `plan_arch` / `plan_anchors` from the dataset with nnU-Net's (un-vendored)
`get_pool_and_conv_props` (nndet/planning/architecture/boxes/base.py:637-641). BASELINE.json's
configs name patch sizes that do not occur in the reference, so -- like SURVEY.md section 8 -- we
instantiate the planner defaults (nndet/planning/architecture/boxes/c002.py:42-54: start 32,
fpn 128, head 128, batch 4, 4 decoder levels, max 320 channels) on those patch sizes. The dict
keys are exactly the ones `RetinaUNetModule.from_config_plan` reads
(nndet/ptmodule/retinaunet/base.py:338-466), `MODEL_CFG_V001` restates
nndet/conf/train/v001.yaml:61-107.
"""
import copy

MODEL_CFG_V001 = {
    "encoder_kwargs": {},
    "decoder_kwargs": {
        "min_out_channels": 8, "upsampling_mode": "transpose",
        "num_lateral": 1, "norm_lateral": False, "activation_lateral": False,
        "num_out": 1, "norm_out": False, "activation_out": False,
    },
    "head_kwargs": {},
    "head_classifier_kwargs": {"num_convs": 1, "norm_channels_per_group": 16, "norm_affine": True,
                               "reduction": "mean", "loss_weight": 1., "prior_prob": 0.01},
    "head_regressor_kwargs": {"num_convs": 1, "norm_channels_per_group": 16, "norm_affine": True,
                              "reduction": "sum", "loss_weight": 1., "learn_scale": True},
    "head_sampler_kwargs": {"batch_size_per_image": 32, "positive_fraction": 0.33, "pool_size": 20, "min_neg": 1},
    "segmenter_kwargs": {"dice_kwargs": {"batch_dice": True}},
    "matcher_kwargs": {"num_candidates": 4, "center_in_gt": False},
    "plan_arch_overwrites": {},
    "plan_anchors_overwrites": {},
}

TRAINER_CFG_V001 = {  # nndet/conf/train/v001.yaml:29-58
    "initial_lr": 0.01, "sgd_momentum": 0.9, "sgd_nesterov": True, "weight_decay": 3.e-5,
    "warm_iterations": 4000, "warm_lr": 1.e-6, "poly_gamma": 0.9,
    "max_num_epochs": 50, "num_train_batches_per_epoch": 2500,
}

_W4 = [(4, 8, 16), (8, 16, 32), (16, 32, 64), (32, 64, 128)]


def _arch(n_stages, strides, levels, start=32, fpn=128, head=128, max_ch=320, in_ch=1):
    return {
        "arch_name": "RetinaUNetV001", "dim": 3, "in_channels": in_ch,
        "conv_kernels": [[3, 3, 3]] * n_stages, "strides": [list(s) for s in strides],
        "start_channels": start, "fpn_channels": fpn, "head_channels": head, "max_channels": max_ch,
        "classifier_classes": 1, "seg_classes": 1, "decoder_levels": tuple(levels),
    }


def _anch(w):
    return {"width": [tuple(x) for x in w], "height": [tuple(x) for x in w], "depth": [tuple(x) for x in w], "stride": 1}


PLANS = {
    # BASELINE.json configs[0]: toy Task000, 64^3, batch 2 (SURVEY 8d "Config 1")
    "toy64": {"patch_size": (64, 64, 64), "batch_size": 2,
              "arch": _arch(5, [[2, 2, 2]] * 4, (1, 2, 3, 4)), "anchors": _anch(_W4)},
    # configs[1]/[2]: Task016_Luna-like, 160x160x96, batch 4 per GPU (SURVEY 8 header)
    "luna160": {"patch_size": (160, 160, 96), "batch_size": 4,
                "arch": _arch(6, [[2, 2, 2]] * 4 + [[2, 2, 1]], (2, 3, 4, 5)), "anchors": _anch(_W4)},
    # configs[3]: Task012_LIDC-like, 192x192x128
    "lidc192": {"patch_size": (192, 192, 128), "batch_size": 4,
                "arch": _arch(6, [[2, 2, 2]] * 5, (2, 3, 4, 5)), "anchors": _anch(_W4)},
    # small network for gradient / golden tests (same block types, A = 27, odd grid sizes at the top level)
    "tiny": {"patch_size": (32, 32, 24), "batch_size": 2,
             "arch": _arch(3, [[2, 2, 2]] * 2, (1, 2), start=32, fpn=64, head=32, max_ch=64),
             "anchors": _anch([(4, 6, 8), (8, 12, 16)])},
}


def get_plan(name: str) -> dict:
    return copy.deepcopy(PLANS[name])
