"""Model factory with the reference's signature (/root/reference/models/model_utilities.py:182-237)."""
import torch

from . import _lib
from .snunet import SNUNet_ECAM


def initialize_cd_model(configs, model_configs, phase="train"):
    method = configs["method"].lower()
    if method == "snunet":
        model = SNUNet_ECAM(configs["num_channels"], configs["num_classes"], base_channel=model_configs["base_channel"],
                            precision=configs.get("precision", "bf16" if configs.get("mixed_precision") else "fp32"))
    else:
        raise _lib.KsmiError(f"method {method!r} has no HIP implementation yet (in scope this round: snunet; "
                             "changeformer / FloodViT are the next rows of SURVEY.md §8)")
    model = model.to(configs["device"])
    if configs.get("resume_checkpoint"):
        ck = torch.load(configs["resume_checkpoint"], map_location=configs["device"])
        model.load_state_dict(ck["model_state_dict"])
    print(model.__class__.__name__, f"({sum(p.numel() for p in model.parameters())} parameters, precision={model.precision})")
    return model
