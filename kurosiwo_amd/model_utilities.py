"""Model factory with the reference's signature (/root/reference/models/model_utilities.py:182-237)."""
import torch

from . import _lib
from .snunet import SNUNet_ECAM


def initialize_cd_model(configs, model_configs, phase="train"):
    method = configs["method"].lower()
    if method == "snunet":
        model = SNUNet_ECAM(configs["num_channels"], configs["num_classes"], base_channel=model_configs["base_channel"],
                            precision=configs.get("precision", "bf16" if configs.get("mixed_precision") else "fp32"))
        if configs.get("sync_bn") is not None:                     # optional key (not in the reference's configs): SyncBN under data parallelism
            model.sync_bn = bool(configs["sync_bn"])
    elif method == "changeformer":
        from .changeformer import ChangeFormerV6
        if model_configs.get("multi_scale_train"):
            # change_detection_trainer.py:155-162 feeds the int64 [B,H,W] mask to F.interpolate(mode="nearest"), which raises
            # ("compute_indices_weights_nearest" not implemented for 'Long'): the reference cannot run this branch either
            raise NotImplementedError("changeformer: multi_scale_train (the reference's own branch raises on the int64 mask; "
                                      "only the last output is differentiable here)")
        model = ChangeFormerV6(embed_dim=model_configs["embed_dim"], input_nc=configs["num_channels"], output_nc=configs["num_classes"],
                               decoder_softmax=model_configs["decoder_softmax"],
                               precision=configs.get("precision", "bf16" if configs.get("mixed_precision") else "fp32"))
    elif method in ("siam-conc", "siam-diff"):             # model_utilities.py:183-190
        from .fcsiam import SiamUnet_conc, SiamUnet_diff
        cls = SiamUnet_conc if method == "siam-conc" else SiamUnet_diff
        model = cls(input_nbr=configs["num_channels"], label_nbr=configs["num_classes"],
                    precision=configs.get("precision", "bf16" if configs.get("mixed_precision") else "fp32"))
    elif method == "bit-cd":                               # model_utilities.py:191-192: define_G(model_configs, in_channels)
        from .bitcd import define_G
        model = define_G(model_configs, in_channels=configs["num_channels"],
                         precision=configs.get("precision", "bf16" if configs.get("mixed_precision") else "fp32"))
    else:
        raise _lib.KsmiError(f"method {method!r} has no HIP implementation (change-detection methods in scope: snunet, changeformer, "
                             "siam-conc, siam-diff, bit-cd)")
    model = model.to(configs["device"])
    if configs.get("resume_checkpoint"):
        ck = torch.load(configs["resume_checkpoint"], map_location=configs["device"])
        model.load_state_dict(ck["model_state_dict"])
        if ck.get("rng_state") and hasattr(model, "manual_seed"):          # continue the Dropout / DropPath stream of the saved run
            # (the words were saved by rank 0, whose seed is the base seed: every rank re-derives its own share)
            model.manual_seed(ck["rng_state"][0], ck["rng_state"][1], fold_rank=True)
    print(model.__class__.__name__, f"({sum(p.numel() for p in model.parameters())} parameters, precision={model.precision})")
    return model


def initialize_segmentation_model(config, model_configs):
    """model_utilities.py:97-167 for the method with a HIP implementation: `finetune` = FloodViT
    (FinetunerSegmentation over a pickled MAE-pretrained ViT encoder, :158-165).  Without `config["encoder"]` a
    randomly initialised encoder of configs/method/mae/mae.json's size is used (no checkpoint can be fetched here)."""
    from .floodvit import FinetunerSegmentation, ViT
    if config["method"].lower() == "unet":
        from .unet import Unet
        model = Unet(encoder_name=model_configs["backbone"], encoder_weights=model_configs["encoder_weights"], in_channels=config["num_channels"],
                     classes=config["num_classes"], precision=config.get("precision", "bf16" if config.get("mixed_precision") else "fp32"))
        print(model.__class__.__name__, f"({sum(p.numel() for p in model.parameters())} parameters, precision={model.precision})")
        return model.to(config["device"])
    if config["method"].lower() != "finetune":
        raise _lib.KsmiError(f"segmentation method {config['method']!r} has no HIP implementation (in scope: unet, finetune = FloodViT)")
    if config.get("encoder"):
        encoder = torch.load(config["encoder"], map_location="cpu", weights_only=False)
    else:
        mc = model_configs
        encoder = ViT(image_size=mc.get("image_size", 224), patch_size=mc.get("patch_size", 16), num_classes=mc.get("num_classes", 1000),
                      dim=mc.get("dim", 1024), depth=mc.get("depth", 24), heads=mc.get("heads", 16), mlp_dim=mc.get("mlp_dim", 2048),
                      channels=config["num_channels"])
    cfg = dict(config)
    cfg.setdefault("decoder", True)
    cfg.setdefault("mlp", False)
    cfg.setdefault("linear_eval", False)
    cfg.setdefault("finetuning_patch_size", 16)
    model = FinetunerSegmentation(encoder=encoder, configs=cfg,
                                  precision=config.get("precision", "bf16" if config.get("mixed_precision") else "fp32"))
    print(model.__class__.__name__, f"({sum(p.numel() for p in model.parameters())} parameters, precision={model.precision})")
    return model.to(config["device"])
