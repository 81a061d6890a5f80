from sound_bubble_amd.losses import MultiResoFuseLoss  # noqa: F401  (JSON: pl_module_args.loss, fine-tune stage)
