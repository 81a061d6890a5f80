from sound_bubble_amd.losses import SNRLPLoss  # noqa: F401  (JSON: pl_module_args.loss)
