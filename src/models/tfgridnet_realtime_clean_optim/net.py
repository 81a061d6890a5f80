from sound_bubble_amd.net import NetOptim as Net  # noqa: F401  (JSON: pl_module_args.model)
