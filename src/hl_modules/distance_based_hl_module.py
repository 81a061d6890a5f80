from sound_bubble_amd.harness import PLModule  # noqa: F401  (JSON: pl_module)
