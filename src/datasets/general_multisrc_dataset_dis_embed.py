from sound_bubble_amd.data import BubbleFolderDataset as Dataset  # noqa: F401  (JSON: train_dataset / val_dataset)
