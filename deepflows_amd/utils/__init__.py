from .model_utils import load_checkpoint, save_checkpoint  # noqa: F401
