"""Drop-in for `open_universe.inference_utils` (reference: open_universe/inference_utils/__init__.py)."""
from .model_loader import load_model, load_model_sharded
from .signature_to_parser import add_enhance_arguments

__all__ = ["load_model", "load_model_sharded", "add_enhance_arguments"]
