"""`add_enhance_arguments` (reference: open_universe/inference_utils/signature_to_parser.py:26-66):
turn the type hints of `model.enhance` into argparse flags, defaults from `model.diff_kwargs`."""
import typing


def add_enhance_arguments(model, parser):
    if not (hasattr(model, "enhance") and callable(model.enhance)):
        raise ValueError("Model does not have an `enhance` method.")
    hints = typing.get_type_hints(model.enhance)
    hints.pop("return", None)
    defaults = getattr(model, "diff_kwargs", {})
    group = parser.add_argument_group("enhance", "Arguments of enhance function")
    for key, hint in hints.items():
        inner = typing.get_args(hint)
        caster = inner[0] if inner else hint
        group.add_argument(f"--{key}", default=defaults.get(key, None), type=caster)
    return parser
