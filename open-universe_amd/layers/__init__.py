from .dyn_range_comp import CompressedMagSTFT, CompressedMagSTFTPadded, IdentityTransform

__all__ = ["CompressedMagSTFT", "CompressedMagSTFTPadded", "IdentityTransform"]
