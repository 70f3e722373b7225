"""raytracer_amd — MI355X (gfx950) backend for the per-pixel tracing loop of matteoilardi/Raytracer.

The product is the C-ABI shared library built from raytracer_amd/csrc (include/raytracer_amd.h);
this package is the thin Python host side used by the tests and bench.py.
"""
from . import abi  # noqa: F401
from .scene import (Camera, FlatScene, SceneBuilder, Transformation, compose, degs_to_rads, identity, rotation_x,  # noqa: F401
                    rotation_y, rotation_z, scaling, translation)
