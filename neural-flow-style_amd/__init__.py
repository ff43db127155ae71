"""neural-flow-style_amd: MI355X-native stylisation hot path.

The directory name follows the project name (it contains a hyphen); import it as
``neural_flow_style_amd`` -- the shim module of that name at the repo root loads
this directory as a regular package.

Layout: ``csrc/`` HIP kernels + C ABI (libnfs_hip.so), ``_lib.py`` ctypes loader,
``ops.py`` tensor-level bindings, ``engine.py`` the stylisation step, and the
host-side mirror of the reference's interface (``config``, ``transform``,
``vgg``, ``styler_base``, ``styler_2p``, ``styler_3p``).
"""
from ._lib import NfsLibraryError, build, lib  # noqa: F401

__all__ = ["NfsLibraryError", "build", "lib"]
