"""Drop-in module name of the reference (`styler_base.py`): re-exports the MI355X implementation
`neural_flow_style_amd.styler_base` so that a reference driver's `import styler_base` keeps working."""
from neural_flow_style_amd.styler_base import *  # noqa: F401,F403
from neural_flow_style_amd import styler_base as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
