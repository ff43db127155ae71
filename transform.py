"""Drop-in module name of the reference (`transform.py`): re-exports the MI355X implementation
`neural_flow_style_amd.transform` so that a reference driver's `import transform` keeps working."""
from neural_flow_style_amd.transform import *  # noqa: F401,F403
from neural_flow_style_amd import transform as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
