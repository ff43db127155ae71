"""Drop-in module name of the reference (`util.py`): re-exports the MI355X implementation
`neural_flow_style_amd.util` so that a reference driver's `import util` keeps working."""
from neural_flow_style_amd.util import *  # noqa: F401,F403
from neural_flow_style_amd import util as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
