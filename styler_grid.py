"""Module name for drivers of the grid (TNST-style) path: re-exports `neural_flow_style_amd.styler_grid`
(`Styler(config).run({'d': frames, 'v': simulation velocities})`, BASELINE configs[3])."""
from neural_flow_style_amd.styler_grid import *  # noqa: F401,F403
from neural_flow_style_amd.styler_grid import Styler  # noqa: F401
