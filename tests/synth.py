"""thin re-export so tests read ``from tests.synth import ...``"""
from neural_flow_style_amd.synthetic import *  # noqa: F401,F403
