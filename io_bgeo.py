"""Drop-in for the reference drivers' ``import partio`` (``import io_bgeo as partio``): re-exports
`neural_flow_style_amd.io_bgeo` (classic .bgeo version 5 reader / writer with the partio calls the drivers make)."""
from neural_flow_style_amd.io_bgeo import *  # noqa: F401,F403
from neural_flow_style_amd.io_bgeo import INT, FLOAT, VECTOR, ParticleSet, create, from_arrays, read, write  # noqa: F401
