"""beamletoptics.jl_amd — MI355X-native engine for BeamletOptics.jl's solve_system!/trace hot path.

Host side (this package): mirrors the reference's System / Beam / component interface, compiles a
scene into the flat tables of include/bmo.h and calls the HIP engine (csrc/libbmo_hip.so) through
the C ABI.  The directory name contains a dot, so import it through the repo-root shim:

    import bmo_amd as bmo
"""
from .linalg import inch, rotate3d as rotation_matrix, align3d as alignment_matrix, normal3d, sag  # noqa: F401
from .shapes import *  # noqa: F401,F403
from .components import *  # noqa: F401,F403
from .beams import *  # noqa: F401,F403
from .system import System, StaticSystem, CompiledScene, Engine, solve_system, make_batch, release, EngineSolution  # noqa: F401
from . import abi, linalg, shapes, components, beams, system  # noqa: F401
