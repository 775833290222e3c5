"""MI355X-native MPPI rollout-and-cost hot path of ccv_mppi_path_tracker.

The product is libccv_mppi_hip.so (hand-written gfx950 kernels behind the C ABI of include/ccv_mppi.h);
this package is the thin Python host used by the tests and bench.py.  It never imports oracle/.
"""
from . import capi, configs  # noqa: F401
from .controller import MPPIController, calc_ref_path, make_path, plant_step  # noqa: F401
from .node import ControllerNode, FullBodyStateEstimator  # noqa: F401

__all__ = ["capi", "configs", "MPPIController", "ControllerNode", "FullBodyStateEstimator", "calc_ref_path", "make_path", "plant_step"]
