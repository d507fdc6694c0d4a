"""The synthetic workload now lives with the product (voxe_hip/workload.py: bench.py and the tools measure it, the tests
check it); this module keeps the tests' import path."""
from voxe_hip.workload import *  # noqa: F401,F403
from voxe_hip.workload import CAMERA_ANGLE_X, FAR, NEAR, RADIUS, focal_for, random_grid, refine_scene, sphere_grid, synth_pose_angles  # noqa: F401
