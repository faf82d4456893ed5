"""MI355X-native batched NMPC solve path for the iterative-learning quadruped MPC stack.

Hot path only: the per-step multiple-shooting NMPC solve and the
per-rollout tracking-error update as HIP kernels for gfx950 behind a C-ABI
(include/nmpc.h), plus the Python host mirror of the reference's solver/controller surface.
"""
__version__ = "0.1.0"
