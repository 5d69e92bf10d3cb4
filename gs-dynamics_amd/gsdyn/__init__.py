"""gsdyn -- host-side callers of the rasterizer path.

Python mirrors of the reference's own callers of ``diff_gaussian_rasterization`` (the reference is
Python here, so is this): camera record construction, parameter activation, the ``get_loss`` step of
``src/tracking/train_utils.py`` and the one-view-per-GPU data-parallel driver that the north star adds.
Nothing in this package computes a render: every render goes through ``GaussianRasterizer``.
"""
from .camera import setup_camera, look_at_w2c, Rt_to_w2c  # noqa: F401
from .scene import synth_scene_params, synth_ring_cameras, synth_targets  # noqa: F401
from .step import params2rendervar, get_loss, get_loss_views, loss_and_grads_views, LossWeights, initialize_optimizer  # noqa: F401
from .train import train, train_timestep, initialize_per_timestep, initialize_post_first_timestep, params2cpu, save_params  # noqa: F401
