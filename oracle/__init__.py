"""CPU oracles for the differentiable Gaussian rasterizer path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product path (``gs-dynamics_amd/``) never does and has no CPU fallback.

PARITY UNPINNED: the reference (robo-alex/gs-dynamics) delegates this path to the un-vendored,
unpinned third-party CUDA extension ``diff-gaussian-rasterization-w-depth``
(/root/reference/README.md:28-32) and ships no tests or golden vectors for it.  The two oracles
here restate the published algorithm independently of each other:

* ``oracle.tiled``  (O2) -- fp32, tile-based, explicit backward, plain C (``gsr_oracle.c``).
* ``oracle.dense_oracle`` (O1) -- fp64, dense per-pixel PyTorch autograd (no hand-derived backward).
"""
from .tiled import TiledOracle, OracleCamera, build_oracle_lib  # noqa: F401
