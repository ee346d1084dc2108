"""Helpers of the -m gpu parity tests.  The staged frame loop itself lives in the package (cudagaussianrenderer_amd/staged.py)."""
from cudagaussianrenderer_amd.staged import StagedFrame, image_diff_stats  # noqa: F401
