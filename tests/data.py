"""Synthetic point clouds of SURVEY.md section 8(d) -- the generators live in gspn_amd/synth.py (bench.py uses them too)."""
from gspn_amd.synth import batch, cloud_d, cloud_s, cloud_u  # noqa: F401
