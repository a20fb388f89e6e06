"""ambersim_amd — MI355X-native batched rigid-body rollout engine behind ambersim's API.

`ROOT` keeps the meaning of the reference's `ambersim.ROOT` (reference ambersim/__init__.py:8):
model paths such as "models/pendulum/scene.xml" resolve against it.
"""
from pathlib import Path

ROOT = Path(__file__).resolve().parent
__version__ = "0.1.0"
