"""Core: abstract classes (reference: tobac_flow/core/__init__.py)."""
from tobac_flow_amd.core.abstracts import *  # noqa: F401,F403
