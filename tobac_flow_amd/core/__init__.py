"""Abstract contract of the package's central object (see abstracts.py)."""
from tobac_flow_amd.core.abstracts import FLOW_CONTRACT, AbstractFlow

__all__ = ("AbstractFlow", "FLOW_CONTRACT")
