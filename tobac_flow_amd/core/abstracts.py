"""The method contract of a Flow object, as an abstract base class.

The reference declares the same contract in tobac_flow/core/abstracts.py:10-84.  It is written here as a
table -- operation name, what it must do -- from which the abstract methods are generated, so that the list
of obligations can be read (and tested) in one place; a subclass that leaves one of them out cannot be
instantiated, exactly as with hand-written @abstractmethod stubs.
"""
from abc import ABC, abstractmethod

FLOW_CONTRACT = (
    ("__init__", "store the forward and backward flow vectors, (..., 2) arrays of equal shape"),
    ("__getitem__", "a Flow over the indexed sub-volume"),
    ("convolve", "semi-Lagrangian stencil: gather the structure's neighbours along the flow, then reduce with `func`"),
    ("diff", "semi-Lagrangian centred time difference"),
    ("sobel", "semi-Lagrangian 3 x 3 x 3 Sobel magnitude (optionally uphill / downhill only)"),
    ("watershed", "marker-controlled watershed whose t +- 1 neighbours follow the flow"),
    ("label", "connected components that are linked through time along the flow"),
    ("link_overlap", "link given per-step labels through time by their flow-warped overlap"),
)


def _obligation(op, purpose):
    def stub(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.{op}: {purpose}")
    stub.__name__ = op
    stub.__doc__ = purpose
    return abstractmethod(stub)


class AbstractFlow(ABC):
    """Anything that holds forward / backward flow vectors and offers the operations of FLOW_CONTRACT."""

    @property
    @abstractmethod
    def flow(self):
        """(forward_flow, backward_flow)"""


for _op, _purpose in FLOW_CONTRACT:
    setattr(AbstractFlow, _op, _obligation(_op, _purpose))
AbstractFlow.__abstractmethods__ = frozenset({"flow", *(op for op, _ in FLOW_CONTRACT)})
del _op, _purpose

__all__ = ("AbstractFlow", "FLOW_CONTRACT")
