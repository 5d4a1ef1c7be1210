"""Abstract base class: the method contract of a Flow object
(mirrors /root/reference/tobac_flow/core/abstracts.py:10-84)."""
from abc import ABC, abstractmethod

import numpy as np


class AbstractFlow(ABC):
    @abstractmethod
    def __init__(self, forward_flow, backward_flow) -> None: ...

    @property
    @abstractmethod
    def flow(self): ...

    @abstractmethod
    def __getitem__(self, items) -> "AbstractFlow": ...

    @abstractmethod
    def convolve(self, data, structure=None, method="", fill_value=np.nan, dtype=np.float32, func=None): ...

    @abstractmethod
    def diff(self, data, method="linear", dtype=np.float32): ...

    @abstractmethod
    def sobel(self, data, method="linear", dtype=None, fill_value=np.nan, direction=None): ...

    @abstractmethod
    def watershed(self, field, markers, mask=None, structure=None): ...

    @abstractmethod
    def label(self, data, structure=None, dtype=np.int32, overlap=0, subsegment_shrink=0): ...

    @abstractmethod
    def link_overlap(self, data, structure=None, dtype=np.int32, overlap=0): ...


__all__ = ("AbstractFlow",)
