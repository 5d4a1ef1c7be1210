"""Time-coordinate helpers used by the growth-rate recipes
(reference: utils/datetime_utils.py:108-166; pandas only, xarray optional)."""
from datetime import datetime

import numpy as np


def get_datetime_from_coord(coord) -> list[datetime]:
    import pandas as pd
    return pd.to_datetime(np.asarray(getattr(coord, "values", coord))).to_pydatetime().tolist()


def time_diff(datetime_list: list[datetime]) -> list[float]:
    """Centred first differences in fractional minutes, one-sided at the ends."""
    n = len(datetime_list)
    head = [(datetime_list[1] - datetime_list[0]).total_seconds() / 60]
    mid = [(datetime_list[i + 2] - datetime_list[i]).total_seconds() / 120 for i in range(n - 2)]
    tail = [(datetime_list[-1] - datetime_list[-2]).total_seconds() / 60]
    return head + mid + tail


def get_time_diff_from_coord(coord) -> np.ndarray:
    return np.array(time_diff(get_datetime_from_coord(coord)))


__all__ = ("get_datetime_from_coord", "time_diff", "get_time_diff_from_coord")
