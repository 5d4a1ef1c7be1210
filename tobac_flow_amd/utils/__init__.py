"""Helpers of the hot path that do not depend on the other modules of the package.

The reference gathers its helper modules the same way (tobac_flow/utils/__init__.py); only the helpers the
hot path and its recipes touch exist here, and what the package namespace exposes is spelled out.
"""
from tobac_flow_amd.utils.datetime_utils import get_datetime_from_coord, get_time_diff_from_coord, time_diff
from tobac_flow_amd.utils.flow_utils import select_border_mode, select_interp_mode, select_of_model, warp_flow
from tobac_flow_amd.utils.label_utils import (apply_func_to_labels, find_overlapping_labels, flat_label,
                                              get_step_labels_for_label, labeled_comprehension, make_step_labels,
                                              relabel_objects, remap_labels, slice_labels)
from tobac_flow_amd.utils.normalisation_utils import (inverse_log_norm, linear_norm, linearise_field, local_linear_norm,
                                                      log_norm, select_normalisation_method, to_8bit, uniform_norm, z_norm)
from tobac_flow_amd.utils.stats_utils import mse

__all__ = (
    "get_datetime_from_coord", "get_time_diff_from_coord", "time_diff",
    "select_border_mode", "select_interp_mode", "select_of_model", "warp_flow",
    "apply_func_to_labels", "find_overlapping_labels", "flat_label", "get_step_labels_for_label",
    "labeled_comprehension", "make_step_labels", "relabel_objects", "remap_labels", "slice_labels",
    "inverse_log_norm", "linear_norm", "linearise_field", "local_linear_norm", "log_norm",
    "select_normalisation_method", "to_8bit", "uniform_norm", "z_norm",
    "mse",
)
