"""configure_dataarray: wrap ndarray results back into the first xr.DataArray argument
(mirrors /root/reference/tobac_flow/decorators.py:21-61).  xarray is optional: without it the
decorated function is returned unchanged (ndarray in -> ndarray out)."""
import functools
from typing import Any, Callable, Optional

try:
    import xarray as xr
except ImportError:       # glue only: the hot path itself never needs xarray
    xr = None


def handle_output(arg, output, name, drop_attrs, attributes):
    if not isinstance(output, xr.DataArray):
        output = arg.copy(data=output).drop_encoding()
        output.name = name
    for key in drop_attrs:
        output.attrs.pop(key, None)
    output.attrs.update(attributes)
    return output


def configure_dataarray(name: Optional[str] = None,
                        drop_attrs=("valid_range", "cell_methods", "units_metadata", "_FillValue", "missing_value"),
                        **attributes) -> Callable:
    def deco(func) -> Callable:
        if xr is None:
            return func

        @functools.wraps(func)
        def wrapper(*args, name=name, drop_attrs=drop_attrs, attributes=attributes, **kwargs) -> Any:
            template = next((a for a in list(args) + list(kwargs.values()) if isinstance(a, xr.DataArray)), None)
            if template is None:
                return func(*args, **kwargs)
            output = func(*args, **kwargs)
            if type(output) == tuple:
                return tuple(handle_output(template, o, name, drop_attrs, attributes) for o in output)
            return handle_output(template, output, name, drop_attrs, attributes)
        return wrapper
    return deco
