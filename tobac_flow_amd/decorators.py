"""`configure_dataarray`: give array results the coordinates of an xarray input.

Public behaviour follows the reference decorator of the same name (tobac_flow/decorators.py:21-61): the
first xr.DataArray found among the positional, then the keyword arguments serves as a template; every
result that is not already a DataArray is wrapped into a copy of it (encoding dropped) and renamed; the
listed attributes are removed and the given ones set, on wrapped and on passed-through DataArrays alike;
a tuple of results is treated element-wise; a call without any DataArray is passed straight through.

xarray is optional here: without it nothing can be a DataArray, so the decorator is the identity.
"""
import functools

try:
    import xarray as xr
except ImportError:                      # glue only: the hot path itself never needs xarray
    xr = None

_DEFAULT_DROPPED = ("valid_range", "cell_methods", "units_metadata", "_FillValue", "missing_value")


class _Rewrapper:
    """How one decorated function labels its results."""

    def __init__(self, name, drop_attrs, attributes):
        self.name, self.drop_attrs, self.attributes = name, tuple(drop_attrs), dict(attributes)

    def template_of(self, args, kwargs):
        for candidate in (*args, *kwargs.values()):
            if isinstance(candidate, xr.DataArray):
                return candidate
        return None

    def label(self, template, result):
        if isinstance(result, xr.DataArray):
            labelled = result
        else:
            labelled = template.copy(data=result).drop_encoding()
            labelled.name = self.name
        for key in self.drop_attrs:
            labelled.attrs.pop(key, None)
        labelled.attrs.update(self.attributes)
        return labelled

    def __call__(self, template, result):
        if type(result) == tuple:                      # noqa: E721 - exactly a tuple, as in the reference
            return tuple(self.label(template, item) for item in result)
        return self.label(template, result)


def configure_dataarray(name=None, drop_attrs=_DEFAULT_DROPPED, **attributes):
    def decorate(func):
        if xr is None:
            return func

        @functools.wraps(func)
        def wrapped(*args, name=name, drop_attrs=drop_attrs, attributes=attributes, **kwargs):
            rewrap = _Rewrapper(name, drop_attrs, attributes)
            template = rewrap.template_of(args, kwargs)
            result = func(*args, **kwargs)
            return result if template is None else rewrap(template, result)

        return wrapped

    return decorate


__all__ = ("configure_dataarray",)
