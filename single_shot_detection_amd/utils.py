"""Behaviour of bf/utils/misc_utils.py:22-29 that the hot path's constructors depend on."""
import inspect


def filter_kwargs(func):
    """Drop keyword arguments the callee does not name (misc_utils.py:22-26).  A ``**kwargs`` parameter does NOT
    make every name acceptable -- only literal parameter names survive -- which is what turns
    ``SigmoidFocalLoss(reduction='sum', ...)`` into ``reduction='mean'`` in the reference."""
    def wrapped_func(*args, **kwargs):
        names = inspect.signature(func).parameters.keys()
        kwargs = {k: v for k, v in kwargs.items() if k in names}
        return func(*args, **kwargs)
    return wrapped_func


def get_ctor(module, name):
    return filter_kwargs(getattr(module, name))
