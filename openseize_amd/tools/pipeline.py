"""Pipeline: a picklable chain of callables, each fed the previous one's
result (the composition helper the reference offers in ``tools/pipeline.py``
:16-124 and one of the callers of the hot path).  No
numerics here: the stages are this package's filters, resamplers and
estimators frozen with ``functools.partial``.
"""

import inspect
from copy import copy
from functools import partial


class Pipeline:
    """``p.append(func, **kwargs)`` freezes every argument of ``func`` but the
    data one; ``p(data)`` runs the stages in order on a shallow copy of
    ``data``."""

    def __init__(self):
        self.callers = []

    def validate(self, caller, **kwargs):
        """TypeError unless ``caller`` bound with ``kwargs`` (defaults applied)
        leaves at most one parameter free."""
        sig = inspect.signature(caller)
        bound = sig.bind_partial(**kwargs)
        bound.apply_defaults()
        free = len(sig.parameters) - len(bound.arguments)
        if free > 1:
            raise TypeError("Pipeline callers must have exactly one unbound argument."
                            f" {caller.__name__} has {free} unbound arguments.")

    def append(self, caller, **kwargs):
        self.validate(caller, **kwargs)
        self.callers.append(partial(caller, **kwargs))

    def __contains__(self, caller):
        return caller in [part.func for part in self.callers]

    def __call__(self, data):
        res = copy(data)
        for caller in self.callers:
            res = caller(res)
        return res
