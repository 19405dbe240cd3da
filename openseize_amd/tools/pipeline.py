"""Pipeline: a picklable chain of callables, each fed the previous one's
result -- the composition helper of the reference's ``tools/pipeline.py``
(:16-124), one of the callers of the hot path.  No numerics
here: the stages are this package's filters, resamplers and estimators with
all but their data argument frozen.
"""

import copy
import functools
import inspect

_VARIADIC = (inspect.Parameter.VAR_POSITIONAL, inspect.Parameter.VAR_KEYWORD)


def _open_parameters(func, frozen):
    """Names of the parameters of ``func`` that neither ``frozen`` nor a
    default value (nor ``*args`` / ``**kwargs``) fills."""
    params = inspect.signature(func).parameters
    unknown = set(frozen) - set(params)
    if unknown and not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
        raise TypeError(f"{func.__name__} got unexpected keyword arguments {sorted(unknown)}")
    return [name for name, p in params.items()
            if name not in frozen and p.default is p.empty and p.kind not in _VARIADIC]


class Pipeline:
    """``append(func, **kwargs)`` adds a stage; calling the pipeline runs the
    stages in order, each on the previous result, starting from a shallow copy
    of the input.  ``callers`` holds the stages as ``functools.partial``
    objects; ``func in pipeline`` tests membership by function."""

    def __init__(self):
        self.callers = []

    def validate(self, caller, **kwargs):
        """A stage may leave at most its data argument open."""
        still_open = _open_parameters(caller, kwargs)
        if len(still_open) > 1:
            raise TypeError(
                "Pipeline callers must have exactly one unbound argument. "
                f"{caller.__name__} has {len(still_open)} unbound arguments.")

    def append(self, caller, **kwargs):
        self.validate(caller, **kwargs)
        self.callers.append(functools.partial(caller, **kwargs))

    def __contains__(self, caller):
        return any(stage.func is caller or stage.func == caller for stage in self.callers)

    def __call__(self, data):
        result = copy.copy(data)
        for stage in self.callers:
            result = stage(result)
        return result
