"""Parameter containers with the semantics of the reference's ``BaseConfiguration``
(mythos/energy/configuration.py:16-123): required / non-optimizable / dependent parameters,
``opt_params``, ``init_params`` (derives the dependent ones), ``from_dict``, ``to_dictionary`` and
the ``|`` merge.  Instances are immutable; every mutator returns a new object.

Instead of one hand-written dataclass per term, a configuration class is described by a small
spec (ordered required names, declared dependent names, optional extras and a derive function);
``mythos_amd.energy.terms`` instantiates the specs with the reference's parameter names.
"""

from __future__ import annotations

import warnings
from typing import Any, Callable

ERR_MISSING_REQUIRED_PARAMS = "Required properties {props} are not initialized."
ERR_OPT_DEPENDENT_PARAMS = "Only {req_params} permitted for optimization, but found {given_params}"
WARN_INIT_PARAMS_NOT_IMPLEMENTED = "init_params not implemented"
WARN_DEPENDENT_PARAMS_NOT_INITIALIZED = "Dependent parameters not initialized"

OPT_ALL = ("*",)
_META = ("params_to_optimize", "required_params", "non_optimizable_required_params", "dependent_params")


class BaseConfiguration:
    """Frozen parameter record.  Subclasses set the class attributes below."""

    required_params: tuple = ()
    dependent_params: tuple = ()
    hidden_dependent_params: tuple = ()  # set by init_params but not declared (dna2 coaxial, Debye)
    optional_params: tuple = ()
    non_optimizable_required_params: tuple = ()
    OPT_ALL: tuple = OPT_ALL
    _derive: Callable[["BaseConfiguration"], dict] | None = None

    def __init__(self, params_to_optimize: tuple = (), non_optimizable_required_params=None, **values: Any):
        cls = type(self)
        fields = set(cls.required_params) | set(cls.dependent_params) | set(cls.hidden_dependent_params) | set(cls.optional_params)
        unknown = set(values) - fields
        if unknown:
            raise TypeError(f"{cls.__name__} got unexpected parameters {sorted(unknown)}")
        object.__setattr__(self, "_values", {k: values.get(k) for k in self._field_order()})
        object.__setattr__(self, "params_to_optimize", tuple(params_to_optimize))
        if non_optimizable_required_params is not None:
            object.__setattr__(self, "non_optimizable_required_params", tuple(non_optimizable_required_params))
        missing = [p for p in cls.required_params if self._values.get(p) is None]
        if missing:
            raise ValueError(ERR_MISSING_REQUIRED_PARAMS.format(props=",".join(missing)))
        optimizable = set(cls.required_params) - set(self.non_optimizable_required_params)
        bad = set(self.params_to_optimize) - optimizable
        if bad and bad != set(OPT_ALL):
            raise ValueError(
                ERR_OPT_DEPENDENT_PARAMS.format(req_params=",".join(sorted(optimizable)), given_params=",".join(sorted(bad)))
            )

    @classmethod
    def _field_order(cls) -> tuple:
        return (*cls.required_params, *cls.optional_params, *cls.dependent_params, *cls.hidden_dependent_params)

    # ---- dependent parameters on demand -----------------------------------------------------------
    # ``init_params`` marks the record; the derivation runs when a dependent value is first read (``cfg["b_low_stack"]``,
    # ``to_dictionary(include_dependent=True)``, ``items()``, pickling, printing).  The kernels never read them - the flat
    # parameter vector is derived from the required values in one pass (energy/flat_params.py) - and eagerly deriving
    # them cost a composed function's ``with_params`` eight full derivations, 7.8 ms per call: more than the 6 400-frame
    # energy launch it precedes in a DiffTRe iteration (round 4, scripts/prof_host_r04.py).
    def _materialise(self) -> None:
        d = object.__getattribute__(self, "__dict__")
        if d.get("_lazy"):
            d["_lazy"] = False
            d["_values"].update(type(self)._derive(self))

    def _is_dependent(self, name: str) -> bool:
        cls = type(self)
        return name in cls.dependent_params or name in cls.hidden_dependent_params

    # ---- mapping-ish access ---------------------------------------------------------------------
    def __getattr__(self, name: str):
        vals = object.__getattribute__(self, "_values")
        if name in vals:
            if self._is_dependent(name):
                self._materialise()
            return vals[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        raise AttributeError(f"{type(self).__name__} is frozen; use replace()")

    def __contains__(self, name: str) -> bool:
        return name in self._values or name in _META

    def __getitem__(self, name: str):
        if name in self._values:
            if self._is_dependent(name):
                self._materialise()
            return self._values[name]
        return getattr(self, name)

    def keys(self):
        return list(self._values.keys())

    def items(self):
        self._materialise()
        return list(self._values.items())

    def __iter__(self):
        return iter(self._values)

    def __repr__(self) -> str:
        self._materialise()
        body = ", ".join(f"{k}={v!r}" for k, v in self._values.items() if v is not None)
        return f"{type(self).__name__}({body})"

    def __getstate__(self):
        self._materialise()
        return {
            "values": self._values,
            "opt": self.params_to_optimize,
            "noopt": tuple(self.non_optimizable_required_params),
        }

    def __setstate__(self, state):
        object.__setattr__(self, "_values", state["values"])
        object.__setattr__(self, "params_to_optimize", state["opt"])
        object.__setattr__(self, "non_optimizable_required_params", state["noopt"])

    # ---- reference API ----------------------------------------------------------------------------
    def replace(self, **changes: Any) -> "BaseConfiguration":
        meta = {k: changes.pop(k) for k in list(changes) if k in ("params_to_optimize", "non_optimizable_required_params")}
        vals = dict(self._values)
        vals.update(changes)
        new = type(self)(
            params_to_optimize=meta.get("params_to_optimize", self.params_to_optimize),
            non_optimizable_required_params=meta.get("non_optimizable_required_params", self.non_optimizable_required_params),
            **vals,
        )
        if self.__dict__.get("_lazy"):  # dependents still owed: the copy derives them (from ITS values) when asked
            object.__setattr__(new, "_lazy", True)
        return new

    @property
    def opt_params(self) -> dict:
        """The parameters selected for optimisation (configuration.py:36-48)."""
        if tuple(self.params_to_optimize) == OPT_ALL:
            return {
                k: v
                for k, v in self._values.items()
                if k in self.required_params and k not in self.non_optimizable_required_params
            }
        return {k: v for k, v in self._values.items() if k in self.params_to_optimize}

    def init_params(self) -> "BaseConfiguration":
        """Derive the dependent parameters (configuration.py:66-72)."""
        derive = type(self)._derive
        if derive is None:
            if type(self).dependent_params:
                warnings.warn(WARN_INIT_PARAMS_NOT_IMPLEMENTED, stacklevel=1)
            return self
        new = self.replace()
        object.__setattr__(new, "_lazy", True)  # derived at the first read of a dependent value (_materialise)
        return new

    @classmethod
    def from_dict(cls, params: dict, params_to_optimize: tuple = ()) -> "BaseConfiguration":
        return cls(params_to_optimize=params_to_optimize, **params)

    def to_dictionary(self, *, include_dependent: bool, exclude_non_optimizable: bool) -> dict:
        params = {k: self._values[k] for k in self.required_params}
        if include_dependent:
            self._materialise()
            for k in self.dependent_params:
                if self._values.get(k) is not None:
                    params[k] = self._values[k]
                else:
                    warnings.warn(WARN_DEPENDENT_PARAMS_NOT_INITIALIZED, stacklevel=1)
        if exclude_non_optimizable:
            for k in self.non_optimizable_required_params:
                params.pop(k, None)
        return params

    def __or__(self, other):
        if isinstance(other, BaseConfiguration):
            return self.replace(**{k: v for k, v in other.items() if v is not None})
        if isinstance(other, dict):
            return self.replace(**other)
        return NotImplemented
