"""Simulator protocol (mythos/simulators/base.py:17-43)."""

from __future__ import annotations

import dataclasses as dc
import uuid
from typing import Any, ClassVar


@dc.dataclass(frozen=True)
class SimulatorOutput:
    """Output container for simulators (mythos/simulators/base.py:17-22)."""

    observables: list
    state: dict = dc.field(default_factory=dict)


@dc.dataclass(frozen=True, kw_only=True)
class Simulator:
    """Base class for a simulation (mythos/simulators/base.py:25-43)."""

    name: str = dc.field(default_factory=lambda: str(uuid.uuid4()))
    exposed_observables: ClassVar[list[str]] = ["trajectory"]

    def run(self, *_args, opt_params: dict[str, Any], **_kwargs) -> SimulatorOutput:
        raise NotImplementedError

    def exposes(self) -> list[str]:
        return [f"{obs}.{self.__class__.__name__}.{self.name}" for obs in self.exposed_observables]

    @classmethod
    def create_n(cls, n: int, name: str | None = None, **kwargs) -> list["Simulator"]:
        name = name or str(uuid.uuid4())
        return [cls(name=f"{name}.{i}", **kwargs) for i in range(n)]
