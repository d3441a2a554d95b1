"""Trajectory container of a simulator (mythos/simulators/io.py:18-170), on torch tensors."""

from __future__ import annotations

import dataclasses as dc
from typing import Any, Callable

import numpy as np
import torch

from mythos_amd.energy.base import Quaternion, RigidBody
from mythos_amd.input.trajectory import quaternion_to_axes, write_frames


@dc.dataclass(frozen=True)
class SimulatorTrajectory(RigidBody):
    """center (S,N,3), orientation.vec (S,N,4), optional box_size (S,3), temperature (S,) in kT
    units and per-state metadata (mythos/simulators/io.py:18-60)."""

    box_size: torch.Tensor | None = None
    temperature: torch.Tensor | None = None
    metadata: dict | None = None

    @classmethod
    def from_rigid_body(cls, rigid_body: RigidBody, **kwargs: Any) -> "SimulatorTrajectory":
        return cls(center=rigid_body.center, orientation=rigid_body.orientation, **kwargs)

    def replace(self, **kw) -> "SimulatorTrajectory":
        return dc.replace(self, **kw)

    def length(self) -> int:
        return int(self.center.shape[0])

    def with_state_metadata(self, **metadata) -> "SimulatorTrajectory":
        new = dict(self.metadata) if self.metadata is not None else {}
        for k, v in metadata.items():
            new[k] = torch.stack([torch.as_tensor(v)] * self.length())
        return self.replace(metadata=new)

    def filter(self, filter_fn: Callable[[Any], torch.Tensor]) -> "SimulatorTrajectory":
        idx = torch.where(filter_fn(self.metadata))[0]
        return self.slice(idx)

    def slice(self, key) -> "SimulatorTrajectory":
        if isinstance(key, int):
            key = slice(key, key + 1)
        if not isinstance(key, slice):
            key = torch.as_tensor(key)
        meta = None if self.metadata is None else {k: v[key, ...] for k, v in self.metadata.items()}
        return self.replace(
            center=self.center[key, ...],
            orientation=Quaternion(vec=self.orientation.vec[key, ...]),
            box_size=None if self.box_size is None else self.box_size[key, ...],
            temperature=None if self.temperature is None else self.temperature[key, ...],
            metadata=meta,
        )

    @classmethod
    def concat(cls, trajectories: list["SimulatorTrajectory"]) -> "SimulatorTrajectory":
        if not trajectories:
            raise ValueError("Cannot concatenate an empty list of trajectories.")
        if len(trajectories) == 1:
            return trajectories[0]
        box = _concat_optional([t.box_size for t in trajectories], "box sizes")
        temp = _concat_optional([t.temperature for t in trajectories], "temperatures")
        meta = _merge_metadata([t.metadata for t in trajectories], [t.length() for t in trajectories])
        return trajectories[0].replace(
            center=torch.cat([t.center for t in trajectories], dim=0),
            orientation=Quaternion(vec=torch.cat([t.orientation.vec for t in trajectories], dim=0)),
            box_size=box,
            temperature=temp,
            metadata=meta,
        )

    def __add__(self, other: "SimulatorTrajectory") -> "SimulatorTrajectory":
        return type(self).concat([self, other])

    def to_file(self, filepath, box_size=(0, 0, 0), *, native: bool | None = None) -> None:
        """oxDNA text trajectory; velocities, angular momenta and energies are written as zeros
        (mythos/simulators/io.py:146-170)."""
        c = self.center.detach().cpu().double().numpy()
        a1, _, a3 = quaternion_to_axes(self.orientation.vec.detach().cpu().double().numpy())
        s, n = c.shape[0], c.shape[1]
        frames = np.concatenate([c, a1, a3, np.zeros((s, n, 6))], axis=2)
        boxes = self.box_size.detach().cpu().double().numpy() if self.box_size is not None else np.broadcast_to(
            np.asarray(box_size, dtype=np.float64), (s, 3))
        write_frames(filepath, np.arange(s, dtype=np.float64), boxes, np.zeros((s, 3)), frames, native=native)


def _concat_optional(values, label):
    if all(v is None for v in values):
        return None
    if any(v is None for v in values):
        raise ValueError(f"Cannot concatenate, trajectories have incompatible {label}.")
    return torch.cat(values, dim=0)


def _merge_metadata(metadata_list, lengths):
    if all(not m for m in metadata_list):
        return None
    dicts = [dict(m or {}) for m in metadata_list]
    for key in {k for d in dicts for k in d}:
        present = [d[key] for d in dicts if key in d]
        shape = present[0].shape[1:]
        if any(p.shape[1:] != shape for p in present[1:]):
            raise ValueError(f"Metadata key '{key}' has mismatched shapes when adding trajectories.")
        for d, n in zip(dicts, lengths):
            d.setdefault(key, torch.full((n, *shape), float("nan"), dtype=present[0].dtype, device=present[0].device))
    return {k: torch.cat([d[k] for d in dicts], dim=0) for k in dicts[0]}
