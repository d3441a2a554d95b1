"""ctypes binding of libmythos_hip.so (the C ABI declared in include/mythos_hip.h).

There is no CPU fallback: if the shared library is missing, or a compute entry point is
called without a visible GPU, the call raises.  Building: ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C mythos_amd/csrc``.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libmythos_hip.so"

c_int_p = C.POINTER(C.c_int32)
c_double_p = C.POINTER(C.c_double)
c_uint8_p = C.POINTER(C.c_uint8)


class MythosHipError(RuntimeError):
    """An error reported by libmythos_hip.so."""


_lib = None


def lib_path() -> Path:
    return Path(os.environ.get("MYTHOS_HIP_LIB", _LIB_PATH))


def load() -> C.CDLL:
    """Load the library once and declare every prototype of include/mythos_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise MythosHipError(
            f"{path} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "mythos_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    V = C.c_void_p
    sigs = {
        "mythos_version": (C.c_char_p, []),
        "mythos_last_error": (C.c_char_p, []),
        "mythos_device_count": (C.c_int, []),
        "mythos_debug_set": (C.c_int, [C.c_int, C.c_int64]),
        "mythos_debug_get": (C.c_int64, [C.c_int]),
        "mythos_oxdna_param_count": (C.c_int, []),
        "mythos_oxdna_param_name": (C.c_char_p, [C.c_int]),
        "mythos_oxdna_create": (V, [C.c_int, C.c_int, c_int_p, c_uint8_p, C.c_int, c_int_p, c_double_p, C.c_int, C.c_int]),
        "mythos_oxdna_destroy": (None, [V]),
        "mythos_oxdna_set_params": (C.c_int, [V, c_double_p, C.c_int]),
        "mythos_oxdna_set_pseq": (C.c_int, [V, c_double_p, c_int_p, C.c_int, c_double_p, C.c_int]),
        "mythos_oxdna_set_nucleotide_types": (C.c_int, [V, c_uint8_p]),
        "mythos_oxdna_set_neighbors": (C.c_int, [V, c_int_p, C.c_int]),
        "mythos_oxdna_build_neighbors": (C.c_int, [V, V, C.c_double, C.c_double, V]),
        "mythos_oxdna_neighbor_stats": (C.c_int, [V, C.POINTER(C.c_int), c_double_p]),
        "mythos_oxdna_energy": (C.c_int, [V, V, V, C.c_int, V, V, V, V, V]),
        "mythos_oxdna_energy_obs": (C.c_int, [V, V, V, C.c_int, V, V, V, V, V, V, V]),
        "mythos_oxdna_energy_dpseq": (C.c_int, [V, V, V, C.c_int, V, V, V, V, V, V, V]),
        "mythos_observables_create": (V, [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, c_int_p, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int]),
        "mythos_observables_destroy": (None, [V]),
        "mythos_observables_width": (C.c_int, [V]),
        "mythos_observables_eval": (C.c_int, [V, V, V, C.c_int, V, V]),
        "mythos_langevin_create": (V, [V, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_double_p, C.c_uint64]),
        "mythos_langevin_destroy": (None, [V]),
        "mythos_langevin_set_neighbor_policy": (C.c_int, [V, C.c_double, C.c_double, C.c_int]),
        "mythos_langevin_init_momenta": (C.c_int, [V, V, V, V]),
        "mythos_langevin_run": (C.c_int, [V, V, V, V, V, C.c_int, C.c_int, V, V, V, V]),
        "mythos_langevin_load": (C.c_int, [V, V, V, V, V, V]),
        "mythos_langevin_advance": (C.c_int, [V, C.c_int, C.c_int, V, V, V, V]),
        "mythos_langevin_store": (C.c_int, [V, V, V, V, V, V]),
        "mythos_langevin_get_step": (C.c_int64, [V]),
        "mythos_langevin_set_step": (C.c_int, [V, C.c_int64]),
        "mythos_langevin_set_seed": (C.c_int, [V, C.c_uint64]),
        "mythos_langevin_last_kernel_ms": (C.c_int, [V, c_double_p, c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mythos_langevin_set_option": (C.c_int, [V, C.c_int, C.c_int64]),
        "mythos_langevin_set_timing": (C.c_int, [V, C.c_int]),
        "mythos_martini_langevin_set_timing": (C.c_int, [V, C.c_int]),
        "mythos_langevin_last_recoveries": (C.c_int, [V, C.POINTER(C.c_int)]),
        "mythos_langevin_last_rebuilds": (C.c_int, [V, C.POINTER(C.c_int)]),
        "mythos_oxdna_read_trajectory": (C.c_int, [C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]),
        "mythos_oxdna_write_trajectory": (C.c_int, [C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
        "mythos_martini_create": (
            V,
            [C.c_int, c_int_p, C.c_int, c_double_p, c_double_p, C.c_int, c_int_p, c_double_p, c_double_p, C.c_int,
             c_int_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_int, C.c_int],
        ),
        "mythos_martini_destroy": (None, [V]),
        "mythos_martini_energy": (C.c_int, [V, V, V, C.c_int, V, V, V]),
        "mythos_martini_param_grads": (C.c_int, [V, V, V, C.c_int, V, V, V, V, V, V, V]),
        "mythos_martini_langevin_create": (V, [V, C.c_double, C.c_double, C.c_double, c_double_p, C.c_uint64]),
        "mythos_martini_langevin_destroy": (None, [V]),
        "mythos_martini_langevin_set_neighbor_policy": (C.c_int, [V, C.c_double, C.c_int]),
        "mythos_martini_langevin_set_inner_list": (C.c_int, [V, C.c_double, C.c_int]),
        "mythos_martini_langevin_init_velocities": (C.c_int, [V, V, V]),
        "mythos_martini_langevin_run": (C.c_int, [V, V, V, c_double_p, C.c_int, C.c_int, V, V, V]),
        "mythos_martini_langevin_load": (C.c_int, [V, V, V, c_double_p, V]),
        "mythos_martini_langevin_advance": (C.c_int, [V, C.c_int, C.c_int, V, V, V]),
        "mythos_martini_langevin_store": (C.c_int, [V, V, V, V]),
        "mythos_martini_langevin_get_step": (C.c_int64, [V]),
        "mythos_martini_langevin_last_rebuilds": (C.c_int, [V, C.POINTER(C.c_int)]),
        "mythos_martini_langevin_last_kernel_ms": (C.c_int, [V, c_double_p, c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mythos_martini_langevin_last_recoveries": (C.c_int, [V, C.POINTER(C.c_int)]),
        "mythos_martini_langevin_neighbor_stats": (C.c_int, [V, C.POINTER(C.c_int), c_double_p]),
        "mythos_martini_langevin_get_rows": (C.c_int, [V, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


DECLARED_SYMBOLS = (
    "mythos_version",
    "mythos_last_error",
    "mythos_device_count",
    "mythos_debug_set",
    "mythos_debug_get",
    "mythos_oxdna_param_count",
    "mythos_oxdna_param_name",
    "mythos_oxdna_create",
    "mythos_oxdna_destroy",
    "mythos_oxdna_set_params",
    "mythos_oxdna_set_pseq",
    "mythos_oxdna_set_nucleotide_types",
    "mythos_oxdna_set_neighbors",
    "mythos_oxdna_build_neighbors",
    "mythos_oxdna_neighbor_stats",
    "mythos_oxdna_energy",
    "mythos_oxdna_energy_obs",
    "mythos_oxdna_energy_dpseq",
    "mythos_observables_create",
    "mythos_observables_destroy",
    "mythos_observables_width",
    "mythos_observables_eval",
    "mythos_langevin_create",
    "mythos_langevin_destroy",
    "mythos_langevin_set_neighbor_policy",
    "mythos_langevin_init_momenta",
    "mythos_langevin_run",
    "mythos_langevin_load",
    "mythos_langevin_advance",
    "mythos_langevin_store",
    "mythos_langevin_get_step",
    "mythos_langevin_set_step",
    "mythos_langevin_set_seed",
    "mythos_langevin_last_kernel_ms",
    "mythos_langevin_last_recoveries",
    "mythos_langevin_last_rebuilds",
    "mythos_langevin_set_option",
    "mythos_langevin_set_timing",
    "mythos_martini_langevin_set_timing",
    "mythos_oxdna_read_trajectory",
    "mythos_oxdna_write_trajectory",
    "mythos_martini_create",
    "mythos_martini_destroy",
    "mythos_martini_energy",
    "mythos_martini_param_grads",
    "mythos_martini_langevin_create",
    "mythos_martini_langevin_destroy",
    "mythos_martini_langevin_set_neighbor_policy",
    "mythos_martini_langevin_set_inner_list",
    "mythos_martini_langevin_init_velocities",
    "mythos_martini_langevin_run",
    "mythos_martini_langevin_load",
    "mythos_martini_langevin_advance",
    "mythos_martini_langevin_store",
    "mythos_martini_langevin_get_step",
    "mythos_martini_langevin_last_rebuilds",
    "mythos_martini_langevin_last_kernel_ms",
    "mythos_martini_langevin_neighbor_stats",
    "mythos_martini_langevin_get_rows",
    "mythos_martini_langevin_last_recoveries",
)


def last_error() -> str:
    return load().mythos_last_error().decode()


def check(rc: int, what: str = "") -> None:
    """Raise on a negative mythos_status (error conventions of SURVEY.md section 8b)."""
    if rc == 0:
        return
    msg = f"{what}: {last_error()} (status {rc})"
    if rc == -1:
        raise ValueError(msg)
    if rc == -6:
        raise FloatingPointError(msg)
    raise MythosHipError(msg)


# keys of mythos_debug_set (include/mythos_hip.h: enum mythos_debug_key) - test and diagnostic switches
DEBUG_KEYS = {"cell_bucket_cap": 0, "energy_list_cap": 1, "md_segment": 2, "md_overflow_at": 3, "md_dense": 4, "md_items_big": 5, "md_lanes": 6}


def debug_set(key: str, value: int) -> None:
    """Process-wide test / diagnostic switch of the library; 0 restores the default."""
    check(load().mythos_debug_set(DEBUG_KEYS[key], int(value)), f"debug_set({key})")


def debug_get(key: str) -> int:
    return int(load().mythos_debug_get(DEBUG_KEYS[key]))


_PARAM_NAMES: list[str] = []


def param_names() -> list[str]:
    """Names of the flat parameter vector, in its order (a property of the library: read once)."""
    if not _PARAM_NAMES:
        lib = load()
        _PARAM_NAMES.extend(lib.mythos_oxdna_param_name(i).decode() for i in range(lib.mythos_oxdna_param_count()))
    return list(_PARAM_NAMES)


def device_count() -> int:
    return int(load().mythos_device_count())


def ptr(t):
    """Device (or host) data pointer of a torch tensor / numpy array as c_void_p (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)
