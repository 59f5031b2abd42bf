"""ctypes binding of ``libamid_hip.so`` (the C ABI declared in ``include/amid_hip.h``).

The prototypes are parsed from the header itself, so the binding cannot drift
from the declared ABI, and :func:`declared_symbols` is what the CPU-side test
uses to check that the built library exports every declared entry point.

There is no fallback: if the shared library is missing or a symbol cannot be
resolved, importing the compute path raises (``AmidLibraryError``).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMID_LIB_PATH") or os.path.join(_HERE, "libamid_hip.so")      # (AMID_LIB_PATH: diagnostic builds)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "amid_hip.h")


class AmidLibraryError(RuntimeError):
    pass


class AmidError(RuntimeError):
    def __init__(self, fn: str, code: int, text: str):
        super().__init__(f"{fn} failed with code {code}: {text}")
        self.fn, self.code = fn, code


_SCALARS = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "unsigned": ctypes.c_uint,
    "unsigned int": ctypes.c_uint,
}


def _ctype_of(decl: str):
    """C parameter/return declaration (without the name) -> ctypes type."""
    d = decl.replace("const", " ").strip()
    d = re.sub(r"\s+", " ", d)
    if d == "char*":
        return ctypes.c_char_p
    if "*" in d:
        return ctypes.c_void_p
    if d == "void":
        return None
    return _SCALARS[d]


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # drop comments
    text = re.sub(r"//[^\n]*", " ", text)
    protos: Dict[str, Tuple[object, List[object]]] = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(amid_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)([A-Za-z_]\w*)$", a, flags=re.S)   # strip the parameter name
                argtypes.append(_ctype_of(mm.group(1)))
        protos[name] = (_ctype_of(ret), argtypes)
    return protos


def declared_symbols() -> List[str]:
    return sorted(parse_header())


class _Lib:
    def __init__(self) -> None:
        if not os.path.exists(LIB_PATH):
            raise AmidLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C amid_amd/csrc`).  There is no CPU fallback for the amid_amd compute path.")
        try:
            self._dll = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise AmidLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        self._fn = {}
        self.timer = None          # optional KernelTimer (bench.py): HIP events around every launch
        for name, (restype, argtypes) in parse_header().items():
            try:
                f = getattr(self._dll, name)
            except AttributeError as e:
                raise AmidLibraryError(f"{LIB_PATH} does not export {name} declared in include/amid_hip.h") from e
            f.restype = restype
            f.argtypes = argtypes
            self._fn[name] = f

    def raw(self, name: str):
        return self._fn[name]

    def call(self, name: str, *args) -> None:
        """Call an int-returning entry point; raise AmidError on a non-zero code."""
        if self.timer is not None and name not in _UNTIMED:
            return self.timer.timed_call(self, name, args)
        code = self._fn[name](*args)
        if code != 0:
            text = self._fn["amid_error_string"](code)
            raise AmidError(name, code, text.decode() if text else "?")

    def value(self, name: str, *args):
        """Call an entry point that returns a plain value (sizes, counts)."""
        return self._fn[name](*args)


_UNTIMED = {"amid_event_create", "amid_event_record", "amid_event_sync", "amid_event_elapsed_ms", "amid_event_destroy",
            "amid_graph_capture_begin", "amid_graph_capture_end", "amid_graph_launch", "amid_graph_destroy", "amid_step_state_pack",
            "amid_reduce_entry_pack"}


class KernelTimer:
    """Brackets every kernel-launching C-ABI call with HIP events recorded on the stream the kernels
    run on (the last argument of every launching entry point); durations are read after a sync."""

    def __init__(self, only=None):
        self.records = []      # (name, start_event, stop_event)
        self.only = only       # optional set of entry names: the others are launched without events (a queue that stays full)

    def timed_call(self, L: "_Lib", name: str, args) -> None:
        if self.only is not None and name not in self.only:
            code = L._fn[name](*args)
            if code != 0:
                text = L._fn["amid_error_string"](code)
                raise AmidError(name, code, text.decode() if text else "?")
            return
        stream = args[-1]
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        L._fn["amid_event_create"](ctypes.byref(e0))
        L._fn["amid_event_create"](ctypes.byref(e1))
        L._fn["amid_event_record"](e0, stream)
        code = L._fn[name](*args)
        L._fn["amid_event_record"](e1, stream)
        self.records.append((name, e0, e1))
        if code != 0:
            text = L._fn["amid_error_string"](code)
            raise AmidError(name, code, text.decode() if text else "?")

    def collect(self, L: "_Lib") -> Dict[str, List[float]]:
        """name -> list of durations in ms (call order preserved); destroys the events."""
        out: Dict[str, List[float]] = {}
        for name, e0, e1 in self.records:
            L._fn["amid_event_sync"](e1)
            ms = ctypes.c_float()
            L._fn["amid_event_elapsed_ms"](e0, e1, ctypes.byref(ms))
            out.setdefault(name, []).append(float(ms.value))
            L._fn["amid_event_destroy"](e0)
            L._fn["amid_event_destroy"](e1)
        self.records = []
        return out


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def ptr_array(ptrs) -> ctypes.Array:
    """Host array of device pointers for the ``const float* const*`` parameters."""
    arr = (ctypes.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
