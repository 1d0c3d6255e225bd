"""ctypes binding of libclimb_hip.so.  `include/climb_hip.h` is the single source of truth: prototypes are parsed
from it, so the Python side cannot drift from the C ABI.  There is NO fallback: if the library is missing the
product path raises."""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "climb_hip.h")
# Two builds of the same sources and ABI: bf16 operands (BASELINE configs[1], the default) and IEEE-half operands (8x smaller operand
# rounding at the same MFMA rate; the engine then scales the loss gradient).  A process holds ONE of them: chosen by CLIMB_AMD_H16, or by
# the first engine that asks (select_h16); CLIMB_AMD_LIB overrides the path (developer builds of the same ABI).
_LIB_FILES = {"bf16": "libclimb_hip.so", "fp16": "libclimb_hip_f16.so"}
_h16_choice = os.environ.get("CLIMB_AMD_H16") or None
LIB_PATH = os.environ.get("CLIMB_AMD_LIB") or os.path.join(_HERE, "csrc", _LIB_FILES[_h16_choice or "bf16"])


def select_h16(name: str):
    """Ask for the library build with 16-bit operand type `name` ("bf16" / "fp16").  Fails loudly if the process already loaded the other."""
    global _h16_choice, LIB_PATH
    if name not in _LIB_FILES:
        raise ValueError(f"unknown 16-bit operand type {name!r}")
    if _lib is not None:
        if h16() != name:
            raise RuntimeError(f"climb_amd: this process loaded the {h16()} build of the HIP library ({LIB_PATH}); a {name} engine needs the other "
                               f"one. Set CLIMB_AMD_H16={name} (or create the {name} model first): one process, one 16-bit operand type.")
        return
    if _h16_choice is not None and _h16_choice != name and not os.environ.get("CLIMB_AMD_LIB"):
        raise RuntimeError(f"climb_amd: CLIMB_AMD_H16={_h16_choice} but a {name} engine was requested")
    _h16_choice = name
    if not os.environ.get("CLIMB_AMD_LIB"):
        LIB_PATH = os.path.join(_HERE, "csrc", _LIB_FILES[name])


def h16() -> str:
    """16-bit operand type of the loaded library."""
    return load().climb_h16().decode()


def torch_h16():
    import torch
    return torch.float16 if h16() == "fp16" else torch.bfloat16


_PROTO = re.compile(r"^\s*(int|const char\*)\s+(climb_\w+)\s*\(([^)]*)\)\s*;", re.M)


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[str]]]:
    """name -> (return type, [argument C types])."""
    with open(path) as f:
        txt = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    out = {}
    for ret, name, args in _PROTO.findall(txt):
        args = args.strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                types.append("ptr" if "*" in a else a.split()[-2] if len(a.split()) > 1 else a)
        out[name] = (ret, types)
    return out


_CT = {"ptr": ctypes.c_void_p, "long": ctypes.c_long, "int": ctypes.c_int, "float": ctypes.c_float}
_lib = None
_protos = None


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"climb_amd: HIP library not built ({LIB_PATH} missing). Run `python -m climb_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the device path.")
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, types) in _protos.items():
        fn = getattr(lib, name)   # AttributeError here == header/library mismatch
        fn.restype = ctypes.c_int if ret == "int" else ctypes.c_char_p
        fn.argtypes = [_CT[t] for t in types]
    _lib = lib
    # developer knob for A/B measurements: CLIMB_AMD_OPTIONS="7=0,5=1" -> climb_set_option(7, 0); climb_set_option(5, 1)
    for kv in filter(None, os.environ.get("CLIMB_AMD_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        if lib.climb_set_option(int(k), int(v)) != 0:
            raise RuntimeError(f"CLIMB_AMD_OPTIONS: climb_set_option({k}, {v}) rejected")
    return lib


def error_string(code: int) -> str:
    s = load().climb_error_string(code)
    return s.decode() if s else f"error {code}"


def _conv(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a


def call(name: str, *args):
    """Invoke an `int climb_*` launcher; tensors are passed as raw device pointers; non-zero -> RuntimeError."""
    fn = getattr(load(), name)
    rc = fn(*[_conv(a) for a in args])
    if rc != 0:
        raise RuntimeError(f"{name} failed: {error_string(rc)} (code {rc})")


def query(name: str) -> int:
    return getattr(load(), name)()


def query_arg(name: str, *args) -> int:
    """an `int f(int, ...)` entry point whose return value is the answer, not an error code (climb_get_option)"""
    return int(getattr(load(), name)(*args))
