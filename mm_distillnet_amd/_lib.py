"""ctypes binding of libmmdistill_hip.so — the only way the Python host reaches the HIP kernels.

Signatures are parsed from include/mmdistill.h (the C-ABI contract), so the binding cannot drift from
the header.  There is NO fallback: if the library is missing the import of any compute entry point
raises, and every call raises on a non-zero status.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMD_LIB") or os.path.join(PKG, "libmmdistill_hip.so")      # MMD_LIB: another build of the same library (A/B timing)
HEADER = os.path.join(os.path.dirname(PKG), "include", "mmdistill.h")

_CT = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong,
       "unsigned long long": ctypes.c_ulonglong, "hipStream_t": ctypes.c_void_p, "double": ctypes.c_double}


def parse_header(path: str = HEADER) -> Dict[str, list]:
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    sigs = {}
    for m in re.finditer(r"\bint\s+(mmd_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if a.startswith("const char"):
                    types.append(ctypes.c_char_p)
                elif "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    t = " ".join(a.split()[:-1])
                    types.append(_CT[t])
        sigs[name] = types
    return sigs


class _Lib:
    def __init__(self, path: str = LIB_PATH):
        self._dll = None
        self._sigs = None
        self._path = path

    def load(self):
        if self._dll is None:
            if not os.path.exists(self._path):
                raise RuntimeError(
                    f"{self._path} is missing: build it with `python -m mm_distillnet_amd.build` "
                    "(there is no CPU or PyTorch fallback for the HIP path)")
            self._dll = ctypes.CDLL(self._path)
            self._sigs = parse_header()
            for name, types in self._sigs.items():
                fn = getattr(self._dll, name)
                fn.argtypes = types
                fn.restype = ctypes.c_int
        return self._dll

    def symbols(self):
        self.load()
        return dict(self._sigs)


LIB = _Lib()


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise ValueError("HIP entry points take device tensors")
        if not x.is_contiguous():
            raise ValueError("HIP entry points take contiguous tensors")
        return x.data_ptr()
    return x


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


# Work-skipping switches (timing experiments: results are WRONG with any of them).  They take effect only with the second, explicit opt-in
# MMD_DEV=1; a switch set without it is an error at import (a leaked variable must not silently change a training or bench job), with it the
# import prints a loud warning.  bench.py and train.py refuse to run while any of them is set (`work_skipping_switches()`).
WORK_SKIPPING = ("MMD_DEV_SKIP_CALLS",      # _lib.call: the named entry points are not launched
                 "MMD_DEV_SKIP_WG",         # engine.py: no weight-gradient launches
                 "MMD_DEV_NO_BWD",          # step.py: forward + losses only
                 "MMD_ROWS_ABL")            # csrc/pw_rows.hip: ablated row-slab kernel (no stores / no fetch / one k group)


def work_skipping_switches(env=None) -> list:
    """Names of the work-skipping dev switches present (non-empty) in the environment."""
    env = os.environ if env is None else env
    return [k for k in WORK_SKIPPING if env.get(k)]


def dev_switch(name: str) -> str:
    """Value of a work-skipping switch, honoured only under MMD_DEV=1 (checked at import: set without it raises)."""
    return os.environ.get(name, "") if os.environ.get("MMD_DEV") == "1" else ""


_set = work_skipping_switches()
if _set:
    import sys as _sys
    if os.environ.get("MMD_DEV") != "1":
        raise RuntimeError("work-skipping dev switches %s are set without MMD_DEV=1: unset them (results would be wrong), or opt in "
                           "explicitly with MMD_DEV=1 for a timing experiment" % _set)
    print("\n" + "!" * 100 + "\nmm_distillnet_amd: MMD_DEV=1 with work-skipping switches %s - launches are being SKIPPED, every result of "
          "this process is WRONG (timing experiments only)\n" % _set + "!" * 100 + "\n", file=_sys.stderr, flush=True)
_DEV_SKIP = frozenset(v for v in dev_switch("MMD_DEV_SKIP_CALLS").split(",") if v)


def call(name: str, *args):
    """Invoke `name` with tensors -> device pointers, appending the current torch HIP stream."""
    if _DEV_SKIP and name in _DEV_SKIP:
        return 0
    dll = LIB.load()
    fn = getattr(dll, name)
    conv = [_ptr(a) for a in args]
    if len(fn.argtypes) == len(conv) + 1:
        conv.append(stream_ptr())
    rc = fn(*conv)
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}")
    return rc
