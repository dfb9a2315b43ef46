"""Non-executing reader for pickle files that hold exactly one NumPy ndarray.

The reference's TX replay block is configured with a ``.pckl`` file holding the IQ buffer
(`gr-utsa_ofdm/python/TxSignalTransmitter.py:22-24` calls ``pickle.load``).  Unpickling runs
code chosen by the file, so this module does NOT unpickle: it walks the opcode stream with
``pickletools.genops`` (a disassembler, nothing is imported or called on behalf of the file) and
interprets only the dozen opcodes a plain-ndarray pickle uses, with a whitelist of four global
names that are *recognised*, never resolved.  Anything else raises ``UnsafePickleError``.
"""
from __future__ import annotations

import pickletools

import numpy as np


class UnsafePickleError(ValueError):
    pass


class _Global:
    def __init__(self, name):
        self.name = name


class _Array:
    def __init__(self):
        self.state = None


class _DType:
    def __init__(self, code):
        self.code = code
        self.order = "="


class _Mark:
    pass


_ALLOWED = {
    "numpy.core.multiarray _reconstruct", "numpy._core.multiarray _reconstruct",
    "numpy ndarray", "numpy dtype", "_codecs encode",
    "numpy.core.numeric _frombuffer", "numpy._core.numeric _frombuffer",      # protocol-5 ndarray pickles
}
_DTYPES = {"c16", "c8", "f8", "f4", "i8", "i4", "i2", "i1", "u8", "u4", "u2", "u1", "b1"}


def _reduce(fn, args):
    if not isinstance(fn, _Global):
        raise UnsafePickleError("REDUCE on a non-whitelisted callable")
    if fn.name.endswith("_reconstruct"):
        return _Array()
    if fn.name.endswith("_frombuffer"):
        if not (isinstance(args, tuple) and len(args) == 4 and isinstance(args[0], (bytes, bytearray))
                and isinstance(args[1], _DType) and isinstance(args[2], tuple) and args[3] in ("C", "F")):
            raise UnsafePickleError("unexpected _frombuffer arguments")
        arr = _Array()
        arr.state = (1, args[2], args[1], args[3] == "F", bytes(args[0]))
        return arr
    if fn.name == "numpy dtype":
        if not (isinstance(args, tuple) and args and isinstance(args[0], str) and args[0] in _DTYPES):
            raise UnsafePickleError("dtype %r not allowed" % (args,))
        return _DType(args[0])
    if fn.name == "_codecs encode":
        if not (len(args) == 2 and isinstance(args[0], str) and args[1] == "latin1"):
            raise UnsafePickleError("unexpected _codecs.encode arguments")
        return args[0].encode("latin1")
    raise UnsafePickleError("REDUCE of %s not allowed" % fn.name)


def loads_ndarray(data: bytes) -> np.ndarray:
    stack, memo = [], {}
    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            break
        if n == "GLOBAL":
            if arg not in _ALLOWED:
                raise UnsafePickleError("global %r not allowed" % arg)
            stack.append(_Global(arg))
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            mod = stack.pop()
            full = "%s %s" % (mod, name)
            if full not in _ALLOWED:
                raise UnsafePickleError("global %r not allowed" % full)
            stack.append(_Global(full))
        elif n in ("BININT", "BININT1", "BININT2", "LONG1", "BINUNICODE", "SHORT_BINUNICODE",
                   "BINUNICODE8", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8", "BYTEARRAY8"):
            stack.append(arg)
        elif n in ("BINSTRING", "SHORT_BINSTRING"):       # python-2 era raw str payloads
            stack.append(arg.encode("latin1") if isinstance(arg, str) else arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "MARK":
            stack.append(_Mark())
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(n[-1])
            items = tuple(stack[-k:])
            del stack[-k:]
            stack.append(items)
        elif n == "TUPLE":
            i = len(stack) - 1
            while not isinstance(stack[i], _Mark):
                i -= 1
            items = tuple(stack[i + 1:])
            del stack[i:]
            stack.append(items)
        elif n in ("BINPUT", "LONG_BINPUT"):
            memo[arg] = stack[-1]
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            stack.append(_reduce(fn, args))
        elif n == "BUILD":
            state = stack.pop()
            obj = stack[-1]
            if isinstance(obj, _Array):
                obj.state = state
            elif isinstance(obj, _DType):
                if not (isinstance(state, tuple) and len(state) >= 2 and state[1] in ("<", "|", "=", ">")):
                    raise UnsafePickleError("unexpected dtype state")
                obj.order = state[1]
            else:
                raise UnsafePickleError("BUILD on unexpected object")
        else:
            raise UnsafePickleError("opcode %s not allowed in an ndarray pickle" % n)
    if len(stack) != 1 or not isinstance(stack[0], _Array) or stack[0].state is None:
        raise UnsafePickleError("file does not hold exactly one ndarray")
    st = stack[0].state
    if not (isinstance(st, tuple) and len(st) == 5):
        raise UnsafePickleError("unexpected ndarray state")
    _ver, shape, dt, fortran, raw = st
    if not (isinstance(dt, _DType) and isinstance(raw, (bytes, bytearray))
            and isinstance(shape, tuple) and all(isinstance(s, int) and s >= 0 for s in shape)):
        raise UnsafePickleError("unexpected ndarray state contents")
    order = dt.order if dt.order in ("<", ">") else ""
    dtype = np.dtype(order + dt.code)
    count = int(np.prod(shape)) if shape else 1
    if count * dtype.itemsize != len(raw):
        raise UnsafePickleError("payload size does not match shape/dtype")
    arr = np.frombuffer(bytes(raw), dtype=dtype).reshape(shape, order="F" if fortran else "C")
    return arr.astype(dtype.newbyteorder("=")).copy()


def load_ndarray(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return loads_ndarray(f.read())
