"""Named-array bundles exchanged with the C++ test drivers (tests/cpp/bundle_io.h reads / writes the same layout)."""
import struct

import numpy as np

_DT = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.int64): 3, np.dtype(np.float64): 4}
_RT = {v: k for k, v in _DT.items()}


def save(path, arrays: dict):
    with open(path, "wb") as f:
        f.write(b"AOSB" + struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype not in _DT:
                if a.dtype.kind == "f":
                    a = a.astype(np.float32)
                elif a.dtype.kind in "iub":
                    a = a.astype(np.int32)
                else:
                    raise TypeError(f"{name}: {a.dtype}")
            nb = name.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<BB", _DT[a.dtype], a.ndim))
            f.write(struct.pack("<%dQ" % a.ndim, *a.shape))
            f.write(a.tobytes())


def load(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"AOSB"
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            (nl,) = struct.unpack("<H", f.read(2))
            name = f.read(nl).decode()
            dt, nd = struct.unpack("<BB", f.read(2))
            dims = struct.unpack("<%dQ" % nd, f.read(8 * nd)) if nd else ()
            dtype = _RT[dt]
            cnt = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(cnt * dtype.itemsize), dtype).reshape(dims).copy()
    return out
