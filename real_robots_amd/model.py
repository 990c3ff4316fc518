"""Reader of the compiled model blob (tools/compile_model.py) for host-side code (kinematics, IK, tests)."""
import struct

import numpy as np

from . import _native as nat

_cache = None


def load_model():
    """dict name -> numpy array (read-only views into the decompressed blob)."""
    global _cache
    if _cache is not None:
        return _cache
    b = nat.model_blob()
    magic, n, _ = struct.unpack_from('<8sII', b, 0)
    assert magic == b'RRMODEL1', "bad model blob"
    out = {}
    for i in range(n):
        name, dt, nd, s0, s1, s2, s3, off, nb = struct.unpack_from('<32sII4IQQ', b, 16 + i * 72)
        name = name.split(b'\0')[0].decode()
        np_dt, sz = [(np.float32, 4), (np.int32, 4), (np.uint8, 1)][dt]
        out[name] = np.frombuffer(b, dtype=np_dt, count=nb // sz, offset=off).reshape([s0, s1, s2, s3][:nd])
    _cache = out
    return out
