"""Closed-form deterministic tensor generator (TEST INFRASTRUCTURE).

Golden fixtures under tests/golden/ store only inputs/outputs; the model weights they were produced with are
re-generated from this integer hash, so multi-megabyte weights (geometry_embedding_mlp.0.weight is 2048 x 128N,
reference vhoi/models.py:266) never need committing. Pure numpy uint64 arithmetic: independent of the torch RNG
and of the torch version.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def uniform01(name: str, shape, seed: int = 0) -> np.ndarray:
    """float64 uniforms in [0, 1), a pure function of (name, seed, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode('utf-8'))) ^ (np.uint64(seed) << np.uint64(32))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        bits = _mix(_mix(idx ^ key) + key)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def uniform(name: str, shape, lo: float, hi: float, seed: int = 0) -> np.ndarray:
    return (lo + (hi - lo) * uniform01(name, shape, seed)).astype(np.float32)


def normal(name: str, shape, std: float = 1.0, seed: int = 0) -> np.ndarray:
    """Box-Muller on two hashed uniforms."""
    u1 = uniform01(name + '#a', shape, seed)
    u2 = uniform01(name + '#b', shape, seed)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (std * z).astype(np.float32)


def fill_state_dict(shapes: dict, seed: int = 0, gain: float = 1.0) -> dict:
    """Deterministic values for every entry of a state_dict given its {name: shape} map.

    Weights follow the fan-in scaling torch uses by default (uniform +-1/sqrt(fan_in)) times `gain`, so activations
    stay O(1) through the stack; BatchNorm buffers get non-trivial values so eval-mode is a real test.
    """
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith('num_batches_tracked'):
            out[name] = np.array(3, dtype=np.int64)
        elif name.endswith('running_mean'):
            out[name] = uniform(name, shape, -0.2, 0.2, seed)
        elif name.endswith('running_var'):
            out[name] = uniform(name, shape, 0.5, 1.5, seed)
        elif '.bn.weight' in name:
            out[name] = uniform(name, shape, 0.7, 1.3, seed)
        elif '.bn.bias' in name:
            out[name] = uniform(name, shape, -0.2, 0.2, seed)
        else:
            if len(shape) >= 2:
                fan_in = int(np.prod(shape[1:]))
            else:
                fan_in = 16
            if '_rnn' in name:  # GRU / GRUCell: torch scales by 1/sqrt(hidden)
                fan_in = shape[0] // 3 if 'weight' in name else shape[0] // 3
            if name == 'geometry_embedding_gcn.weight':
                fan_in = shape[1]
            a = gain / np.sqrt(max(fan_in, 1))
            out[name] = uniform(name, shape, -a, a, seed)
    return out
