"""Deterministic synthetic parameters: numpy restatement of include/nrf_synth.h.

value(seed, i) = amp * (2 * u01(seed, i) - 1),  u01 = (lowbias32(i * 0x9E3779B9 + seed) >> 8) * 2**-24

There is no dataset or checkpoint in the build environment; the bench, the tests and the golden
fixtures all draw weights and hash-table entries from this closed form (tests/golden/manifest.txt
lists seed, amplitude and shape per tensor).
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def synth_u32(seed: int, n: int, start: int = 0) -> np.ndarray:
    i = np.arange(start, start + n, dtype=np.uint64)
    x = (i * np.uint64(0x9E3779B9) + np.uint64(seed & 0xFFFFFFFF)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def synth_u01(seed: int, n: int, start: int = 0) -> np.ndarray:
    return (synth_u32(seed, n, start) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def synth_sym(seed: int, shape, amp: float, offset: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    u = synth_u01(seed, n)
    v = np.float32(amp) * (np.float32(2.0) * u - np.float32(1.0))
    if offset != 0.0:
        v = v + np.float32(offset)
    return v.reshape(shape).astype(np.float32)


def load_manifest(path: str) -> dict:
    """tests/golden/manifest.txt -> {tag: [(name, seed, amp, shape), ...]} in named_parameters() order."""
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 5:
                continue
            tag, name, seed, amp = parts[0], parts[1], int(parts[2]), float(parts[3])
            shape = tuple(int(s) for s in parts[4:])
            out.setdefault(tag, []).append((name, seed, amp, shape))
    return out


def params_from_manifest(entries) -> list:
    """[(name, array)] regenerated from manifest entries."""
    return [(name, synth_sym(seed, shape, np.float32(amp))) for name, seed, amp, shape in entries]


def blob_from_manifest(entries) -> np.ndarray:
    """One flat fp32 blob in named_parameters() (== checkpoint) order."""
    return np.concatenate([a.reshape(-1) for _, a in params_from_manifest(entries)]).astype(np.float32)
