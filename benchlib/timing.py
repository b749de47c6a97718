"""The timed region of bench.py and the measured choice of the Chunk loop's lane count.

Round 4's driver-measured headline (30.1 ms per frame) was timed with the library's per-kernel event bracketing ON (two timing events around every kernel of
both lanes: host time per launch, and the kernels of a lane separated on the device) while every secondary line was timed with it off.  The timed region now
refuses to run with the bracketing on; per-kernel times come from separate short passes after it."""
import time


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else 0.0


class ProfilingEnabledError(RuntimeError):
    pass


def timed_region(lib, step, drain, sync, steps):
    """EXACTLY `steps` calls of step(), then drain(), bracketed by sync() (barrier + device synchronise) on both sides -> seconds on this rank.
    Raises if the library's per-kernel event bracketing is on (nrf_profile_is_enabled): a throughput number must not carry it."""
    if lib.nrf_profile_is_enabled():
        raise ProfilingEnabledError("the timed region runs with nrf_profile_enable(0): per-kernel events cost host time per launch and separate the kernels on the device")
    out = None
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    drain()                                   # the last frame's collective, inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if lib.nrf_profile_is_enabled():
        raise ProfilingEnabledError("nrf_profile_enable was switched on during the timed region")
    return dt, out


def choose_lanes(lib, set_lanes, step, drain, sync, agree_max, candidates=(1, 2), frames=6, rounds=2):
    """Which lane count of the Chunk loop is faster ON THIS BOX: `rounds` interleaved blocks of `frames` whole synchronised steps per candidate (1, 2, 1, 2: a
    drifting clock hits both), the best block per candidate counts; `agree_max(x)` returns the maximum of x over the ranks (every rank must take the same
    decision: each step holds a collective).  Leaves the winner set; returns (lanes, {lanes: ms per step})."""
    best = {c: float("inf") for c in candidates}
    for _ in range(rounds):
        for c in candidates:
            set_lanes(c)
            step(); drain()                                   # first call with a new lane count sizes the workspace
            dt, _ = timed_region(lib, step, drain, sync, frames)
            best[c] = min(best[c], agree_max(dt / frames))
    # a tie (within 1 %) goes to the smaller count: fewer streams, same time
    order = sorted(candidates)
    win = order[0]
    for c in order[1:]:
        if best[c] < 0.99 * best[win]:
            win = c
    set_lanes(win)
    return win, {int(c): best[c] * 1e3 for c in candidates}
