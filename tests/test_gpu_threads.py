"""include/x266hip.h: a context belongs to one host thread, any number of threads may each drive their own context on the same
device at the same time ("thread-safe per context").  Here several threads do exactly that through the host-pointer
batch calls (which run the three-stream staging pipeline and its helper download thread) and the device-pointer calls, all at
once, and every result must equal the oracle's.  ctypes releases the GIL for the duration of a foreign call, so the calls really
overlap."""
import threading

import numpy as np
import pytest

import x266_amd

pytestmark = pytest.mark.gpu


def _residual(rng, shape):
    return (rng.integers(0, 256, shape) - rng.integers(0, 256, shape)).astype(np.int16)


def test_contexts_on_one_device_driven_from_several_threads(oracle):
    n_threads, rounds = 6, 3
    rng = np.random.default_rng(0x7EAD)
    jobs = []
    for t in range(n_threads):
        n = 3000 + 4111 * t                                            # > one 16 MiB chunk for most threads: the pipelined path
        x = _residual(rng, (n, 1024))
        d = rng.integers(-32768, 32768, (50000 + 977 * t, 64)).astype(np.int16)
        jobs.append((x, oracle.dct32_fwd(x, threads=8), d, oracle.satd8x8(d, threads=8)))
    errors, start = [], threading.Barrier(n_threads)

    def body(t):
        try:
            cd = x266_amd.Codec(0)                                     # this thread's own context
            x, want_z, d, want_c = jobs[t]
            start.wait()
            for r in range(rounds):
                z = cd.dct32_fwd(x)
                if not np.array_equal(z, want_z):
                    errors.append("thread %d round %d: forward DCT differs" % (t, r))
                if not np.array_equal(cd.satd8x8(d), want_c):
                    errors.append("thread %d round %d: SATD differs" % (t, r))
                back = cd.dct32_inv(z)
                if np.abs(back.astype(np.int32) - x).max() > 4:
                    errors.append("thread %d round %d: inverse does not return the residual" % (t, r))
            cd.close()
        except Exception as e:                                          # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not any(x.is_alive() for x in th), "a thread did not finish"
    assert not errors, errors


def test_contexts_created_and_freed_concurrently():
    """Context creation uploads the operand tables and builds the staging slots; freeing releases them: no shared state."""
    errors = []

    def body(t):
        try:
            for _ in range(5):
                cd = x266_amd.Codec(0)
                x = np.full((2, 1024), t - 3, np.int16)
                z = cd.dct32_fwd(x)
                if not np.array_equal(z[0], z[1]):
                    errors.append("thread %d: two equal blocks transformed differently" % t)
                cd.close()
        except Exception as e:                                          # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=body, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not any(x.is_alive() for x in th) and not errors, errors
