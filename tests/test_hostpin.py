"""not gpu: cagroup3d_amd/hostpin.py picks a distinct CPU block per local rank and leaves the scheduler alone when the
process is not allowed enough CPUs (os.sched_* mocked)."""
import os
from unittest import mock

from cagroup3d_amd import hostpin


def _layout(ncpu, nlocal, env=None):
    got = []
    for r in range(nlocal):
        e = {"LOCAL_WORLD_SIZE": str(nlocal)}
        e.update(env or {})
        with mock.patch.object(os, "sched_getaffinity", lambda pid: set(range(ncpu)), create=True), \
                mock.patch.object(os, "sched_setaffinity", lambda pid, s: got.append(sorted(s)), create=True), \
                mock.patch.dict(os.environ, e):
            os.environ.pop("CG3D_HOST_PIN", None) if not env else None
            hostpin.pin_host_threads(r)
    return got


def test_blocks_are_distinct_per_rank():
    got = _layout(256, 8)
    assert got[0] == [8, 9, 10, 11, 12, 13] and got[7] == [64, 65, 66, 67, 68, 69]
    flat = [c for b in got for c in b]
    assert len(flat) == len(set(flat)) == 48
    got = _layout(60, 8)                       # not enough room for the wide stride: blocks back to back
    assert len(got) == 8 and len({c for b in got for c in b}) == 48


def test_scarce_cpus_are_left_alone():
    assert _layout(16, 8) == []                # 8 ranks x 6 CPUs do not fit: no pinning
    assert _layout(6, 1) == []                 # nothing to narrow


def test_overrides():
    assert _layout(64, 1, {"CG3D_HOST_PIN": "0"}) == []
    assert _layout(64, 1, {"CG3D_HOST_PIN": "20-21,30"}) == [[20, 21, 30]]


def test_threads_started_after_the_pin_stay_inside_the_ranks_block():
    """The coordinate-prefetch worker (and every helper thread the runtime starts later) is created AFTER bench.py / train.py pin
    the process: it inherits the rank's CPU block, so eight ranks' issuing threads and workers never share a core.  Real
    affinity calls, on whatever CPUs this process may use (skipped where there are too few to narrow anything)."""
    import threading
    import pytest
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no affinity calls on this OS")
    allowed = os.sched_getaffinity(0)
    if len(allowed) < 16:
        pytest.skip("too few CPUs to pin a block")
    blocks = []
    try:
        for r in (0, 1):
            with mock.patch.dict(os.environ, {"LOCAL_WORLD_SIZE": "2"}):
                os.environ.pop("CG3D_HOST_PIN", None)
                os.sched_setaffinity(0, allowed)
                _, pinned = hostpin.pin_host_threads(r)
            seen = []
            t = threading.Thread(target=lambda: seen.append(os.sched_getaffinity(threading.get_native_id())), name="cg3d-coordinate-prefetch")
            t.start()
            t.join()
            assert seen[0] == pinned and len(pinned) == 6
            blocks.append(pinned)
        assert not (blocks[0] & blocks[1])
    finally:
        os.sched_setaffinity(0, allowed)
