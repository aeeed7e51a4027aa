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
