"""Keep the launching thread where it is.

A training step here is ~3000 kernel launches issued by ONE Python thread whose time per step (~44 ms) is the same as
the GPU's, so every migration of that thread across a 256-core host shows up in the step time: unpinned 45.7-48.7 ms per
step (52 on a busy box), pinned to 2-4 cores 44.2-44.9 ms (measured with taskset on the same box, five runs each).
`pin_host_threads` narrows the affinity of the calling thread -- and of every thread created after it (the autograd
engine, the coordinate-prefetch worker, HIP's and RCCL's helpers) -- to a small block of the CPUs the process is allowed to
use, a different block per local rank (six CPUs since round 2: the prefetch worker is a third busy thread)."""
import os


def pin_host_threads(local_rank=0, width=6, stride=8):
    """Returns (original affinity set, pinned set), or (None, None) where the OS has no affinity call.
    CG3D_HOST_PIN=0 switches it off; CG3D_HOST_PIN="a-b" (a cpu list, e.g. "8-11") overrides the choice."""
    if not hasattr(os, "sched_setaffinity"):
        return None, None
    spec = os.environ.get("CG3D_HOST_PIN", "")
    if spec == "0":
        return None, None
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) <= width:
        return set(allowed), set(allowed)
    if spec:
        want = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            want.update(range(int(a), int(b or a) + 1))
        pick = sorted(want & set(allowed)) or allowed[:width]
    else:
        # blocks of `width` CPUs, `stride` apart, so that neighbouring ranks do not share a core complex
        nlocal = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
        base = 8 if len(allowed) >= 64 else 0            # leave the first cores to the OS's own work
        for st in (stride, width):
            nblocks = (len(allowed) - base - width) // st + 1
            if nblocks >= nlocal:
                break
        else:
            return set(allowed), set(allowed)            # not enough CPUs for a block per rank: leave the scheduler alone
        start = base + (int(local_rank) % nblocks) * st
        pick = allowed[start:start + width]
    os.sched_setaffinity(0, set(pick))
    return set(allowed), set(pick)
