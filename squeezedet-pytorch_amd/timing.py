"""Optional per-launch HIP-event brackets (bench.py / profiling only): ``set_timer(KernelTimer())`` makes every launch wrapper
of ``ops`` / ``plans`` record a start / end event pair on the stream the kernel is launched on."""
from __future__ import annotations

import torch


class KernelTimer:
    """Optional per-launch HIP-event bracket (bench.py / profiling only).  ``select`` limits the
    bracketing to kernels whose name is in the set (None = all).  Events are recorded on the stream
    the kernels are launched on (torch's current stream)."""

    def __init__(self, select=None):
        self.select = select
        self.records = []          # (name, tag, flops, bytes, start_event, end_event)

    def wants(self, name):
        return self.select is None or name in self.select

    def summary(self, nsteps=1):
        """{name: dict(launches, ms, flops, bytes, tags)} per step -- call after a synchronize.  ``nsteps`` = number
        of identical steps that were bracketed: per (kernel, shape) the MEDIAN launch time over all its samples is
        used, so a bracket that absorbed a host stall (GPU idle between the start marker and the launch) cannot
        distort the totals."""
        groups = {}
        for name, tag, fl, by, e0, e1 in self.records:
            g = groups.setdefault((name, tag), dict(ms=[], flops=fl, bytes=by))
            g['ms'].append(e0.elapsed_time(e1))
        out = {}
        for (name, tag), g in groups.items():
            ms = sorted(g['ms'])
            med = ms[len(ms) // 2]
            per_step = len(ms) / float(nsteps)             # launches of this shape per step
            d = out.setdefault(name, dict(launches=0.0, ms=0.0, flops=0.0, bytes=0.0, tags={}))
            d['launches'] += per_step; d['ms'] += med * per_step
            d['flops'] += g['flops'] * per_step; d['bytes'] += g['bytes'] * per_step
            d['tags'][tag] = [per_step, med * per_step, g['flops'] * per_step, g['bytes'] * per_step]
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


class _Bracket:
    __slots__ = ('rec',)

    def __init__(self, name, tag, flops, nbytes):
        t = _timer
        self.rec = None
        if t is not None and t.wants(name):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            self.rec = (name, tag, flops, nbytes, e0, e1)
            # a start marker that directly follows a kernel is time-stamped while that kernel still runs
            # (measured: +40..80 us vs rocprofv3); a preceding fence marker makes it wait for the stream
            torch.cuda.Event(enable_timing=True).record()
            e0.record()

    def done(self):
        if self.rec is not None:
            self.rec[5].record()
            _timer.records.append(self.rec)


def active():
    """Whether a timer is installed (launch wrappers skip building bracket descriptions otherwise)."""
    return _timer is not None
