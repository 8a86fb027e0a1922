"""Several batches in flight on one GPU: ``PipelinedInference`` (across calls: what ``bench.py`` replays) and
``TwoHalves`` (inside ONE call of ``model(x)``), on streams that are checked to really run beside each other
(``concurrent_streams``).  A forward is a chain of kernels bound by different units with a tail after every launch; a
second batch fills what the first leaves idle (reference call sites: examples/cifar10.py:140-149)."""
from __future__ import annotations

import os
import threading
from typing import List, Optional

import torch
import torch.nn as nn

from .executor import FusedResNet


STREAM_PROBE_LOG: List[float] = []      # (chain time of candidate + chosen stream) / (one chain), per candidate probed


def concurrent_streams(device: torch.device, n: int, against=(), tries: int = 12) -> List["torch.cuda.Stream"]:
    """``n`` HIP streams whose work really runs beside each other's and beside that of the streams in ``against``.  The
    runtime maps streams onto a handful of hardware queues (four by default), two streams that share a queue run back to
    back, and which queue a stream gets depends on how many streams the process used before (`tools/exp_stream_queues.py`).
    So every candidate is checked against the streams already chosen — a chain of four ~0.1 ms spin kernels
    (`torch.cuda._sleep`) on each, submitted alternately, must take about as long as one chain — and dropped if it is
    serialised behind one of them (kernels of ONE stream carry the packet barrier bit, which orders them behind
    everything in front of them in the hardware queue, the other stream's kernels included; a single kernel per stream
    would overlap even on a shared queue); after ``tries`` candidates the best effort is kept.  Synchronises the device
    (a few milliseconds, once per executor)."""
    import time as _time
    with torch.cuda.device(device):
        if not hasattr(torch.cuda, "_sleep"):
            return [torch.cuda.Stream(device=device) for _ in range(n)]
        cyc, reps = 200_000, 4

        def spin(streams) -> float:
            torch.cuda.synchronize(device)
            t0 = _time.perf_counter()
            for _ in range(reps):
                for st in streams:
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(cyc)
            torch.cuda.synchronize(device)
            return _time.perf_counter() - t0
        fixed = list(against)
        chosen: List["torch.cuda.Stream"] = []
        if not fixed:
            chosen.append(torch.cuda.Stream(device=device))
        ref = (fixed + chosen)[0]
        spin([ref])                                    # (first use of a stream: lazy initialisation)
        one = min(spin([ref]) for _ in range(2))
        keep = []                                      # rejected candidates stay alive until the end: the pool hands out others
        for _ in range(tries):
            if len(chosen) == n:
                break
            cand = torch.cuda.Stream(device=device)
            spin([cand])
            ratio = max(min(spin([c, cand]) for _ in range(2)) for c in fixed + chosen) / one
            STREAM_PROBE_LOG.append(round(ratio, 2))
            if ratio < 1.3:
                chosen.append(cand)
            else:
                keep.append(cand)
        while len(chosen) < n:                         # nothing better found: serialised streams still give correct results
            chosen.append(keep.pop() if keep else torch.cuda.Stream(device=device))
        return chosen


class PipelinedInference:
    """Several batches in flight: ``n_streams`` graph-captured copies of the fused executor (sharing the
    model's weights, each with its own static input / activations) replayed round-robin on their own HIP
    streams.  One forward is a chain of kernels bound by different units — the stem by the matrix cores and
    LDS, the 64-channel convs by HBM, the rest by the integer ALU — so two batches interleave well:
    +20 % images/s over one stream on MI355X (three streams: +12 %).

        pipe = PipelinedInference(model, example_batch)
        for i, batch in enumerate(loader):
            pipe.input(i).copy_(batch, non_blocking=True)      # e.g. the H2D copy target
            logits = pipe.launch(i)                            # valid after pipe.wait(i) / synchronize()
    """

    def __init__(self, model: nn.Module, example: torch.Tensor, n_streams: int = 2, fresh_input: bool = False,
                 **fused_kwargs) -> None:
        """``fresh_input``: every ``launch(i, x)`` reads the tensor the caller passes (stem launch on ``x`` + HIP graph
        of the rest, ``FusedResNet.forward_fresh``) instead of a static input buffer the caller has to fill."""
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        dev = example.device
        env = os.environ.get("BNN_AMD_THROUGHPUT")   # "0" / "1": override for experiments
        fused_kwargs.setdefault("throughput_mode", n_streams > 1 if env is None else env == "1")
        self.fresh_input = fresh_input
        self.streams = concurrent_streams(dev, n_streams)
        self.engines: List[FusedResNet] = []
        cur = torch.cuda.current_stream(dev)
        for s in self.streams:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                eng = FusedResNet(model, **fused_kwargs)
                if fresh_input:
                    eng.forward_fresh(example, clone=False)      # captures the graph behind the stem for this stream
                else:
                    eng.capture(example)
                self.engines.append(eng)
        for s in self.streams:
            cur.wait_stream(s)

    def __len__(self) -> int:
        return len(self.engines)

    def input(self, i: int) -> torch.Tensor:
        return self.engines[i % len(self.engines)].static_input

    def stream(self, i: int) -> torch.cuda.Stream:
        return self.streams[i % len(self.streams)]

    def launch(self, i: int, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Replay slot ``i % n`` on its stream — with whatever its static input holds, or (``fresh_input``) reading
        ``x`` directly (the caller keeps ``x`` alive and unchanged until the slot's stream has passed the launch; work
        that produced ``x`` on another stream must be ordered before it by the caller, e.g.
        ``pipe.stream(i).wait_stream(...)``).  Returns the slot's logits buffer (overwritten by the next launch of
        the same slot)."""
        k = i % len(self.engines)
        with torch.cuda.stream(self.streams[k]):
            if self.fresh_input:
                if x is None:
                    raise ValueError("PipelinedInference(fresh_input=True).launch needs the input tensor")
                return self.engines[k].forward_fresh(x, clone=False)
            if x is not None:
                self.engines[k].static_input.copy_(x, non_blocking=True)
            return self.engines[k](self.engines[k].static_input)

    def wait(self, i: int) -> None:
        torch.cuda.current_stream(self.streams[0].device).wait_stream(self.streams[i % len(self.streams)])

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()


class TwoHalves:
    """One call, two batches in flight: ``model(x)`` with the batch cut in two halves that run on two HIP streams (the
    caller's and one more), each as "stem launch on its half of the caller's tensor + HIP graph of the rest" (``FusedResNet.forward_fresh``) on
    an executor in throughput mode.  A forward is a chain of kernels bound by different units (stem: matrix cores, the
    64-channel convs: HBM, the rest: integer ALU) with a tail after every launch; a second half-batch fills what the
    first leaves idle — what ``PipelinedInference`` does across calls, done inside one call, so that the reference's
    own ``net(x)`` loop (examples/cifar10.py:147-149) gets it without knowing.  Bit-identical logits (images are
    independent).  ``AutoFusion`` uses it from ``MIN_PIXELS`` input pixels on (``BNN_AMD_SPLIT_BATCH=0`` turns it off)."""

    # from this many input pixels on (batch x height x width): measured on MI355X at 224 x 224 (k images/s, two halves vs
    # one batch) — batch 64: 132 vs 143, batch 128: 183 vs 175, batch 256: 227 vs 213
    MIN_PIXELS = 128 * 224 * 224

    def __init__(self, model: nn.Module, device: torch.device) -> None:
        # Half 0 runs on the CALLER's stream, half 1 on one side stream.  (Until late round 4 both halves had a side
        # stream of their own.  Which hardware queues the three streams then sat on — a matter of how many streams the
        # process had used before — made the same call 1.10, 1.22, 1.26 or 1.5-1.7 ms per batch, the last slower than the
        # halves one after the other; the fast layouts were those where half 0's stream shared the caller's queue
        # (`tools/exp_stream_queues.py`).  One stream less, two cross-stream waits less, and nothing left to chance but
        # the side stream's queue, which `concurrent_streams` checks against the caller's.)
        cur = torch.cuda.current_stream(device)
        self.side = concurrent_streams(device, 1, against=[cur])[0]
        self.engines = [FusedResNet(model, throughput_mode=True)]
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self.engines.append(FusedResNet(model, throughput_mode=True))
        cur.wait_stream(self.side)
        self._lock = threading.Lock()       # one call at a time enqueues its two halves (callers on different streams)
        self._done: Optional[torch.cuda.Event] = None   # the previous call has read both halves' output buffers

    @staticmethod
    def wanted(x: torch.Tensor) -> bool:
        return (x.shape[0] >= 2 and x.shape[0] * x.shape[2] * x.shape[3] >= TwoHalves.MIN_PIXELS
                and os.environ.get("BNN_AMD_SPLIT_BATCH", "1") != "0")

    def _streams(self, device):
        return torch.cuda.current_stream(device), self.side

    # Share of the batch that runs on the CALLER's stream.  The side half is enqueued first, and the two stems cannot run
    # side by side (a stem fills the register files): the half that starts second finishes ~one stem later, and nothing
    # of the NEXT call may start before both are done (the caller's tensors are ordered on the caller's stream).  The
    # side half therefore gets fewer images: 136 + 120 of 256 measured best (k images/s, 20 steps / sustained: 128 + 128
    # 224.6 / 240.2, 136 + 120 239.3 / 248.2, 144 + 112 221.9 / 232.3; BNN_AMD_SPLIT_SHARE overrides).  Three or four parts on
    # three / four streams: 230.4 / 235.8 and 230.8 / 236.6 against 243.1 / 247.5 — smaller batches per launch cost more
    # than the shorter first stem gives back.
    SHARE_CUR = float(os.environ.get("BNN_AMD_SPLIT_SHARE", "0.53"))

    def _split(self, n: int) -> int:
        return min(n - 1, max(1, int(round(n * self.SHARE_CUR / 8.0)) * 8 if n >= 64 else (n + 1) // 2))

    def captured(self, x: torch.Tensor) -> bool:
        h = self._split(x.shape[0])
        keys = [((n,) + tuple(x.shape[1:]), st.cuda_stream) for n, st in zip((h, x.shape[0] - h), self._streams(x.device))]
        return all(k in e._split for k, e in zip(keys, self.engines))

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        dev = x.device
        cur, side = self._streams(dev)
        h = self._split(x.shape[0])
        with self._lock:
            if self._done is not None:           # a caller on ANOTHER stream may still be reading the halves' static
                cur.wait_event(self._done)       # output buffers (its torch.cat): overwrite them only behind it
            side.wait_stream(cur)                # the caller's tensor is ready on the caller's stream
            with torch.cuda.stream(side):        # (the side half first: it is under way while the host issues the other)
                y1 = self.engines[1].forward_fresh(x[h:], clone=False)
            y0 = self.engines[0].forward_fresh(x[:h], clone=False)
            cur.wait_stream(side)                # (also orders the caller's later reuse of x behind both halves)
            out = torch.cat((y0, y1), 0)
            self._done = torch.cuda.Event()
            self._done.record(cur)
        return out
