"""hipGraph replay of the training update and of the acting step.

One update is ~1110 short kernel launches (two sequential scans of small GEMMs); launched eagerly
from Python it is host-bound.  The launch sequence is static (fixed shapes, no host reads, RNG and
Adam step counters live in device memory), so it is captured once into HIP graphs and replayed:
MI355X-native replacement for the reference's (inert) torch.compile switch (dreamer.py:75-79).

Graph segments, with the data-parallel collectives kept OUTSIDE capture (eager RCCL calls between
replays):   [world model fwd+bwd] -> all-reduce -> [WM clip+Adam | behaviour fwd+bwd] -> all-reduce x2
-> [actor / critic clip+Adam].  On a single rank the all-reduces are no-ops.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import _dev, _lib, ops


class CaptureRefused(RuntimeError):
    """The HIP runtime (or torch) refused stream capture / graph instantiation.  The ONLY condition under which the
    runners fall back to eager launches; shape bugs, a non-zero return code of a dv3 kernel (DV3Error) or an
    asynchronous HIP fault propagate to the caller."""


# What the HIP runtime says when it refuses stream capture or graph instantiation (hipErrorStreamCapture*,
# hipErrorCapturedEvent, hipErrorGraphExecUpdateFailure and the texts hipGetErrorString gives them).  Anything else --
# an autograd error that happens to mention a "graph", a shape bug -- is NOT a refusal and propagates.
_CAPTURE_REFUSALS = (
    "hiperrorstreamcapture", "cudaerrorstreamcapture", "hiperrorcapturedevent", "hiperrorgraphexecupdatefailure",
    "operation not permitted when stream is capturing", "operation failed due to a previous error during capture",
    "operation would result in a merge of separate capture sequences",
    "capture was not ended in the same stream as it began", "capturing stream has unjoined work",
    "dependency created on uncaptured work in another stream",
    "operation would make the legacy stream depend on a capturing blocking stream",
    "operation not permitted on an event last recorded in a capturing stream",
    "attempt to terminate a thread-local capture sequence from another thread",
    "hipgraphinstantiate", "cudagraphinstantiate", "graph instantiation",
)


def _is_capture_refusal(e: BaseException) -> bool:
    msg = str(e).lower()
    return any(t in msg for t in _CAPTURE_REFUSALS)


def _capture(graph, fn, **kw):
    """Run fn under stream capture into `graph`; translate a refusal of the capture itself into CaptureRefused."""
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local", **kw):
            fn()
    except _lib.DV3Error:
        raise
    except RuntimeError as e:
        if _is_capture_refusal(e):
            raise CaptureRefused(f"{type(e).__name__}: {e}") from e
        raise


class SegmentRecorder:
    """Capture of one phase as a SEQUENCE of graphs, one per lane (engine.Lanes): [main] -> fork -> [side] | [scan] ->
    join -> [main] ...  The code being captured calls cut(lane) where the lane changes (engine.SideStream does); each
    segment is an ordinary single-stream capture, on the lane's own CU-masked stream for "side" / "scan", so that its
    kernels are dispatched by that stream's hardware queue when the graph is launched there."""

    def __init__(self, pool, device):
        from . import engine

        self.pool, self.device = pool, device
        self.lanes = engine.Lanes.get(device)
        # lanes only when the segments will be launched on the Lanes' own whole-chip stream (engine.SideStream)
        self.on_whole = self.lanes is not None and torch.cuda.current_stream(device) == self.lanes.streams["whole"]
        self.segments = []  # (lane, graph)
        self._cur = None
        self._flat = _dev.flag("DV3_LANES_FLAT", False)  # dev: the same cuts, every segment on the caller's stream
        self._late_fork = _dev.flag("DV3_LANES_LATE_FORK", True)
        self._late_join = _dev.flag("DV3_LANES_LATE_JOIN", True)
        self._mark2 = torch.cuda.Event(blocking=_dev.flag("DV3_LANES_BLOCKING_WAIT", True))
        blocking = _dev.flag("DV3_LANES_BLOCKING_WAIT", True)  # the host sleeps in its two waits instead of spinning
        self._mark = torch.cuda.Event(blocking=blocking)

    def _begin(self, lane):
        if self._flat:
            lane = "main"
        g = torch.cuda.CUDAGraph()
        kw = dict(pool=self.pool, capture_error_mode="thread_local")
        if lane != "main":
            kw["stream"] = self.lanes.streams[lane]
        ctx = torch.cuda.graph(g, **kw)
        ctx.__enter__()
        self._cur = (lane, g, ctx)

    def _end(self, *exc):
        lane, g, ctx = self._cur
        self._cur = None
        ctx.__exit__(*(exc or (None, None, None)))
        if not exc or exc[0] is None:
            self.segments.append((lane, g))

    def cut(self, lane):
        self._end()
        self._begin(lane)

    def lane_sync_point(self):
        """Inside a lane segment: the host holds back the JOIN (and the main segment behind it) until the lane has got
        here -- the main queue then sits blocked beside the lane's dependent launches only for the rest of the segment."""
        if self._cur is None or self._cur[0] == "main" or not self._late_join:
            return
        lane = self._cur[0]
        self._end()
        self.segments.append(("sync_lane", None))
        self._begin(lane)

    def sync_point(self):
        """The host holds back the lane segments that follow until the GPU has got here (see replay)."""
        if any(lane == "sync" for lane, _ in self.segments):
            return
        self._end()
        self.segments.append(("sync", None))
        self._begin("main")

    def record(self, fn):
        from . import engine

        prev = engine.SideStream.recorder
        engine.SideStream.recorder = self
        try:
            self._begin("main")
            try:
                fn()
            except BaseException as e:
                if self._cur is not None:
                    try:
                        self._end(type(e), e, e.__traceback__)
                    except Exception:  # the capture is already broken: the first error is the one to report
                        pass
                raise
            self._end()
        except _lib.DV3Error:
            raise
        except RuntimeError as e:
            if _is_capture_refusal(e):
                raise CaptureRefused(f"{type(e).__name__}: {e}") from e
            raise
        finally:
            engine.SideStream.recorder = prev
        return self

    def replay(self, trace=None, on_join=None):
        """Launch the segments: consecutive lane segments run side by side, a "main" segment waits for them.
        trace (tools/wm_bench.py): a list that receives (lane, start event, end event) per segment.  on_join: called once
        the main stream has been ordered behind the lanes, before the segment that follows them is launched (the
        deferred weight gradients are complete there: UpdateRunner starts their all-reduce).

        A lane segment is NOT put on its queue while the main stream is still far from the fork: a queue whose head is a
        blocked barrier packet costs the other queue ~1.3 us per dependent launch (MI355X, r03: the segment in front of
        the fork 5.87 ms with the lanes already waiting, 5.35 without -- more than the overlap returns).  So the host
        waits at the "sync" mark (after the forward scan; the chip-filling decoder / head launches that follow give it
        ~2 ms to put the lanes in place) before it launches them.  This is the one place where the update call blocks
        the host; the reference's update reads its metrics back on every call (models.py:158-168).  The same holds the
        other way round: with the join queued behind the lanes at once, the main queue sits blocked beside the reverse
        scan's ~260 dependent launches (scan segment 2.16-2.22 ms against 1.81 alone, whatever runs on the side lane),
        so the scan is cut at three quarters ("sync_lane") and the host queues the join when the GPU is there."""
        cur = torch.cuda.current_stream(self.device)
        forked = []
        mark = mark2 = None
        last = None
        for lane, g in self.segments:
            if lane == "sync":
                if self._late_fork:
                    mark = self._mark
                    mark.record(cur)
                continue
            if lane == "sync_lane":
                mark2 = self._mark2
                mark2.record(last)
                continue
            if lane != "main" and mark is not None:
                mark.synchronize()
                mark = None
            if lane == "main":
                if mark2 is not None:
                    mark2.synchronize()
                    mark2 = None
                for s in forked:
                    cur.wait_stream(s)
                if forked and on_join is not None:
                    on_join()
                forked = []
                self._launch(g, cur, lane, trace)
            else:
                s = self.lanes.streams[lane]
                if s not in forked:
                    s.wait_stream(cur)  # nothing has been put on `cur` since the fork point
                    forked.append(s)
                with torch.cuda.stream(s):
                    self._launch(g, s, lane, trace)
                last = s
        for s in forked:
            cur.wait_stream(s)

    @staticmethod
    def _launch(g, stream, lane, trace):
        if trace is None:
            g.replay()
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        g.replay()
        b.record(stream)
        trace.append((lane, a, b))


# Schedule of the pipelined update (UpdateRunner.step_pipelined).  a_split / c_split: step index at which the rollout /
# reverse rollout leaves the side lane for the whole chip (None: all of it on the lane); defer: where the world model's
# deferred weight gradients run ("side": behind the reverse rollout on the side lane; "post" / "mid": in line on the
# whole chip).  Development switches DV3_PIPE_* (build.py --dev) override for A/B runs.
_PIPE_PLAN = dict(on=_dev.flag("DV3_PIPE", True), a_split=_dev.value("DV3_PIPE_A_SPLIT", 0) or None,
                  c_split=_dev.value("DV3_PIPE_C_SPLIT", 0) or None, defer=_dev.value("DV3_PIPE_DEFER", "auto", str),
                  late=_dev.flag("DV3_PIPE_LATE", True), mode=_dev.value("DV3_PIPE_MODE", "auto", str),
                  defer_split=_dev.value("DV3_PIPE_DEFER_SPLIT", 0) or None)


class PhaseRecorder:
    """Capture of one phase of the update (world model, or behaviour) as LABELLED hipGraph segments.  The code being
    captured calls engine.Cuts.mark(label) where a segment ends; UpdateRunner.step_pipelined then places the segments of
    two different updates side by side: the behaviour phase of update k beside the world-model phase of update k+1.

    lane_of(label) -> "main" | "scan" | "side": the stream a segment is CAPTURED on.  A hipGraph takes the CU mask of
    the stream it is launched on, so the lane is decided at replay; but the per-stream scratch buffers of dv3hip.ops
    are keyed by the capture stream, so two segments that may run at the same time are captured on different
    streams.  Labels with an "@" are optional cuts: taken only when listed in `optional`."""

    def __init__(self, pool, device, lanes, lane_of, first_label, optional=()):
        self.pool, self.device, self.lanes = pool, device, lanes
        self.lane_of, self.first, self.optional = lane_of, first_label, set(optional)
        self.segments = []  # (label, graph) in capture order
        self._cur = None

    def _begin(self, label):
        g = torch.cuda.CUDAGraph()
        kw = dict(pool=self.pool, capture_error_mode="thread_local")
        lane = self.lane_of(label)
        if lane != "main":
            kw["stream"] = self.lanes.streams[lane]
        ctx = torch.cuda.graph(g, **kw)
        ctx.__enter__()
        self._cur = (label, g, ctx)

    def _end(self, *exc):
        label, g, ctx = self._cur
        self._cur = None
        ctx.__exit__(*(exc or (None, None, None)))
        if not exc or exc[0] is None:
            self.segments.append((label, g))

    def mark(self, label):
        if "@" in label and label not in self.optional:
            return
        self._end()
        self._begin(label)

    def record(self, fn, at_end=None):
        """at_end: called inside the last segment (PhasedRng.finish_phase)."""
        from . import engine

        if engine.Cuts.recorder is not None:
            raise RuntimeError("nested PhaseRecorder")
        engine.Cuts.recorder = self
        try:
            self._begin(self.first)
            try:
                fn()
                if at_end is not None:
                    at_end()
            except BaseException as e:
                if self._cur is not None:
                    try:
                        self._end(type(e), e, e.__traceback__)
                    except Exception:  # the capture is already broken: the first error is the one to report
                        pass
                raise
            self._end()
        except _lib.DV3Error:
            raise
        except RuntimeError as e:
            if _is_capture_refusal(e):
                raise CaptureRefused(f"{type(e).__name__}: {e}") from e
            raise
        finally:
            engine.Cuts.recorder = None
        return self

    def by_label(self):
        return dict(self.segments)


class UpdateRunner:
    def __init__(self, wm, beh, use_graph: bool = True, warm: int = 2):
        self.wm, self.beh = wm, beh
        cfg = wm._config
        self.use_graph = use_graph and cfg.critic["slow_target_update"] == 1
        self.warm = warm
        self._calls = 0
        self._static: Dict[str, torch.Tensor] = {}
        self._g_wm = self._g_beh = None  # (per-lane graphs of the forward/backward, optimizer graph); (behaviour, optimizers)
        self._cap = {}  # what the captured halves returned (static tensors the replays rewrite)
        self._pool = torch.cuda.graph_pool_handle() if torch.cuda.is_available() else None
        self._m1, self._m2, self._beh_out = {}, {}, None
        self._stager = None
        self._stream = None
        self._home = None  # ("whole", stream) or ("caller", None): where the first call put the update
        self.last_metrics = {}
        self.last_post = self.last_context = self.last_data = None  # what a further behaviour (Plan2Explore) trains on
        # two-update software pipeline (step_pipelined / flush): labelled segments of both phases + their schedule
        # (on_wm(metrics), on_beh(metrics)): called right behind the optimizer graph of a phase, on the stream that graph
        # was launched on -- the caller's accumulation of the device-resident metrics is then stream-ordered with their
        # producer and needs no other queue to wait (a blocked queue costs every launch of the others ~1.3 us)
        self.metric_sinks = None
        # data parallel: all-reduce the decoder / head half of the world-model gradient as soon as the lanes have joined,
        # beside the encoder's backward (_model_cut).  OFF until it has been measured on more than one GPU: in the
        # one-rank RCCL rehearsal (DV3_BENCH_PG1=1 DV3_FORCE_ALLREDUCE=1 bench.py --plain --serial) the two asynchronous
        # collectives cost 17.58 ms per update against 16.51 with the single all-reduce (16.43 without a process group)
        self.dp_split = _dev.flag("DV3_DP_SPLIT", False)
        self._pipe = None
        self._pipe_trace = None
        self._pipe_pending = False  # a world-model phase has been issued whose behaviour phase has not
        self.pipe_plan = dict(_PIPE_PLAN)
        from . import engine

        engine.Lanes.on_teardown(self.close)

    # -- the two halves of one update -----------------------------------------------------------------
    # World-model half: [fwd+bwd as per-lane graphs] -> all-reduce -> [clip+Adam].  Behaviour half: [imagine, returns,
    # losses, backward] -> all-reduce x2 -> [clip+Adam x2].  step() runs both; WorldModel._train / ImagBehavior._train
    # reach them one at a time through train_wm() / train_behavior() (a driver written against the reference's classes).
    def _load(self, data):
        if not self._static:
            for k, v in data.items():
                self._static[k] = v.clone()
            return
        for k, v in data.items():
            self._static[k].copy_(v, non_blocking=True)

    def _replaying(self, eager):
        return not (eager or not self.use_graph or self._calls <= self.warm)

    def _wm_half(self, data, eager=False):
        self._calls += 1
        if self._replaying(eager):
            self._check_weights()
            self._load(data)
            if self._g_wm is None:
                torch.cuda.synchronize()
                try:
                    self._capture_wm()  # records only: nothing has executed yet, so fall through and replay
                except CaptureRefused as e:  # a runtime that refuses capture: keep training, launch eagerly
                    self._refused(e)
        if not self._replaying(eager):
            post, ctx, m1 = self.wm._train_eager(data)
            self._m1, self.last_post, self.last_context, self.last_data = m1, post, ctx, data
            self._sink(0, m1)
            return
        g1, ga = self._g_wm
        # (queueing the collectives only once the segment in front is over -- so that RCCL's stream does not sit blocked
        # beside it -- measured slower under a one-rank RCCL group: 16.43 vs 16.34 ms)
        mb = self.wm._model_opt.bucket
        cut = self._model_cut()
        if cut is None:
            g1.replay()
            mb.allreduce()
        else:
            # data parallel: the decoder / head half of the gradient is complete where the lanes join (its weight
            # gradients ran beside the reverse scan) -- it crosses xGMI while the encoder's backward still runs
            works = []
            g1.replay(on_join=lambda: works.append(mb.allreduce_range(cut, None, async_op=True)))
            works.append(mb.allreduce_range(0, cut if works else None, async_op=True))
            for w in works:
                if w is not None:
                    w.wait()
        ga.replay()
        self._m1, self.last_post, self.last_context, self.last_data = self._cap["m1"], self._cap["post"], self._cap["ctx"], self._static
        self._sink(0, self._m1)

    def _beh_half(self, eager=False):
        if self._replaying(eager) and self._g_beh is None:
            torch.cuda.synchronize()
            try:
                self._capture_beh()
            except CaptureRefused as e:
                self._refused(e)
        if not self._replaying(eager):
            self._beh_out = self.beh._train_eager(self.last_post, None)
            self._m2 = self._beh_out[-1]
            self._sink(1, self._m2)
            return
        gb, g3 = self._g_beh
        gb.replay()
        self.beh._actor_opt.bucket.allreduce()
        self.beh._value_opt.bucket.allreduce()  # (the return-normalisation EMA values ride in its tail)
        g3.replay()
        self._beh_out, self._m2 = self._cap["beh_out"], self._cap["beh_out"][-1]
        self._sink(1, self._m2)

    def _sink(self, which, metrics):
        if self.metric_sinks is not None and self.metric_sinks[which] is not None:
            self.metric_sinks[which](metrics)

    def _model_cut(self):
        """Where the world-model gradient bucket is cut into two all-reduces (floats): the first decoder / head
        parameter -- `heads.*` follows encoder and dynamics in WorldModel.parameters().  None on a single rank."""
        mb = self.wm._model_opt.bucket
        if not mb.distributed() or not self.dp_split:
            return None
        if "cut" not in self.__dict__:
            first = next((p for n, p in self.wm.named_parameters() if n.startswith("heads.")), None)
            off = mb.ensure().offset_of(first) if first is not None else None
            names = [n for n, _ in self.wm.named_parameters()]
            i0 = next((i for i, n in enumerate(names) if n.startswith("heads.")), len(names))
            # (only a contiguous tail can be cut off: every parameter behind the first head parameter is a head's)
            self.cut = off if off and all(n.startswith("heads.") for n in names[i0:]) else None
        return self.cut

    def _refused(self, e):
        import sys

        print(f"[dv3hip] hipGraph capture refused ({e}); falling back to eager launches", file=sys.stderr)
        self.use_graph, self._g_wm, self._g_beh = False, None, None
        torch.cuda.synchronize()  # an asynchronous HIP error surfaces here instead of being trained over

    def close(self):
        """Drop every captured graph (engine.Lanes calls this before it destroys the streams they were captured on)."""
        self._g_wm = self._g_beh = self._pipe = None
        self._cap, self._beh_out = {}, None
        self._pipe_pending = False
        self.use_graph = False

    def _check_weights(self):
        """The captured launches hold raw pointers into the three flat parameter buckets: replaying them over weights
        that have been moved since (Module.to, a rebuilt bucket) would train memory nobody reads."""
        where = tuple(b.flat.data_ptr() if b.settled() else 0 for b in
                      (self.wm._model_opt.bucket, self.beh._actor_opt.bucket, self.beh._value_opt.bucket))
        if self._g_wm is None:
            self._weights = where
        elif where != self._weights:
            raise RuntimeError("the parameters were moved after the update's launch sequence was captured (Module.to / "
                               "a rebuilt ParamBucket): build a new UpdateRunner")

    def _capture_wm(self):
        wm = self.wm
        # (capture_error_mode thread_local: with a process group alive, RCCL's watchdog thread polls events while we
        # capture; the default global mode would treat that as a capture violation)
        # the forward/backward is a sequence of graphs: its reverse scan and the weight gradients that nothing reads
        # before the optimizer run on two CU-masked lanes (engine.Lanes), everything else on the whole chip
        dev = next(iter(self._static.values())).device
        g1 = SegmentRecorder(self._pool, dev).record(lambda: wm.train_fwd_bwd(self._static))
        ga = torch.cuda.CUDAGraph()
        cap = self._cap

        def opt():
            cap["post"], cap["ctx"], cap["m1"] = wm.train_opt(allreduce=False)

        _capture(ga, opt, pool=self._pool)
        self._g_wm = (g1, ga)

    def _capture_beh(self):
        beh = self.beh
        gb, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        _capture(gb, lambda: beh.train_fwd_bwd(self.last_post), pool=self._pool)
        _capture(g3, lambda: self._cap.update(beh_out=beh.train_opt(allreduce=False)), pool=self._pool)
        self._g_beh = (gb, g3)

    def _on_launch_stream(self, fn):
        """Stream-ordered with the caller's current stream.  When that is the NULL stream the update itself runs on a stream
        of the runner's own: the CU-masked lanes are blocking streams, and beside work on the NULL stream (which
        synchronises with every blocking stream at every launch) the update took 18.4 ms instead of 16.3.  A caller on
        a stream of its own keeps the update there, in line (engine.SideStream).  Whichever it was on the first call is
        the runner's home for good -- its graphs were cut for it; a caller that changes streams later is ordered with
        the home stream explicitly."""
        s = self.launch_stream()
        if self._home is None:
            from . import engine

            cur0 = torch.cuda.current_stream()
            ln = engine.Lanes._by_dev.get(str(torch.device("cuda", cur0.device.index)))
            if s is None and ln is not None and cur0 == ln.streams["whole"]:
                s = cur0  # (the caller already is on launch_stream(), as Dreamer._train and the staged bench leg are)
            self._home = ("whole", s) if s is not None else ("caller", None)
        if self._home[0] != "whole":
            return fn()
        whole = self._home[1]
        self._stream = whole
        cur = torch.cuda.current_stream()
        foreign = cur != torch.cuda.default_stream(cur.device) and cur != whole
        # a BLOCKING stream: HIP orders it with the NULL stream by itself (its launches wait for earlier NULL-stream work,
        # later NULL-stream launches wait for it), and only when such work exists -- an explicit wait_stream pair would
        # leave a blocked barrier packet at the head of the NULL queue for the whole update (measured: 17.3 ms vs 16.4)
        if foreign:
            whole.wait_stream(cur)
        with torch.cuda.stream(whole):
            out = fn()
        if foreign:
            cur.wait_stream(whole)
        return out

    def step(self, data, eager: bool = False):
        """One full update.  data: dict of device tensors (image uint8 [B,T,64,64,3], action, reward, is_first,
        is_terminal ...)."""
        self.flush()

        def both():
            self._wm_half(data, eager)
            self._beh_half(eager)
            self.last_metrics = {**self._m1, **self._m2}

        self._on_launch_stream(both)

    # -- two-update software pipeline ----------------------------------------------------------------------
    # The reference issues its updates back to back inside one agent call -- `for _ in range(steps):
    # self._train(next(self._dataset))` (dreamer.py:95-97; 2 per call at the dmc configs, 100 at pretrain).  Within such a
    # run the behaviour phase of update k only READS the world model that update k's Adam step left behind, and the
    # world-model phase of update k+1 reads the same weights and a fresh batch: the two are independent until update
    # k+1's own Adam step.  step_pipelined() therefore issues world model k+1 BESIDE behaviour k, on the two CU-masked
    # lanes of engine.Lanes (complementary halves of the chip).  Schedule "lanes" (_pipe_iteration_lanes; taken where the
    # world model has the conv stacks): each phase on its own half from end to end -- a segment of this update's
    # launches takes only 1.5-1.8x as long on 128 compute units as on 256, so two half-chip streams of independent work
    # beat one whole-chip stream (cfg 2: 16.3 -> 13.6 ms per update).  Schedule "staged" (_pipe_iteration): only the two
    # observe scans -- chains of dependent 16-row launches that cannot use more than half of the chip -- beside the
    # behaviour's rollout / reverse rollout, everything else on the whole chip (cfg 2: 15.3; cfg 1: 11.5).  Every number is the serial
    # sequence's: the same weights are read (world-model Adam k+1 is ordered behind behaviour k's last read of them),
    # the posterior of update k is copied out before scan k+1 overwrites it ("bh.start"), and each phase draws from
    # the Philox counters the serial order would have given it (ops.PhasedRng).  flush() issues the last behaviour
    # phase alone; anything that reads the actor or the critic (Dreamer._policy, a checkpoint) comes after a flush.
    def pipeline_available(self) -> bool:
        from . import engine

        if not self.use_graph or self._home is None or self._home[0] != "whole":
            return False
        return engine.Lanes._by_dev.get(str(torch.device("cuda", self._home[1].device.index))) is not None

    @property
    def wm_metrics(self):
        """Metrics of the last world-model phase issued (a pipelined call's last_metrics pairs them with the behaviour
        metrics of the update before)."""
        return self._m1

    @property
    def beh_metrics(self):
        return self._m2

    def pipeline_wanted(self) -> bool:
        """Plan switch on and a schedule exists for this shape (engine.RSSMEngine.pipeline_mode)."""
        return bool(self.pipe_plan.get("on", True)) and self._pipe_mode() is not None

    def _pipe_mode(self):
        mode = self.pipe_plan.get("mode", "auto")
        if mode == "auto":
            wm = self.wm
            mode = wm.dynamics.engine.pipeline_mode(bool(wm.encoder.cnn_shapes) and bool(wm.heads["decoder"].cnn_shapes))
        return mode

    def step_pipelined(self, data):
        """Like step(), but the behaviour phase of this update is issued only with the NEXT call (beside that update's
        world-model phase) or by flush().  Falls back to step() where the pipeline is not available (eager warm-up
        calls, no CU-masked streams, a caller on a stream of its own)."""
        def run():
            if not self._pipe_pending:
                # prologue: nothing to run beside -- the serial world-model half (its own per-lane graphs)
                self._wm_half(data)
                if self._replaying(False) and self.pipeline_available() and self.pipeline_wanted():
                    self._pipe_pending = True
                    self.last_metrics = dict(self._m1)
                else:
                    self._beh_half()
                    self.last_metrics = {**self._m1, **self._m2}
                return
            if self._pipe is None:
                torch.cuda.synchronize()
                try:
                    self._capture_pipe()
                except CaptureRefused as e:
                    self._refused(e)
                    self._pipe_pending = False
                    self._beh_half()
                    return self.step(data)
            if not self._pipe["entered"]:
                self._pipe_enter()
            self._calls += 1
            self._check_weights()
            if self._pipe["mode"] == "lanes":
                self._pipe_iteration_lanes(data)
            else:
                self._load(data)
                self._pipe_iteration()
            cap = self._pipe["cap"]
            self._m1, self._m2 = cap["m1"], cap["beh_out"][-1]
            self._beh_out = cap["beh_out"]
            self.last_post, self.last_context, self.last_data = cap["post"], cap["ctx"], self._static
            self.last_metrics = {**self._m1, **self._m2}

        self._on_launch_stream(run)

    def flush(self):
        """Issue the behaviour phase step_pipelined() has left pending (no-op otherwise)."""
        if not self._pipe_pending:
            return

        def run():
            self._pipe_pending = False
            if self._pipe is not None and self._pipe["entered"]:
                self._pipe_leave()
            self._beh_half()
            self.last_metrics = dict(self._m2)

        self._on_launch_stream(run)

    def _capture_pipe(self):
        """Both phases as labelled segments (PhaseRecorder) against Philox states of their own, and their optimizer
        graphs.  Records only: nothing executes."""
        import tools
        from . import engine

        wm, beh = self.wm, self.beh
        dev = next(iter(self._static.values())).device
        lanes = engine.Lanes.get(dev)
        plan = self.pipe_plan
        rng_wm, rng_beh = ops.PhasedRng(dev), ops.PhasedRng(dev)
        key_dev = wm.dynamics.W.device
        lane_wm = lambda lb: {"wm.fscan": "scan", "wm.fscan2": "scan", "wm.rscan": "scan", "wm.rscan2": "scan",
                              "wm.defer": "side"}.get(lb, "main")
        lane_beh = lambda lb: "side" if lb.startswith(("bh.A", "bh.C")) else "main"
        mode = self._pipe_mode()
        if plan.get("defer", "auto") == "auto":
            # staged: the side lane beside the reverse scan takes the reverse rollout where there is one (imag_gradient
            # dynamics / both) and the deferred weight gradients run in line on the whole chip (cfg 2: 15.3 against
            # 16.1 ms with them behind the reverse rollout on the lane); else the lane is theirs (reinforce)
            plan["defer"] = "post" if wm._config.imag_gradient in ("dynamics", "both") else "side"
        if mode == "staged" and plan["defer"] != "side":
            # (captured where it runs: ops.gemm picks its tile for the compute units of the capture stream's queue)
            lane_wm = lambda lb: {"wm.fscan": "scan", "wm.fscan2": "scan", "wm.rscan": "scan", "wm.rscan2": "scan"}.get(lb, "main")
        if mode == "lanes":
            # each phase on a lane of its own from end to end: its segments are captured on that lane's stream
            lane_wm = lambda lb: "side" if lb == "wm.defer" else "scan"  # ("wm.defer@i", the rest of them: on the scan lane)
            lane_beh = lambda lb: "side"
        opt_cuts = [f"bh.A@{plan['a_split']}"] if plan.get("a_split") else []
        opt_cuts += [f"bh.C@{plan['c_split']}"] if plan.get("c_split") else []
        cap = {}
        # separate graph pools: the two phases replay in an order other than the capture order
        pool_w, pool_b = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        # (defer_split, development switch: move the deferred launches behind that index -- WorldModel._defer_heads_from names
        # where the heads' begin -- onto the world model's own lane.  With the lane-aware weight-gradient tiles both lanes of
        # cfg 2 end within 0.03 ms of each other with ALL of them on the behaviour's lane: 13.21 ms against 13.35 / 13.43 with
        # the last three / six on the other lane)
        w_cuts = [f"wm.defer@{plan['defer_split']}"] if (plan.get("defer_split") and mode == "lanes") else []
        with tools.rng_override(key_dev, rng_wm):
            W = PhaseRecorder(pool_w, dev, lanes, lane_wm, "wm.pre", optional=w_cuts).record(
                lambda: wm.train_fwd_bwd(self._static), at_end=rng_wm.finish_phase)
            gw = torch.cuda.CUDAGraph()

            def wopt():
                cap["post"], cap["ctx"], cap["m1"] = wm.train_opt(allreduce=False)

            _capture(gw, wopt, pool=pool_w)
        with tools.rng_override(key_dev, rng_beh):
            B = PhaseRecorder(pool_b, dev, lanes, lane_beh, "bh.start", optional=opt_cuts).record(
                lambda: beh.train_fwd_bwd(cap["post"]), at_end=rng_beh.finish_phase)
            gb = torch.cuda.CUDAGraph()
            _capture(gb, lambda: cap.update(beh_out=beh.train_opt(allreduce=False)), pool=pool_b)
        stride = rng_wm.taken + rng_beh.taken
        rng_wm.stride.fill_(stride), rng_beh.stride.fill_(stride)
        mk = lambda: torch.cuda.Event(blocking=True)
        self._pipe = dict(W=W.segments, B=B.segments, wopt=gw, bopt=gb, cap=cap, rng_wm=rng_wm, rng_beh=rng_beh,
                          lanes=lanes, entered=False, shared=tools.default_rng(key_dev), mode=mode,
                          ev={k: mk() for k in ("tail", "fork1", "q1", "mid", "fork2", "q2")},
                          ev2={k: (mk() if k == "mid" else torch.cuda.Event()) for k in ("load", "start", "mid", "defer", "wopt")},
                          ring=[mk(), mk()])

    def _pipe_enter(self):
        """The serial world-model half of update k has run (shared Philox offset S + w): behaviour k draws from there,
        world model k+1 behind behaviour k's counters."""
        P = self._pipe
        sh = P["shared"]
        sh.commit()
        P["rng_beh"].state.copy_(sh.state)
        P["rng_wm"].state.copy_(sh.state)
        P["rng_wm"].state[1:2].add_(P["rng_beh"].taken)
        P["entered"] = True

    def _pipe_leave(self):
        """Back to the shared stream in front of the serial behaviour half: it continues where the pending phase draws."""
        P = self._pipe
        if P.get("lanes_running"):
            cur = torch.cuda.current_stream()
            for s in (P["lanes"].streams["scan"], P["lanes"].streams["side"]):
                cur.wait_stream(s)
            P["lanes_running"] = False
        P["shared"].state.copy_(P["rng_beh"].state)
        P["entered"] = False

    def _pipe_iteration_lanes(self, data):
        """World-model phase of update k+1 on one half of the chip, behaviour phase of update k on the other, each from
        end to end -- no stage joins.  Measured on MI355X (tools/pipe_bench.py --segments): a segment of chip-filling
        launches takes only 1.5-1.8x as long on 128 compute units as on 256 (every launch pays ~4-5 us that do not scale
        with its work), so two half-chip streams of independent work deliver ~1.25x the whole chip's serial rate.

          X (scan lane)  [load . wm.pre] -e_start-> [fwd scan . wm.mid] [rev scan . wm.post] -e_defer-> [all-reduce . Adam] -e_wopt->
          Y (side lane)  -e_wopt(k)-> [bh.start] [rollout . heads . reverse rollout . actor] [all-reduce x2 . Adam x2] -e_mid-> [wm.defer]

        Cross-lane events: the posterior of update k is copied out (bh.start) before scan k+1 overwrites it; the
        deferred weight gradients of update k+1 follow its decoder / heads (e_mid); world-model Adam k+1 follows the last
        read of the weights by behaviour k and the deferred gradients (e_defer); behaviour k+1 follows Adam k+1 (e_wopt)."""
        P = self._pipe
        W, B, ev = dict(P["W"]), dict(P["B"]), P["ev2"]
        L = P["lanes"].streams
        X, Y = L["scan"], L["side"]
        cur = torch.cuda.current_stream()
        first = not P.get("lanes_running")
        it = P["iter"] = P.get("iter", 0) + 1
        ring = P["ring"]
        ring[it % 2].synchronize()  # (the host stays at most two iterations ahead of the GPU)

        def run(segs, labels, stream):
            with torch.cuda.stream(stream):
                for lb in labels:
                    g = segs.get(lb)
                    if g is not None:
                        self._traced(lb, g)

        # the batch: its last reader in update k was the decoder loss (wm.mid).  The HOST waits for that point, not the
        # stream: a queue whose head is a blocked barrier packet costs every launch of the two lanes ~1.3 us, and the host
        # is up to two updates ahead here
        if not first:
            ev["mid"].synchronize()
        self._load(data)
        ev["load"].record(cur)
        X.wait_event(ev["load"])
        if first:
            Y.wait_event(ev["load"])  # (behind the serial world-model half the prologue has queued on this stream)
        else:
            Y.wait_event(ev["wopt"])
        # (host order: both lanes get their first segment at once, then the behaviour lane its whole phase -- its rollout
        # must not wait for the host to have queued the world model's eight graphs: 0.4 ms of idle lane otherwise)
        run(B, ["bh.start"], Y)
        ev["start"].record(Y)
        run(W, ["wm.pre"], X)
        run(B, [lb for lb in B if lb != "bh.start"], Y)  # (capture order: rollout, heads, reverse rollout, actor)
        X.wait_event(ev["start"])
        run(W, ["wm.fscan", "wm.fscan2", "wm.mid"], X)
        ev["mid"].record(X)
        run(W, ["wm.rscan", "wm.rscan2", "wm.post"] + [lb for lb in W if lb.startswith("wm.defer@")], X)
        with torch.cuda.stream(Y):
            self.beh._actor_opt.bucket.allreduce()
            self.beh._value_opt.bucket.allreduce()
            self._traced("bh.opt", P["bopt"])
            self._sink(1, P["cap"]["beh_out"][-1])
        Y.wait_event(ev["mid"])
        run(W, ["wm.defer"], Y)
        ev["defer"].record(Y)
        X.wait_event(ev["defer"])
        with torch.cuda.stream(X):
            self.wm._model_opt.bucket.allreduce()
            self._traced("wm.opt", P["wopt"])
            self._sink(0, P["cap"]["m1"])
        ev["wopt"].record(X)
        ring[it % 2].record(X)
        P["lanes_running"] = True

    def _traced(self, label, g):
        trace = self._pipe_trace
        if trace is None:
            g.replay()
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        trace.append((label, a, b))

    def _pipe_iteration(self):
        """World-model phase of update k+1 beside the behaviour phase of update k.

          main   [bh.start . wm.pre]      [wm.mid . bh.B]                  [wm.post . bh.D] -> all-reduce -> [Adam x3]
          scan              [forward scan k+1]           [reverse scan k+1]
          side              [rollout k       ]           [reverse rollout k . deferred weight gradients k+1]

        The host queues a lane segment LATE (once the GPU is within a segment of the fork) and the join LATE (once the
        scan is three quarters through): a queue whose head is a blocked barrier packet costs every dependent launch
        of the other queues ~1.3 us ("Compute-unit lanes")."""
        P = self._pipe
        W, B, ev, plan = dict(P["W"]), dict(P["B"]), P["ev"], self.pipe_plan
        L = P["lanes"].streams
        scan, side = L["scan"], L["side"]
        cur = torch.cuda.current_stream()
        late = plan.get("late", True)

        trace = self._pipe_trace  # tools/pipe_bench.py: a list that receives (label, start event, end event)

        def run(segs, labels, stream=None):
            with torch.cuda.stream(stream if stream is not None else cur):
                for lb in labels:
                    g = segs.get(lb)
                    if g is None:
                        continue
                    if trace is None:
                        g.replay()
                        continue
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    g.replay()
                    b.record()
                    trace.append((lb, a, b))

        a_rest = [lb for lb in B if lb.startswith("bh.A@")]  # (the part of the rollout the plan keeps off the lane)
        c_rest = [lb for lb in B if lb.startswith("bh.C@")]
        # ---- stage 1 (whole chip): the posterior of update k leaves the scan's buffers; encoder of update k+1
        ev["tail"].record(cur)
        run(B, ["bh.start"])
        run(W, ["wm.pre"])
        ev["fork1"].record(cur)
        if late:
            ev["tail"].synchronize()  # (the GPU has finished the previous call's tail: the lanes' wait below is short)
        scan.wait_event(ev["fork1"]), side.wait_event(ev["fork1"])
        # ---- stage 2 (lanes): forward scan k+1 | rollout k
        run(W, ["wm.fscan"], scan)
        ev["q1"].record(scan)
        run(W, ["wm.fscan2"], scan)
        run(B, ["bh.A"], side)
        if late and "wm.fscan2" in W:
            ev["q1"].synchronize()  # (three quarters through the scan: queue the join and what follows it)
        cur.wait_stream(scan), cur.wait_stream(side)
        # ---- stage 3 (whole chip): decoder / heads of update k+1, heads / returns / critic of update k
        run(W, ["wm.mid"])
        ev["mid"].record(cur)
        run(B, a_rest + ["bh.B", "bh.B@critic", "bh.B@dyn"])
        if plan.get("defer") == "mid":
            run(W, ["wm.defer"])
        ev["fork2"].record(cur)
        if late:
            ev["mid"].synchronize()
        scan.wait_event(ev["fork2"]), side.wait_event(ev["fork2"])
        # ---- stage 4 (lanes): reverse scan k+1 | reverse rollout k (+ the deferred weight gradients of update k+1)
        run(W, ["wm.rscan"], scan)
        ev["q2"].record(scan)
        run(W, ["wm.rscan2"], scan)
        run(B, ["bh.C"], side)
        if plan.get("defer", "side") == "side":
            run(W, ["wm.defer"], side)
        if late and "wm.rscan2" in W:
            ev["q2"].synchronize()
        cur.wait_stream(scan), cur.wait_stream(side)
        # ---- stage 5 (whole chip): encoder backward + scan weight gradients k+1, actor backward k, optimizers.  Data
        # parallel: the model gradient crosses xGMI on RCCL's stream beside them, in two halves -- the decoder / head
        # half as soon as the lanes have joined (where the deferred weight gradients ran on the side lane)
        mb = self.wm._model_opt.bucket
        cut = self._model_cut() if plan.get("defer", "side") == "side" else None
        works = [mb.allreduce_range(cut, None, async_op=True)] if cut is not None else []
        run(W, ["wm.post"])
        if plan.get("defer") == "post":
            run(W, ["wm.defer"])
        works.append(mb.allreduce_range(0, cut, async_op=True))
        run(B, c_rest + ["bh.D"])
        for w in works:
            if w is not None:
                w.wait()
        P["wopt"].replay()
        self._sink(0, P["cap"]["m1"])
        self.beh._actor_opt.bucket.allreduce()
        self.beh._value_opt.bucket.allreduce()
        P["bopt"].replay()
        self._sink(1, P["cap"]["beh_out"][-1])

    # -- the reference's two calls ------------------------------------------------------------------------
    def train_wm(self, data):
        """WorldModel._train(data) (models.py:108-171) through the runner -> (post, context, metrics).  data: what the
        reference hands its world model (a dict of host arrays from the replay sampler), or device tensors."""
        from models import _wrap, DeviceScalar

        self.flush()

        def run():
            host = all(not isinstance(v, torch.Tensor) for v in data.values())
            if not (self.use_graph and self._calls + 1 > self.warm):
                staged = data  # (an eager call stages for itself)
            elif host:
                if self._stager is None:
                    from .staging import BatchStager

                    self._stager = BatchStager(self.wm._config.device)
                staged = self._stager.stage(data)
            else:
                staged = {k: (v if k == "image" else v.to(torch.float32)) for k, v in data.items()}
            self._wm_half(staged)
            # the captured metrics live in buffers the next replay overwrites: the caller gets a snapshot (one launch)
            fresh = _wrap({k: (v._t if isinstance(v, DeviceScalar) else v) for k, v in self._m1.items()})
            return self.last_post, self.last_context, fresh

        return self._on_launch_stream(run)

    def owns(self, start) -> bool:
        """Is `start` the posterior the last train_wm() handed out (what the reference passes on, dreamer.py:195)?"""
        lp = self.last_post
        return lp is not None and (start is lp or (isinstance(start, dict) and set(start) == set(lp) and all(
            isinstance(start[k], torch.Tensor) and start[k].data_ptr() == lp[k].data_ptr()
            and start[k].shape == lp[k].shape for k in lp)))

    def train_behavior(self):
        """ImagBehavior._train(post, reward head) (models.py:327-446) on the posterior of the last train_wm()."""
        from models import _wrap, DeviceScalar

        def run():
            self._beh_half()
            out = self._beh_out
            fresh = _wrap({k: (v._t if isinstance(v, DeviceScalar) else v) for k, v in out[-1].items()})
            return tuple(out[:-1]) + (fresh,)

        return self._on_launch_stream(run)

    def launch_stream(self):
        """The stream step() will issue the update on when the caller's current stream is the NULL stream (None
        otherwise: the current stream itself).  Work that feeds the update (BatchStager uploads) belongs on the same
        stream -- `with torch.cuda.stream(runner.launch_stream() or torch.cuda.current_stream()): ...` -- so that no
        second queue sits blocked beside the update's dependent launches."""
        from . import engine

        cur = torch.cuda.current_stream()
        if self._home is not None and self._home[0] == "whole":
            return self._home[1] if cur != self._home[1] else None
        own = (self._home is None and engine.SideStream.lanes and not engine.SideStream.plain
               and cur == torch.cuda.default_stream(cur.device) and _dev.flag("DV3_RUNNER_OWN_STREAM", True))
        lanes = engine.Lanes.get(cur.device) if own else None
        return lanes.whole_chip_stream() if lanes is not None else None


def weight_buckets(root):
    """Every flat parameter bucket (params.ParamBucket) of the tools.Optimizer instances hanging off root's modules."""
    import tools

    out = []
    for m in root.modules():
        for v in vars(m).values():
            if isinstance(v, tools.Optimizer) and all(v.bucket is not b for b in out):
                out.append(v.bucket)
    return out


class PolicyRunner:
    """hipGraph replay of the acting step (Dreamer._policy: preprocess -> encoder -> obs_step -> actor; SURVEY 8(f)
    N1).  Eager, the step is ~45 launches and host-bound (0.8 ms at 1-16 envs); its launch sequence is static for a
    given (number of envs, training flag), so it is captured once per signature and replayed (0.23 ms at 1 env).  Inputs
    are copied into static device buffers (host observations through one pinned staging buffer per key), the outputs
    of a replay are packed into one buffer by one launch inside the graph and leave it with a single clone, so what
    the caller gets back are fresh tensors exactly as from the eager path; the carried state the graph reads IS that
    packed buffer, so a caller that hands back what it was given pays no copy-in (see _is_last_output)."""

    def __init__(self, agent):
        self.agent = agent
        self._sig = {}
        self._buckets = weight_buckets(agent)
        self._weights = None

    def _weights_where(self):
        return tuple((b.flat.data_ptr(), b.settled()) if b.flat is not None else (0, False) for b in self._buckets)

    def _build(self, obs, state, training):
        ag = self.agent
        dyn = ag._wm.dynamics
        dev = torch.device(ag._config.device)
        n = len(obs["is_first"])
        S, D, De, A = dyn._stoch, dyn._discrete, dyn._deter, dyn._num_actions
        st = dict(obs={}, pin={})
        for k, v in obs.items():
            t = torch.as_tensor(v)
            dt = torch.uint8 if (k == "image" and t.dtype == torch.uint8) else torch.float32
            st["obs"][k] = torch.zeros(tuple(t.shape), dtype=dt, device=dev)
            st["pin"][k] = torch.zeros(tuple(t.shape), dtype=dt).pin_memory()
        # the host fills the pinned buffers through numpy views: torch's CPU copy_ goes parallel above 32 K elements,
        # and a pool of spinning OpenMP workers is what an env loop under a CPU quota cannot afford (r04: one acting
        # step in twelve took ~100 ms, the cgroup's throttling period; 0.29 ms median either way)
        st["pin_np"] = {k: v.numpy() for k, v in st["pin"].items()}
        # one flat buffer holds the step's outputs [action | logprob | stoch | deter | logit]; the carried state the
        # graph READS is the same memory (views of the previous step's outputs): when the caller hands back exactly
        # what the last step returned, nothing has to be copied in (see _load)
        sizes = [n * A, n, n * S * D, n * De, n * S * D]
        st["packed"] = torch.zeros(sum(sizes), device=dev)
        st["sizes"] = sizes
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + sz)
        pk = st["packed"]
        st["action"] = pk[offs[0]:offs[1]].view(n, A)
        st["state"] = {"stoch": pk[offs[2]:offs[3]].view(n, S, D), "deter": pk[offs[3]:offs[4]].view(n, De),
                       "logit": pk[offs[4]:offs[5]].view(n, S, D)}
        st["offs"] = offs

        def core():
            out, (latent, action) = ag._policy_eager(st["obs"], (st["state"], st["action"]), training)
            # every read of the carried state precedes this launch in stream order
            ops.concat_flat([out["action"], out["logprob"].reshape(-1), latent["stoch"], latent["deter"], latent["logit"]],
                            st["packed"])

        self._load(st, obs, state)
        import tools

        rng = tools.default_rng(dev)
        saved = rng.state.clone()
        core()  # warm: code objects, workspaces
        rng.state.copy_(saved)  # the warm call is not a step: leave the Philox stream where the caller had it
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        _capture(g, core)
        st["graph"] = g
        return st

    @staticmethod
    def _load(st, obs, state):
        ev = st.get("h2d_done")
        if ev is not None:
            ev.synchronize()  # the pinned buffers are rewritten below: their previous upload must be over
        for k, v in obs.items():
            if isinstance(v, torch.Tensor):
                st["obs"][k].copy_(v, non_blocking=True)
            else:
                np.copyto(st["pin_np"][k], np.asarray(v), casting="unsafe")
                st["obs"][k].copy_(st["pin"][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["h2d_done"] = ev
        if state is None:
            # obs_step with prev_state None == every env resets (networks.py:176-180): force is_first
            st["obs"]["is_first"].fill_(1.0)
            st["action"].zero_()
        elif not PolicyRunner._is_last_output(st, state):
            latent, action = state
            for k in ("stoch", "deter", "logit"):
                st["state"][k].copy_(latent[k], non_blocking=True)
            st["action"].copy_(action, non_blocking=True)

    @staticmethod
    def _is_last_output(st, state):
        """True when `state` is, untouched, what the previous step() returned: its tensors are views of the clone of
        the packed buffer taken then (same storage, no in-place write since), so the packed buffer still holds the
        same values and the graph can read them where they are."""
        last = st.get("last")
        if last is None:
            return False
        flat, version = last
        latent, action = state
        offs = st["offs"]
        want = ((action, offs[0]), (latent.get("stoch"), offs[2]), (latent.get("deter"), offs[3]),
                (latent.get("logit"), offs[4]))
        for t, off in want:
            if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or not t.is_contiguous():
                return False
            if t.data_ptr() != flat.data_ptr() + 4 * off or t._version != version:
                return False
        return flat._version == version

    def step(self, obs, state, training):
        n = len(obs["is_first"])
        # the static buffers are typed and shaped from the observations the graph was built on: another image dtype
        # (float vs uint8) or shape is another signature, never a silent conversion into the old buffers
        sig = tuple((k, tuple(torch.as_tensor(obs[k]).shape), str(torch.as_tensor(obs[k]).dtype)) for k in sorted(obs))
        key = (n, bool(training), bool(training and self.agent._exploring()), sig)
        # the graphs hold raw pointers to the weights: those must sit in their optimizer's flat bucket BEFORE a capture
        # (the buckets are built lazily, by the first update -- and the reference's loop evaluates before it trains,
        # dreamer.py:534-549), and a graph captured over weights that have moved since (Module.to) is dropped
        where = self._weights_where()
        if where != self._weights or not all(ok for _, ok in where):
            if self._sig:
                torch.cuda.synchronize()
                self._sig.clear()
            for b in self._buckets:
                b.ensure()
            self._weights = self._weights_where()
        st = self._sig.get(key)
        if st is None:
            st = self._build(obs, state, training)
            self._sig[key] = st
        self._load(st, obs, state)
        st["graph"].replay()
        flat = st["packed"].clone()
        st["last"] = (flat, flat._version)
        dyn = self.agent._wm.dynamics
        S, D, De, A = dyn._stoch, dyn._discrete, dyn._deter, dyn._num_actions
        o = st["offs"]
        action, logprob = flat[o[0]:o[1]].view(n, A), flat[o[1]:o[2]].view(n)
        latent = {"stoch": flat[o[2]:o[3]].view(n, S, D), "deter": flat[o[3]:o[4]].view(n, De),
                  "logit": flat[o[4]:o[5]].view(n, S, D)}
        return {"action": action, "logprob": logprob}, (latent, action)
