"""Explicit forward/backward engine of the world-model training hot path.

No autograd: every backward below is hand-derived and launches libdv3hip kernels into
pre-allocated workspaces, so one whole update is a fixed sequence of kernel launches that a
hipGraph can replay.  Sequential work (the two scans) does only what must be sequential; all
weight gradients and everything that does not feed the recurrence is batched over time:

  observe (networks.RSSM.observe -> obs_step, networks.py:127-206): per step only
      blend -> img_in -> GRU -> obs_out(deter part) -> obs_stat -> sample
  runs in the scan (M = batch rows); the embed half of the obs_out Linear (the largest GEMM of
  obs_step), the whole prior head (img_out -> ims_stat -> sample) and every wgrad run once over
  all T*B rows.  Activations are time-major [T,B,...] so each step is a contiguous row block.

  imagine (models.ImagBehavior._imagine, models.py:448-548): per step actor -> sample -> img_step
  on M = B*T rows; the H-th successor, which the reference computes and discards
  (models.py:546), is not computed.

Parameter containers hold torch.nn.Parameters whose .grad are views into a flat gradient bucket
(dv3hip.params.ParamBucket); gradients are accumulated straight into those views.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _dev, ops

F32 = torch.float32


# Development switches (A/B runs of tools/*_bench.py): live only with a `build.py --dev` library (dv3hip/_dev.py);
# the shipped build always takes the defaults.  The non-default forms are exercised at tiny size by
# tests/test_path_gpu.py::test_dev_switch_variants.
_FUSE_BLEND = _dev.flag("DV3_FUSE_BLEND", True)  # reset blend of step t+1 as second output of the kernels of step t
_FUSE_SAMPLE = _dev.flag("DV3_FUSE_SAMPLE", True)  # sampling in the epilogue of the prior-logit GEMM
_GATHER_OBS = _dev.flag("DV3_GATHER_OBS", True)  # one-hot gather for img_in / head first layers in observe
_FUSE_SAMPLE_IN = _dev.flag("DV3_FUSE_SAMPLE_IN", True)  # observe scan: sample(t) + img_in(t+1) in one launch
_FUSE_CARRY = _dev.flag("DV3_FUSE_CARRY", True)  # reverse scan: carry + next straight-through in one launch
# observe scan: GRU gates / obs_out LayerNorm in the prologue of the few-row GEMM that consumes them (csrc/scanops.hip)
_FUSE_SCAN_ROW = _dev.flag("DV3_FUSE_SCAN_ROW", True)
_FUSE_SCAN_LN = _dev.flag("DV3_FUSE_SCAN_LN", True)  # ... forward: obs_out LayerNorm + posterior-logit GEMM
_FUSE_SCAN_LNBWD = _dev.flag("DV3_FUSE_SCAN_LNBWD", True)  # ... reverse: the two LayerNorm backward + data-gradient GEMMs
_FUSE_SCAN_CS = _dev.flag("DV3_FUSE_SCAN_CS", True)  # ... reverse: carry + straight-through + logit data-gradient GEMM
_DEFER_CONV_WGRAD = _dev.flag("DV3_DEFER_CONV_WGRAD", True)  # decoder conv weight gradients beside the reverse scan
_FUSE_SCAN_GRUBWD = _dev.flag("DV3_FUSE_SCAN_GRUBWD", True)  # ... reverse: GRU cell backward + the data-gradient GEMM of its Linear


class Lanes:
    """Two HIP streams whose hardware queues own complementary sets of compute units (csrc/streams.hip): "scan" for a
    chain of dependent few-row launches, "side" for chip-filling work nothing waits for until the chain is over.

    Measured (tools/cumask_probe.py, MI355X): the chain of 16-row GEMMs takes 7.7 us per launch on the whole chip AND on
    128 of its 256 CUs (10.3 on 64); beside 4096^3 GEMMs on the complementary mask it still takes 7.7 and the GEMMs lose
    nothing but the CUs they gave up -- whereas beside the same GEMMs on an UNMASKED second queue the chain makes no
    progress until they are over (its workgroups are placed behind theirs), which is why r01/r02 measured the plain
    second stream slower than no overlap at all.  A hipGraph launched on a masked stream inherits the mask; graph
    BRANCHES do not (they run on streams of the runtime's own), so a captured update is cut into one graph per lane
    (graph.SegmentRecorder)."""

    _by_dev: Dict[str, "Lanes"] = {}

    @classmethod
    def get(cls, device):
        """The lanes of `device`, or None when the runtime refuses CU-masked streams there (the callers then launch in
        line on one stream: the lanes are a scheduling optimisation, not part of the arithmetic)."""
        key = str(torch.device(device))
        if key not in cls._by_dev:
            try:
                cls._by_dev[key] = Lanes(device, _dev.value("DV3_LANES_SCAN_CUS", 0))
            except (RuntimeError, ValueError) as e:  # DV3Error (a HIP error code) included
                import sys

                print(f"[dv3hip] no CU-masked streams on {key} ({e}); the reverse scan and the deferred weight gradients "
                      "run in line", file=sys.stderr)
                cls._by_dev[key] = None
        return cls._by_dev[key]

    def __init__(self, device, scan_cus: int):
        from . import _lib  # (ops imports engine's siblings; keep the loader import local)
        import ctypes

        lib = _lib.load()
        with torch.cuda.device(device):
            n = ctypes.c_int()
            _lib.check(lib.dv3_device_cu_count(ctypes.byref(n)), "dv3_device_cu_count")
            n_cu = n.value
            if n_cu < 128:
                # (a partitioned device, e.g. CPX mode: the 16-row chain needs ~128 CUs to run at full speed -- 10.3 us per
                # launch on 64 against 7.7 -- so there is nothing to give away)
                raise ValueError(f"{n_cu} compute units: too few to split")
            if scan_cus <= 0:
                scan_cus = (n_cu // 2) // 8 * 8  # half of the chip: the 16-row chain runs at full speed on it (MI355X: 128)
            if not 8 <= scan_cus <= n_cu - 8:
                raise ValueError(f"scan lane of {scan_cus} CUs on a {n_cu}-CU device")
            # groups of 8 mask bits dealt out evenly: every XCD gives the same share of its CUs to each lane whether the
            # mask's bits run XCD-major or XCD-interleaved
            groups, want = n_cu // 8, scan_cus // 8
            scan_bits = [((g + 1) * want) // groups != (g * want) // groups for g in range(groups)]
            pattern = _dev.value("DV3_LANES_PATTERN", "group8", str)
            if pattern == "low":  # dev: the scan lane owns the lowest mask bits (whole XCDs if the bits run XCD-major)
                scan_bits = [g < want for g in range(groups)]
            per_bit = None
            if pattern == "mod8":  # dev: ... the bits with the lowest (bit % 8) (whole XCDs if the bits run XCD-interleaved)
                per_bit = [(i % 8) < (8 * want) // groups for i in range(n_cu)]
            words = (n_cu + 31) // 32
            self.cus = {"scan": 8 * sum(scan_bits), "side": n_cu - 8 * sum(scan_bits), "whole": n_cu}
            self._handles, self.streams = {}, {}
            for lane in ("scan", "side", "whole"):
                mask = (ctypes.c_uint32 * words)()
                for i in range(n_cu):
                    g = i // 8
                    in_scan = per_bit[i] if per_bit is not None else (scan_bits[g] if g < groups else False)
                    mine = lane == "whole" or in_scan == (lane == "scan")
                    if mine:
                        mask[i // 32] |= 1 << (i % 32)
                out = ctypes.c_ulonglong()
                _lib.check(lib.dv3_stream_create_cu_masked(words, mask, ctypes.byref(out)), "dv3_stream_create_cu_masked")
                self._handles[lane] = out.value
                self.streams[lane] = torch.cuda.ExternalStream(out.value, device=device)
                ops.LANE_STREAMS[out.value] = self.cus[lane]  # (ops.gemm picks its tile for the CUs the launch can use)
        # the runtime's own teardown of these queues at process exit (static destructors) crashes under rocprofv3: hand them
        # back while the interpreter is still alive
        import atexit

        atexit.register(self._destroy)

    _closers: list = []  # weak references to the close() of whatever holds hipGraphs captured on the lanes' streams

    @classmethod
    def on_teardown(cls, bound_method):
        """bound_method() is called before the lanes' streams are destroyed at interpreter exit: graphs captured on them
        (and the allocator blocks of their pools, which are tagged with the capturing stream) must go first -- an agent
        that is still alive as a module global would otherwise be torn down AFTER its streams (segfault at exit)."""
        import weakref

        cls._closers = [r for r in cls._closers if r() is not None]  # (runners come and go: keep only the living)
        cls._closers.append(weakref.WeakMethod(bound_method))

    def _destroy(self):
        from . import _lib

        handles, self._handles = self._handles, {}
        if not handles:
            return
        try:
            torch.cuda.synchronize()
            closers, Lanes._closers = Lanes._closers, []
            for ref in closers:
                fn = ref()
                if fn is not None:
                    fn()
            import gc

            gc.collect()
            torch.cuda.synchronize()
            lib = _lib.load()
            for h in handles.values():
                lib.dv3_stream_destroy(h)
        except Exception:  # interpreter shutdown: nothing left to report to
            pass

    def whole_chip_stream(self):
        """A blocking stream over every CU (what UpdateRunner runs an update on when its caller sits on the NULL stream)."""
        return self.streams["whole"]


class Cuts:
    """Labelled cut points of the update's launch sequence (graph.PhaseRecorder): `Cuts.mark("wm.fscan")` closes the
    hipGraph segment being captured and opens the next one under that label, so that a replay can place the segments
    of two different updates side by side -- the behaviour phase of update k beside the world-model phase of update
    k+1 (graph.UpdateRunner.step_pipelined).  A no-op outside such a capture: eager launches and the serial graphs
    never see the marks."""

    recorder = None  # graph.PhaseRecorder while UpdateRunner captures the pipelined segments

    @staticmethod
    def mark(label: str):
        rec = Cuts.recorder
        if rec is not None:
            rec.mark(label)

    @staticmethod
    def active() -> bool:
        return Cuts.recorder is not None

    @staticmethod
    def wants(label: str) -> bool:
        """Whether mark(label) would cut (optional "@" labels cut only when the recorder lists them)."""
        rec = Cuts.recorder
        if rec is None:
            return False
        return "@" not in label or label in getattr(rec, "optional", ())


class SideStream:
    """Work that is off the critical path (weight gradients) beside a latency-bound chain of few-row launches.

        side.run(fns)            # the deferred launches
        with side.chain():       # the chain they run beside
            ...
        side.join()              # both are over before anything that follows

    mode "lanes" (default): fns on the "side" lane, the chain on the "scan" lane (Lanes: complementary CU masks).  Under
    hipGraph capture the cut points are handed to graph.SegmentRecorder (one graph per lane); captured by anything else
    (tools that capture a phase in one graph) the work runs inline, because graph branches lose the masks -- and so do
    eager launches on the NULL stream, which synchronises with the (blocking) lane streams at every launch.
    mode "plain" (DV3_SIDE_STREAM=1, dev): one unmasked second stream -- measured slower than inline in r01, r02 and r03.
    mode "off" (DV3_LANES=0, dev): inline."""

    # r01 (cfg 2): plain second stream 29.8 vs 27.7 ms/update; r02 20.0 vs 18.6; r03 world model 10.69 vs 10.30 ms.
    plain = _dev.flag("DV3_SIDE_STREAM", False)
    lanes = _dev.flag("DV3_LANES", True)
    recorder = None  # graph.SegmentRecorder while UpdateRunner captures
    _streams: Dict[str, "torch.cuda.Stream"] = {}

    def __init__(self, device, lanes_pay: bool = True):
        """lanes_pay=False: the caller knows that the lanes do not pay here (RSSMEngine.lanes_pay) -- in line."""
        self.device = device
        self._forked = []
        self._mode = "plain" if SideStream.plain else ("lanes" if (SideStream.lanes and lanes_pay) else "off")
        if Cuts.active():
            # pipelined capture (graph.PhaseRecorder): the fork / chain / join points become labelled cuts and the
            # replay schedule decides which lane a segment runs on
            self._mode = "cuts"
        if self._mode not in ("off", "cuts") and SideStream.recorder is None and torch.cuda.is_current_stream_capturing():
            self._mode = "plain" if self._mode == "plain" else "off"
        if self._mode == "lanes":
            # The lanes are taken only beside the Lanes' OWN whole-chip stream (where UpdateRunner puts the update of a
            # NULL-stream caller).  Eager launches on the NULL stream itself synchronise with the (blocking) lane streams
            # at every launch (20.4 ms per eager update against 19.5 in line); beside a caller's own torch stream the
            # cross-queue dependencies can cost ~25 us each instead of ~1.3 (19.2 ms per update against 16.2 when that
            # stream was made after the masked queues, 16.4 when before -- nothing the caller should have to know).
            ln = Lanes.get(device)
            rec = SideStream.recorder
            on_whole = (rec.on_whole if rec is not None else
                        (ln is not None and torch.cuda.current_stream() == ln.streams["whole"]))
            if ln is None or not on_whole:
                self._mode = "off"
        if self._mode == "plain":
            key = str(device)
            if key not in SideStream._streams:
                SideStream._streams[key] = torch.cuda.Stream(device=device)
            self.stream = SideStream._streams[key]
        self._main = None
        self._cut = False

    @staticmethod
    def host_sync_point(lanes_pay: bool = True):
        """Marks where a captured update lets the host wait before it launches the lane segments (graph.SegmentRecorder)."""
        if not lanes_pay or Cuts.active():
            return
        rec = SideStream.recorder
        if rec is not None and rec.lanes is not None and rec.on_whole and SideStream.lanes and not SideStream.plain:
            rec.sync_point()

    @staticmethod
    def late_join_point():
        """Inside chain(): marks where a captured update lets the host wait before it queues the join and what follows it
        (graph.SegmentRecorder.lane_sync_point)."""
        if Cuts.active():
            Cuts.mark("wm.rscan2")
            return
        rec = SideStream.recorder
        if rec is not None and rec.lanes is not None and rec.on_whole and SideStream.lanes and not SideStream.plain:
            rec.lane_sync_point()

    def run(self, fns, chain: bool = True):
        """Run the deferred callables beside what follows.  chain=False: what follows fills the chip itself (the encoder
        backward), so in lanes mode the callables simply run in line."""
        if not fns:
            return
        # (the deferred callables are weight gradients: nothing reads their outputs before the optimizer and their inputs
        # stay put until they have run, so the small dense ones among them go out as ONE grid: ops.gemm_group)
        if self._mode == "cuts":
            if chain:
                Cuts.mark("wm.defer")
                self._cut = True
            with ops.gemm_group() as grp:
                for i, f in enumerate(fns):
                    if chain and i and Cuts.wants(f"wm.defer@{i}"):
                        grp.flush()
                        Cuts.mark(f"wm.defer@{i}")  # (optional cut: the schedule may share the deferred launches between lanes)
                    f()
            return
        if self._mode == "off" or (self._mode == "lanes" and not chain):
            self._run_all(fns)
            return
        if self._mode == "plain":
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._run_all(fns)
            self._forked.append(self.stream)
            return
        self._on_lane("side", fns)

    @staticmethod
    def _run_all(fns):
        with ops.gemm_group():
            for f in fns:
                f()

    def _on_lane(self, lane, fns):
        rec = SideStream.recorder
        if rec is not None:
            rec.cut(lane)
            self._cut = True
            self._run_all(fns)
            return
        ln = Lanes.get(self.device)
        if self._main is None:
            self._main = torch.cuda.current_stream()
        s = ln.streams[lane]
        s.wait_stream(self._main)
        with torch.cuda.stream(s):
            self._run_all(fns)
        self._forked.append(s)

    class _Chain:
        def __init__(self, side):
            self.side, self.ctx = side, None

        def __enter__(self):
            side = self.side
            if side._mode == "cuts":
                Cuts.mark("wm.rscan")
                side._cut = True
                return self
            if side._mode != "lanes":
                return self
            rec = SideStream.recorder
            if rec is not None:
                rec.cut("scan")
                side._cut = True
                return self
            ln = Lanes.get(side.device)
            if side._main is None:
                side._main = torch.cuda.current_stream()
            s = ln.streams["scan"]
            s.wait_stream(side._main)
            self.ctx = torch.cuda.stream(s)
            self.ctx.__enter__()
            side._forked.append(s)
            return self

        def __exit__(self, *exc):
            if self.ctx is not None:
                self.ctx.__exit__(*exc)
            return False

    def chain(self):
        return SideStream._Chain(self)

    def join(self):
        if self._mode == "cuts":
            if self._cut:
                Cuts.mark("wm.post")
            self._cut = False
            return
        rec = SideStream.recorder
        if self._mode == "lanes" and rec is not None:
            if self._cut:
                rec.cut("main")
            self._cut = False
            return
        cur = self._main if self._main is not None else torch.cuda.current_stream()
        for s in self._forked:
            cur.wait_stream(s)
        self._forked = []
        self._main = None


class Workspace:
    """Named device buffers, allocated on first use and reused (no allocation in steady state).

    A buffer is keyed by (name, shape, dtype) and is NEVER freed or replaced: captured hipGraphs hold raw pointers into
    these buffers, and the same engine serves several shapes in turn (training on B*T rows, acting on #envs rows,
    video prediction on 6 sequences) -- replacing "enc.pre0" when the row count changes would hand the old block
    back to the allocator while a graph still writes to it."""

    def __init__(self, device):
        self.device = device
        self._b: Dict[tuple, torch.Tensor] = {}

    def get(self, name, shape, dtype=F32) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        key = (name, shape, dtype)
        t = self._b.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._b[key] = t
        return t

    def zeros(self, name, shape, dtype=F32) -> torch.Tensor:
        t = self.get(name, shape, dtype)
        t.zero_()
        return t

    def zeros_many(self, name, shapes):
        """Several zero-initialised float32 buffers carved out of ONE allocation and cleared by ONE fill (offsets rounded
        up to 64 floats): the accumulation targets of the reverse observe scan are five buffers per update."""
        shapes = [tuple(int(s) for s in sh) for sh in shapes]
        sizes = []
        for sh in shapes:
            n = 1
            for d in sh:
                n *= d
            sizes.append(n)
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) // 64 * 64
        flat = self.get(name, (total,))
        flat.zero_()
        return [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._b.values())


# ---------------------------------------------------------------------------------------------
# parameter containers (plain references to nn.Parameters; .grad are bucket views)
# ---------------------------------------------------------------------------------------------
@dataclass
class PLin:
    W: torch.Tensor
    b: Optional[torch.Tensor] = None


@dataclass
class PDenseLN:
    W: torch.Tensor
    g: torch.Tensor
    b: torch.Tensor


@dataclass
class PMLP:
    layers: List[PDenseLN]
    out: Optional[PLin] = None  # mean_layer
    out2: Optional[PLin] = None  # std_layer (continuous actor)


@dataclass
class PRSSM:
    W0: torch.Tensor  # dynamics.W [1,De]
    img_in: PDenseLN
    gru: PDenseLN
    img_out: PDenseLN
    obs_out: PDenseLN
    ims: PLin
    obs: PLin


def _g(p):
    return p.grad


def v2(t: torch.Tensor, cols: int) -> torch.Tensor:
    return t.reshape(-1, cols)


# ---------------------------------------------------------------------------------------------
# Linear(no bias) -> LN -> SiLU on row blocks
# ---------------------------------------------------------------------------------------------
def dense_ln_fwd(L: PDenseLN, x1, x2, pre, mean, rstd, y, *, accumulate_pre=False):
    ops.gemm(x1, L.W, pre, A2=x2, accumulate=accumulate_pre)
    ops.ln_act_fwd(pre, L.g, L.b, y, mean, rstd, act=True)


def dense_ln_bwd_pre(L: PDenseLN, dy, pre, mean, rstd, dpre, *, wgrad: bool):
    ops.ln_act_bwd(dy, pre, L.g, L.b, mean, rstd, dpre, _g(L.g) if wgrad else None, _g(L.b) if wgrad else None,
                   act=True)


def lin_wgrad(W, dY, x1, x2=None):
    """W.grad += dY^T @ [x1|x2]."""
    k1 = x1.shape[1]
    gW = _g(W)
    ops.gemm(dY, x1, gW[:, :k1] if x2 is not None else gW, transA=True, transB=False, accumulate=True)
    if x2 is not None:
        ops.gemm(dY, x2, gW[:, k1:], transA=True, transB=False, accumulate=True)


# ---------------------------------------------------------------------------------------------
# MLP trunk + head Linears (networks.MLP, networks.py:588-681) on a row block, fwd / bwd
# ---------------------------------------------------------------------------------------------
class MLPEngine:
    """Activations live in [total_rows, .] buffers; forward() may fill a row range of them (the actor is
    evaluated step by step inside the imagination scan but back-propagated in one batch)."""

    def __init__(self, name: str, P: PMLP, ws: Workspace):
        self.name, self.P, self.ws = name, P, ws
        self.total = 0

    def _bufs(self, total):
        P, ws, nm = self.P, self.ws, self.name
        self.total = total
        acts = []
        for i, L in enumerate(P.layers):
            U = L.W.shape[0]
            acts.append((ws.get(f"{nm}.pre{i}", (total, U)), ws.get(f"{nm}.m{i}", (total,)),
                         ws.get(f"{nm}.r{i}", (total,)), ws.get(f"{nm}.y{i}", (total, U))))
        out = ws.get(f"{nm}.out", (total, P.out.W.shape[0])) if P.out is not None else None
        out2 = ws.get(f"{nm}.out2", (total, P.out2.W.shape[0])) if P.out2 is not None else None
        return acts, out, out2

    def pack_onehot(self, SD: int, defer=None):
        """Transposed copy of the first layer's stoch columns, W0[:, :SD] -> [SD, U], for the one-hot gather
        (ops.onehot_linear_ln).  Call once per update, after the optimizer step that changed W0.  defer: a list that
        receives the (src, dst) pair instead of launching (ops.transpose2d_many runs several packs as one launch)."""
        W0 = self.P.layers[0].W
        wt = self.ws.get(f"{self.name}.wt0", (SD, W0.shape[0]))
        if defer is not None:
            defer.append((W0[:, :SD], wt))
        else:
            ops.transpose2d(W0[:, :SD], wt)
        return wt

    def forward(self, x1, x2=None, *, row0=0, total=None, idx=None, D=0, head=None, base0=None):
        """x = [x1|x2] (R rows) -> (trunk output, mean_layer output, std_layer output) for rows
        [row0, row0+R) of the engine's buffers.

        idx (int32 [R,S]) + D: x1 is the exact one-hot expansion of idx ([R, S*D], the flattened stoch of
        get_feat, networks.py:154-159).  The first layer then multiplies only x2 on the MFMA path and gathers the S
        weight columns of every row (pack_onehot must have run), fused with its LayerNorm + SiLU.
        head (dict, actor only): run the last layer's LayerNorm + SiLU, both head Linears, the action sample and
        the entropy as ONE launch (ops.actor_head); keys: action, ent, noise, rng, eps_out, onehot, min_std,
        max_std, unimix, act_idx, forced, flips.
        base0 ([R, U], with idx): x2 @ W0[:, SD:]^T computed elsewhere (RSSMEngine.img_step_fwd's stacked GEMM)."""
        P = self.P
        R = x1.shape[0]
        total = R if total is None else total
        acts, out, out2 = self._bufs(total)
        rs = slice(row0, row0 + R)
        h1, h2 = x1, x2
        n_layers = len(P.layers)
        for i, ((pre, mean, rstd, y), L) in enumerate(zip(acts, P.layers)):
            last_fused = head is not None and i == n_layers - 1
            if i == 0 and idx is not None:
                SD = idx.shape[1] * D
                wt = self.ws.get(f"{self.name}.wt0", (SD, L.W.shape[0]))
                base = base0
                if base is None and x2 is not None:
                    ops.gemm(x2, L.W[:, SD:], pre[rs])
                    base = pre[rs]
                if last_fused:
                    ops.onehot_linear_ln(idx, D, wt, pre[rs], base=base)
                else:
                    ops.onehot_linear_ln(idx, D, wt, pre[rs], base=base, gamma=L.g, beta=L.b, y=y[rs], mean=mean[rs],
                                         rstd=rstd[rs])
            elif last_fused:
                ops.gemm(h1, L.W, pre[rs], A2=h2)
            else:
                dense_ln_fwd(L, h1, h2, pre[rs], mean[rs], rstd[rs], y[rs])
            h1, h2 = y[rs], None
        if head is not None:
            pre, mean, rstd, y = acts[-1]
            L = P.layers[-1]
            onehot = bool(head.get("onehot", False))
            ops.actor_head(pre[rs], L.g, L.b, y[rs], mean[rs], rstd[rs], P.out.W, P.out.b,
                           None if onehot else P.out2.W, None if onehot else P.out2.b, out[rs],
                           None if onehot else out2[rs], head["action"], head["ent"], noise=head.get("noise"),
                           rng=head.get("rng"), eps_out=head.get("eps_out"), act_idx=head.get("act_idx"),
                           forced=head.get("forced"), flips=head.get("flips"), min_std=head.get("min_std", 0.1),
                           max_std=head.get("max_std", 1.0), unimix=head.get("unimix", 0.01), onehot=onehot)
            return h1, out[rs], None if onehot else out2[rs]
        o = o2 = None
        if out is not None:
            o = out[rs]
            ops.gemm(h1, P.out.W, o, A2=h2, bias=P.out.b)
        if out2 is not None:
            o2 = out2[rs]
            ops.gemm(h1, P.out2.W, o2, bias=P.out2.b)
        return h1, o, o2

    def backward(self, x1, x2, rows: slice, dout=None, dout2=None, *, wgrad: bool, dx1=None, dx2=None,
                 acc_dx=False, dh=None, defer=None):
        """Back-propagate rows `rows` of the stored activations.  x1/x2: the inputs those rows were computed
        from (same row count).  dout/dout2: gradients of the head Linears' outputs for those rows; dh: gradient
        on the trunk output itself (head-less MLPs such as the proprio encoder).  defer: a list that receives
        the weight-gradient launches as callables instead of running them inline (see SideStream)."""
        P, ws, nm = self.P, self.ws, self.name
        acts, _, _ = self._bufs(self.total)
        R = x1.shape[0]
        n_layers = len(P.layers)
        h = acts[-1][3][rows] if n_layers else x1
        first = True
        if dh is not None:
            first = False
        else:
            dh = ws.get(f"{nm}.dh", (R, h.shape[1]))
        for lin, d in ((P.out, dout), (P.out2, dout2)):
            if lin is None or d is None:
                continue
            ops.gemm(d, lin.W, dh, transB=False, accumulate=not first)
            first = False
            if wgrad:
                def _head_wgrad(lin=lin, d=d, h=h):
                    lin_wgrad(lin.W, d, h)
                    if lin.b is not None:
                        ops.colsum(d, _g(lin.b), accumulate=True)
                (defer.append if defer is not None else (lambda f: f()))(_head_wgrad)
        if first:
            raise ValueError("MLP backward without any upstream gradient")
        dy = dh
        for i in reversed(range(n_layers)):
            L = P.layers[i]
            pre, mean, rstd, _ = acts[i]
            xin1, xin2 = (acts[i - 1][3][rows], None) if i > 0 else (x1, x2)
            dpre = ws.get(f"{nm}.dpre{i}", (R, pre.shape[1]))
            dense_ln_bwd_pre(L, dy, pre[rows], mean[rows], rstd[rows], dpre, wgrad=wgrad)
            if wgrad:
                def _wg(L=L, dpre=dpre, xin1=xin1, xin2=xin2):
                    lin_wgrad(L.W, dpre, xin1, xin2)
                (defer.append if defer is not None else (lambda f: f()))(_wg)
            if i > 0:
                dprev = ws.get(f"{nm}.dy{i - 1}", (R, xin1.shape[1]))
                ops.gemm(dpre, L.W, dprev, transB=False)
                dy = dprev
            else:
                k1 = x1.shape[1]
                if dx1 is not None:
                    ops.gemm(dpre, L.W[:, :k1] if x2 is not None else L.W, dx1, transB=False, accumulate=acc_dx)
                if dx2 is not None and x2 is not None:
                    ops.gemm(dpre, L.W[:, k1:], dx2, transB=False, accumulate=acc_dx)


# ---------------------------------------------------------------------------------------------
# RSSM scans
# ---------------------------------------------------------------------------------------------
class RSSMEngine:
    def __init__(self, P: PRSSM, ws: Workspace, *, stoch: int, discrete: int, deter: int, hidden: int,
                 num_actions: int, embed: int, unimix: float):
        self.P, self.ws = P, ws
        self.S, self.D, self.De, self.Hd, self.A, self.E = stoch, discrete, deter, hidden, num_actions, embed
        self.SD = stoch * discrete
        self.unimix = unimix

    # -- initial state (networks.py:99-123, 235-239): deter0 = tanh(W), stoch0 = mode(prior head) -----
    def init_state_fwd(self):
        P, ws = self.P, self.ws
        d0 = ws.get("init.deter", (1, self.De))
        ops.tanh_fwd(P.W0, d0)
        x0pre = ws.get("init.x0pre", (1, self.Hd))
        x0 = ws.get("init.x0", (1, self.Hd))
        m0, r0 = ws.get("init.m", (1,)), ws.get("init.r", (1,))
        dense_ln_fwd(P.img_out, d0, None, x0pre, m0, r0, x0)
        l0 = ws.get("init.logit", (1, self.SD))
        ops.gemm(x0, P.ims.W, l0, bias=P.ims.b)
        s0 = ws.get("init.stoch", (1, self.SD))
        ops.onehot_sample(l0.view(self.S, self.D), s0.view(self.S, self.D), unimix=self.unimix, mode=True,
                          idx=ws.get("init.idx", (self.S,), torch.int32))
        return s0, d0

    def init_state_bwd(self, dstoch0, ddeter0):
        """dstoch0 [SD], ddeter0 [De] (accumulated by the scan) -> grads of W0 and the prior head."""
        P, ws = self.P, self.ws
        l0, x0, x0pre, d0 = ws.get("init.logit", (1, self.SD)), ws.get("init.x0", (1, self.Hd)), \
            ws.get("init.x0pre", (1, self.Hd)), ws.get("init.deter", (1, self.De))
        m0, r0 = ws.get("init.m", (1,)), ws.get("init.r", (1,))
        dl0 = ws.get("init.dlogit", (1, self.SD))
        ops.onehot_st_bwd(l0.view(self.S, self.D), dstoch0.view(self.S, self.D), dl0.view(self.S, self.D),
                          unimix=self.unimix, mode=True)
        dx0 = ws.get("init.dx0", (1, self.Hd))
        ops.gemm(dl0, P.ims.W, dx0, transB=False)
        lin_wgrad(P.ims.W, dl0, x0)
        ops.colsum(dl0, _g(P.ims.b), accumulate=True)
        dx0pre = ws.get("init.dx0pre", (1, self.Hd))
        dense_ln_bwd_pre(P.img_out, dx0, x0pre, m0, r0, dx0pre, wgrad=True)
        lin_wgrad(P.img_out.W, dx0pre, d0)
        dd = ddeter0.view(1, self.De)
        ops.gemm(dx0pre, P.img_out.W, dd, transB=False, accumulate=True)
        ops.tanh_bwd(d0, dd, _g(P.W0), accumulate=True)

    # -- observe ------------------------------------------------------------------------------------
    def observe_fwd(self, embed_tm, action_tm, first_tm, *, q_prior=None, q_post=None, rng=None, force=None,
                    state0=None):
        """embed_tm [T,B,E], action_tm [T,B,A], first_tm [T,B] (float 0/1; row 0 is forced to 1, as
        prev_state=None does in networks.py:176-180).  Noise [T,B,S,D] ~ Exp(1) or rng state.
        force (parity tests): dict(post=[T,B,S] int32, prior=[T,B,S] int32, flips=int32[1]) teacher-forces the
        sampled classes and counts the draws that differ.  state0 = (stoch [B,SD], deter [B,De]): the carried
        state of RSSM.observe(..., state) (networks.py:127-143): step 0 then blends it with is_first[:, 0] as given
        instead of starting every row from the initial state.  Returns dict of time-major buffers."""
        P, ws = self.P, self.ws
        T, B = embed_tm.shape[0], embed_tm.shape[1]
        S, D, SD, De, Hd, A, E = self.S, self.D, self.SD, self.De, self.Hd, self.A, self.E
        self.T, self.B = T, B
        TB = T * B
        force = force or {}
        f_post, f_prior, flips = force.get("post"), force.get("prior"), force.get("flips")
        s0, d0 = self.init_state_fwd()
        first = ws.get("obs.first", (T, B))
        first.copy_(first_tm)
        if state0 is None:
            first[0].fill_(1.0)
        g = ws.get
        sin, din, ain = g("obs.sin", (T, B, SD)), g("obs.din", (T, B, De)), g("obs.ain", (T, B, A))
        x1pre, x1 = g("obs.x1pre", (T, B, Hd)), g("obs.x1", (T, B, Hd))
        m1, r1 = g("obs.m1", (T, B)), g("obs.r1", (T, B))
        gpre, mg, rg = g("obs.gpre", (T, B, 3 * De)), g("obs.mg", (T, B)), g("obs.rg", (T, B))
        deter = g("obs.deter", (T, B, De))
        x3pre, x3 = g("obs.x3pre", (T, B, Hd)), g("obs.x3", (T, B, Hd))
        m3, r3 = g("obs.m3", (T, B)), g("obs.r3", (T, B))
        post_logit, post_stoch = g("obs.post_logit", (T, B, S, D)), g("obs.post_stoch", (T, B, S, D))
        # class indices of the (one-hot) posterior samples and of the blended step inputs: the Linears that read
        # them gather weight columns instead of multiplying the one-hot (ops.onehot_linear_ln)
        post_idx, idx_in = g("obs.post_idx", (T, B, S), torch.int32), g("obs.idx_in", (T, B, S), torch.int32)
        init_idx = g("init.idx", (S,), torch.int32)
        gather = _GATHER_OBS
        if gather:
            # (a copy of its own: the imagination of the PREVIOUS update may be reading "rssm.img_in_wt" while this
            # scan runs beside it -- graph.UpdateRunner.step_pipelined)
            wt_in = self.pack_img_in(name="obs.img_in_wt")
        # embed half of obs_out for all steps at once: x3pre = embed @ W_obs[:, De:]^T
        ops.gemm(v2(embed_tm, E), P.obs_out.W[:, De:], v2(x3pre, Hd))
        # Reset blends (networks.py:183-191) off the per-step critical path: the action blend needs no state (one
        # launch for all T), step 0 starts from the initial state, and the blend of step t+1's stoch / deter is a
        # second output of the kernels that produce them at step t (fused when the vector GRU kernel applies).
        ops.reset_blend(v2(action_tm, A), None, first.view(TB), v2(ain, A))
        fuse = ((De % 256 == 0 and De <= 1024) or (De % 1024 == 0 and De <= 4096)) and _FUSE_BLEND
        fuse_in = fuse and gather and _FUSE_SAMPLE_IN and ops.sample_linear_ln_ok(S, D, Hd) and Hd % 4 == 0
        fuse_row = _FUSE_SCAN_ROW and _FUSE_SCAN_LN and B <= 64 and ops.scan_ln_gemm_ok(Hd, SD)
        Cuts.mark("wm.fscan")  # (pipelined capture: the forward scan is a lane segment of its own)
        for t in range(T):
            if T >= 16 and t == (3 * T) // 4:
                Cuts.mark("wm.fscan2")  # (... cut where the host queues the join behind it, as in the reverse scan)
            if t == 0 or not fuse:
                prev_s = post_stoch[t - 1].view(B, SD) if t > 0 else (None if state0 is None else state0[0])
                prev_d = deter[t - 1] if t > 0 else (None if state0 is None else state0[1])
                ops.obs_blend(prev_s, s0.view(SD), prev_d, d0.view(De), action_tm[t], first[t], sin[t], din[t], ain[t])
                if gather:
                    ops.onehot_to_idx(sin[t].view(B, S, D), idx_in[t].view(-1))
            nxt = fuse and t + 1 < T
            if fuse_in and t > 0:
                pass  # x1[t] came out of step t-1's fused sample + img_in launch
            elif gather:
                ops.onehot_linear_ln(idx_in[t], D, wt_in, x1pre[t], x2=ain[t], gamma=P.img_in.g, beta=P.img_in.b,
                                     y=x1[t], mean=m1[t], rstd=r1[t])
            else:
                dense_ln_fwd(P.img_in, sin[t], ain[t], x1pre[t], m1[t], r1[t], x1[t])
            ops.gemm(x1[t], P.gru.W, gpre[t], A2=din[t])
            ops.gru_fwd(gpre[t], P.gru.g, P.gru.b, din[t], deter[t], mg[t], rg[t],
                        next_blend=(first[t + 1], d0.view(De), din[t + 1]) if nxt else None)
            ops.gemm(deter[t], P.obs_out.W[:, :De], x3pre[t], accumulate=True)
            if fuse_row:  # the obs_out LayerNorm rides in the logit GEMM (5 launches per step instead of 6)
                ops.scan_ln_gemm(x3pre[t], P.obs_out.g, P.obs_out.b, x3[t], m3[t], r3[t], P.obs.W,
                                 post_logit[t].view(B, SD), bias=P.obs.b)
            else:
                ops.ln_act_fwd(x3pre[t], P.obs_out.g, P.obs_out.b, x3[t], m3[t], r3[t], act=True)
                ops.gemm(x3[t], P.obs.W, post_logit[t].view(B, SD), bias=P.obs.b)
            if fuse_in and nxt:
                # the posterior sample of this step and the img_in layer of the next one in one launch
                ops.onehot_sample_linear_ln(post_logit[t], post_stoch[t], noise=None if q_post is None else q_post[t],
                                            rng=rng, unimix=self.unimix, idx=post_idx[t].view(-1),
                                            forced=None if f_post is None else f_post[t], flips=flips,
                                            next_first=first[t + 1], init=s0.view(SD), init_idx=init_idx,
                                            next_out=sin[t + 1].view(B, S, D), next_idx=idx_in[t + 1].view(-1),
                                            WT=wt_in, x2=ain[t + 1], pre=x1pre[t + 1], gamma=P.img_in.g,
                                            beta=P.img_in.b, y=x1[t + 1], mean=m1[t + 1], rstd=r1[t + 1])
            else:
                ops.onehot_sample(post_logit[t], post_stoch[t], noise=None if q_post is None else q_post[t], rng=rng,
                                  unimix=self.unimix,
                                  next_blend=(first[t + 1], s0.view(SD), sin[t + 1].view(B, S, D), init_idx,
                                              idx_in[t + 1].view(-1)) if nxt else None,
                                  forced=None if f_post is None else f_post[t], flips=flips, idx=post_idx[t].view(-1))
        Cuts.mark("wm.mid")
        # prior head for all steps at once
        x2pre, x2 = g("obs.x2pre", (T, B, Hd)), g("obs.x2", (T, B, Hd))
        m2, r2 = g("obs.m2", (T, B)), g("obs.r2", (T, B))
        prior_logit, prior_stoch = g("obs.prior_logit", (T, B, S, D)), g("obs.prior_stoch", (T, B, S, D))
        dense_ln_fwd(P.img_out, v2(deter, De), None, v2(x2pre, Hd), m2.view(TB), r2.view(TB), v2(x2, Hd))
        ops.gemm(v2(x2, Hd), P.ims.W, v2(prior_logit, SD), bias=P.ims.b)
        ops.onehot_sample(prior_logit, prior_stoch, noise=q_prior, rng=rng, unimix=self.unimix, forced=f_prior,
                          flips=flips)
        self._embed = embed_tm
        return dict(post_stoch=post_stoch, post_logit=post_logit, deter=deter, prior_stoch=prior_stoch,
                    prior_logit=prior_logit, action=ain, post_idx=post_idx)

    def lanes_pay(self, heavy_side: bool) -> bool:
        """Whether the reverse scan should run beside the deferred weight gradients on the two CU-masked lanes.
        Measured on MI355X (ms per update, lanes / in line): cfg 2 16.22 / 16.65, cfg 3 26.50 / 27.24 -- but cfg 1 (vector
        decoder: ~0.2 ms of deferred work against two host waits and three more graph launches) 12.95 / 12.79, cfg 5
        (deter 2048: 100 MB of weights per scan step) 267.9 / 255.6, cfg 4 (deter 4096) 391.6 / 383.9: once a step of the
        scan streams more weights than half of the chip's L2 / Infinity-Cache paths deliver in its fixed launch cost,
        the chain is bandwidth-bound and wants every CU.  heavy_side: the deferred launches include the conv decoder's."""
        gru_weight_bytes = 4 * 3 * self.De * (self.Hd + self.De)
        return bool(heavy_side) and gru_weight_bytes <= (32 << 20) and self.B <= 64

    def pipeline_mode(self, conv: bool):
        """How graph.UpdateRunner.step_pipelined runs the behaviour phase of update k beside the world-model phase of
        update k+1 at this shape -- "lanes": each phase on one half of the chip from end to end; "staged": only the two
        scans beside the rollout / reverse rollout, everything else on the whole chip; None: one update after the other.
        Measured on MI355X (tools/pipe_bench.py, ms per update serial / staged / lanes): cfg 2 16.2 / 15.3 / 13.75,
        cfg 3 26.6 / 25.5 / 24.2 -- a segment of the update's chip-filling launches takes only 1.5-1.8x as long on 128
        compute units as on 256, so two half-chip streams of independent work beat one whole-chip stream; cfg 1 (MLP
        encoder / decoder: the world-model phase is 7 ms of the 12.8, the behaviour lane would be the long pole)
        12.8 / 11.4 / 12.7.  conv: the world model has the convolutional encoder / decoder."""
        gru_weight_bytes = 4 * 3 * self.De * (self.Hd + self.De)
        if gru_weight_bytes > (32 << 20) or getattr(self, "B", 0) > 64:
            return None  # (wide cells: the scans are bandwidth-bound and want every compute unit, see lanes_pay)
        return "lanes" if conv else "staged"

    def observe_bwd(self, dpost_logit, dprior_logit, gs, gd, dembed, extra_side=None, lanes_pay=False):
        """Backward of observe_fwd.

        dpost_logit/dprior_logit [T,B,S,D]: gradient on the logits (from the KL; dpost_logit is updated in
        place with the straight-through term).  gs [T,B,SD], gd [T,B,De]: gradient on post stoch / deter
        from the heads (both are used as scratch).  dembed [T,B,E] receives the encoder-output gradient.
        All RSSM parameter gradients are accumulated into their .grad views.  extra_side: callables (weight
        gradients of the heads / decoder) to run on the side stream beside the reverse scan; lanes_pay: what
        self.lanes_pay() said about them.  Returns the SideStream so that the caller can put more work beside the encoder
        backward and join()."""
        P, ws = self.P, self.ws
        T, B, S, D, SD, De, Hd, A, E = self.T, self.B, self.S, self.D, self.SD, self.De, self.Hd, self.A, self.E
        TB = T * B
        g = ws.get
        first = g("obs.first", (T, B))
        sin, din, ain = g("obs.sin", (T, B, SD)), g("obs.din", (T, B, De)), g("obs.ain", (T, B, A))
        x1pre, x1 = g("obs.x1pre", (T, B, Hd)), g("obs.x1", (T, B, Hd))
        m1, r1 = g("obs.m1", (T, B)), g("obs.r1", (T, B))
        gpre, mg, rg = g("obs.gpre", (T, B, 3 * De)), g("obs.mg", (T, B)), g("obs.rg", (T, B))
        deter = g("obs.deter", (T, B, De))
        x3pre, x3 = g("obs.x3pre", (T, B, Hd)), g("obs.x3", (T, B, Hd))
        m3, r3 = g("obs.m3", (T, B)), g("obs.r3", (T, B))
        post_logit = g("obs.post_logit", (T, B, S, D))
        x2pre, x2 = g("obs.x2pre", (T, B, Hd)), g("obs.x2", (T, B, Hd))
        m2, r2 = g("obs.m2", (T, B)), g("obs.r2", (T, B))
        # ---- prior head, batched: prior_logit -> ims -> LN/SiLU -> img_out -> deter
        dx2 = g("obs.dx2", (TB, Hd))
        dpl2 = v2(dprior_logit, SD)
        side = SideStream(dprior_logit.device, lanes_pay=lanes_pay)
        ops.gemm(dpl2, P.ims.W, dx2, transB=False)
        dx2pre = g("obs.dx2pre", (TB, Hd))
        dense_ln_bwd_pre(P.img_out, dx2, v2(x2pre, Hd), m2.view(TB), r2.view(TB), dx2pre, wgrad=True)
        ops.gemm(dx2pre, P.img_out.W, v2(gd, De), transB=False, accumulate=True)

        def _prior_wgrads():
            lin_wgrad(P.ims.W, dpl2, v2(x2, Hd))
            ops.colsum(dpl2, _g(P.ims.b), accumulate=True)
            lin_wgrad(P.img_out.W, dx2pre, v2(deter, De))

        # (pipelined capture: the prior head's weight gradients stay in line -- the deferred launches may then run beside
        # the segment behind the join, where the init-state backward adds into the same gradients)
        prior_inline = side._mode == "cuts"
        side.run((extra_side or []) + ([] if prior_inline else [_prior_wgrads]))  # beside the reverse scan below
        with side.chain():  # (lanes: on the scan lane, beside the deferred launches on the side lane)
            # ---- reverse scan
            # Every GEMM of the reverse scan ACCUMULATES into buffers zeroed here in bulk: a few-row GEMM with
            # accumulate="atomic" may split K over workgroups (atomics onto C), which is what fills the chip at B rows.  The two
            # data gradients of the GRU matmul share one GEMM: dxd[t] = [dx1 | ddin], whose right half gru_bwd
            # pre-loads with the direct dh path.
            dx3, dxd, dsin, dstoch0, ddeter0 = ws.zeros_many(
                "obs.zeroed", [(T, B, Hd), (T, B, Hd + De), (T, B, SD), (SD,), (De,)])
            dx3pre = g("obs.dx3pre", (T, B, Hd))
            dgpre = g("obs.dgpre", (T, B, 3 * De))
            dx1pre = g("obs.dx1pre", (T, B, Hd))
            fuse_carry = _FUSE_CARRY
            # row operations in the prologue of the few-row GEMM that consumes them (csrc/scanops.hip): 5 launches per step
            fuse_row = (_FUSE_SCAN_ROW and _FUSE_SCAN_LNBWD and B <= 64 and ops.scan_lnbwd_gemm_ok(Hd, De)
                        and ops.scan_lnbwd_gemm_ok(Hd, SD))
            fuse_cs = (_FUSE_SCAN_ROW and _FUSE_SCAN_CS and B <= 64 and fuse_carry and ops.scan_carry_st_gemm_ok(S, D, Hd))
            # (the fused first launch re-reads its inputs from every column tile: the finished logit gradient goes to its
            # own buffer instead of in place)
            dpl_out = g("obs.dpl_out", (T, B, S, D)) if fuse_cs else dpost_logit
            # (measured, world-model update: cfg 2 (De 512, 16 rows) 10.30 -> 10.15 ms; cfg 3 (De 1024, 32 rows) 17.19 -> 17.48:
            # 288 workgroups x 2 row blocks each re-reading 192 KB of factors -- the wide cell keeps its own launch)
            fuse_gru = (_FUSE_SCAN_ROW and _FUSE_SCAN_GRUBWD and B <= 32 and De <= 512
                        and ops.scan_grubwd_gemm_ok(De, Hd + De))
            if fuse_gru:
                # ... and the GRU cell's backward: every transcendental factor for all steps in one launch
                xhg, afg = g("obs.xhg", (T, B, 3 * De)), g("obs.afg", (T, B, 3 * De))
                p1g, p2g, ahg = g("obs.p1g", (T, B, De)), g("obs.p2g", (T, B, De)), g("obs.ahg", (T, B, De))
                ops.scan_gru_factors(v2(gpre, 3 * De), P.gru.g, P.gru.b, v2(din, De), mg.view(TB), rg.view(TB), xhg, afg, p1g,
                                     p2g, ahg)
            if fuse_row:
                # what the two LayerNorm + SiLU backward prologues need from the forward pass, for all steps at once
                xh3, jc3 = g("obs.xh3", (T, B, Hd)), g("obs.jc3", (T, B, Hd))
                xh1, jc1 = g("obs.xh1", (T, B, Hd)), g("obs.jc1", (T, B, Hd))
                ops.scan_ln_factors(v2(x3pre, Hd), P.obs_out.g, P.obs_out.b, m3.view(TB), r3.view(TB), xh3, jc3)
                ops.scan_ln_factors(v2(x1pre, Hd), P.img_in.g, P.img_in.b, m1.view(TB), r1.view(TB), xh1, jc1)
            for t in reversed(range(T)):
                if T >= 16 and t == T // 4 - 1:
                    # (captured update: the join behind the scan is queued once the GPU is here -- a main queue blocked at
                    # the join costs every launch of the chain ~1.3 us, see graph.SegmentRecorder.replay)
                    SideStream.late_join_point()
                gs_t, gd_t = gs[t], gd[t]  # already hold the carry from step t+1 (folded in by obs_blend_bwd)
                dx1, ddin = dxd[t][:, :Hd], dxd[t][:, Hd:]
                if fuse_cs:
                    # the carry out of step t+1, this step's straight-through gradient and dx3 += dlogit W_obs
                    carry = None if t == T - 1 else (dsin[t + 1], dxd[t + 1][:, Hd:], first[t + 1], gd_t, dstoch0, ddeter0)
                    ops.scan_carry_st_gemm(gs_t.view(B, SD), post_logit[t], dpost_logit[t], dpl_out[t], P.obs.W, dx3[t],
                                           unimix=self.unimix, carry=carry)
                else:
                    if t == T - 1 or not fuse_carry:  # (otherwise done by step t+1's fused carry + straight-through launch)
                        ops.onehot_st_bwd(post_logit[t], gs_t.view(B, S, D), dpost_logit[t], unimix=self.unimix,
                                          accumulate=True)
                    ops.gemm(dpost_logit[t].view(B, SD), P.obs.W, dx3[t], transB=False, accumulate="atomic")
                if fuse_row:
                    ops.scan_lnbwd_gemm(dx3[t], xh3[t], jc3[t], P.obs_out.g, r3[t], dx3pre[t], P.obs_out.W[:, :De], gd_t,
                                        _g(P.obs_out.g), _g(P.obs_out.b))
                else:
                    dense_ln_bwd_pre(P.obs_out, dx3[t], x3pre[t], m3[t], r3[t], dx3pre[t], wgrad=True)
                    ops.gemm(dx3pre[t], P.obs_out.W[:, :De], gd_t, transB=False, accumulate="atomic")
                if fuse_gru:
                    ops.scan_grubwd_gemm(gd_t, xhg[t], afg[t], p1g[t], p2g[t], ahg[t], P.gru.g, rg[t], dgpre[t], ddin, P.gru.W,
                                         dxd[t], _g(P.gru.g), _g(P.gru.b))
                else:
                    ops.gru_bwd(gd_t, gpre[t], P.gru.g, P.gru.b, din[t], mg[t], rg[t], dgpre[t], ddin, _g(P.gru.g),
                                _g(P.gru.b))
                    ops.gemm(dgpre[t], P.gru.W, dxd[t], transB=False, accumulate="atomic")
                if fuse_row:
                    ops.scan_lnbwd_gemm(dx1, xh1[t], jc1[t], P.img_in.g, r1[t], dx1pre[t], P.img_in.W[:, :SD], dsin[t],
                                        _g(P.img_in.g), _g(P.img_in.b))
                else:
                    dense_ln_bwd_pre(P.img_in, dx1, x1pre[t], m1[t], r1[t], dx1pre[t], wgrad=True)
                    ops.gemm(dx1pre[t], P.img_in.W[:, :SD], dsin[t], transB=False, accumulate="atomic")
                if fuse_cs:
                    if t == 0:
                        ops.obs_blend_bwd(dsin[0], ddin, first[0], None, None, dstoch0, ddeter0)
                elif fuse_carry and t > 0:
                    ops.obs_carry_st_bwd(dsin[t], ddin, first[t], gs[t - 1], gd[t - 1], dstoch0, ddeter0, post_logit[t - 1],
                                         dpost_logit[t - 1], unimix=self.unimix)
                else:
                    ops.obs_blend_bwd(dsin[t], ddin, first[t], gs[t - 1] if t > 0 else None,
                                      gd[t - 1] if t > 0 else None, dstoch0, ddeter0)
            # ---- the encoder-output gradient (critical path) and, beside it, the batched weight gradients
        side.join()  # the init-state backward below adds into the same prior-head gradients
        ops.gemm(v2(dx3pre, Hd), P.obs_out.W[:, De:], v2(dembed, E), transB=False)
        dpl = v2(dpl_out, SD)

        def _scan_wgrads():
            # (one grid for these products, ops.gemm_group in SideStream.run; the init-state backward's additions into the
            # prior head's gradients overlap the prior's own and go out as a second one behind it)
            if prior_inline:
                _prior_wgrads()
            lin_wgrad(P.obs.W, dpl, v2(x3, Hd))
            ops.colsum(dpl, _g(P.obs.b), accumulate=True)
            lin_wgrad(P.obs_out.W, v2(dx3pre, Hd), v2(deter, De), v2(self._embed, E))
            lin_wgrad(P.gru.W, v2(dgpre, 3 * De), v2(x1, Hd), v2(din, De))
            lin_wgrad(P.img_in.W, v2(dx1pre, Hd), v2(sin, SD), v2(ain, A))
            self.init_state_bwd(dstoch0, ddeter0)

        side.run([_scan_wgrads], chain=False)  # (plain second stream: beside the encoder backward the caller launches next)
        return side

    # -- one img_step on a row block (networks.py:208-233), used by the policy path and imagine ---------
    def pack_img_in(self, defer=None, name="rssm.img_in_wt"):
        """Transposed copy of the img_in weight, [Hd, SD+A] -> [SD+A, Hd], for the one-hot gather path of img_step
        (once per update: the world model's weights are frozen during imagination, models.py:335)."""
        W = self.P.img_in.W
        wt = self.ws.get(name, (W.shape[1], W.shape[0]))
        if defer is not None:
            defer.append((W, wt))
        else:
            ops.transpose2d(W, wt)
        return wt

    def img_step_fwd(self, stoch, deter, action, bufs, *, noise=None, rng=None, sample=True, forced=None,
                     flips=None, idx=None, idx_out=None, wcat=None, head=True):
        """stoch [M,SD], deter [M,De], action [M,A]; bufs: dict of per-step buffers (see imagine_fwd).
        idx (int32 [M,S]): the class indices of stoch (an exact one-hot): img_in then runs as gather + LayerNorm +
        SiLU in one launch (pack_img_in must have run) instead of GEMM + LN.  idx_out (int32 [M,S]) receives the
        class indices of the sampled successor.  wcat ([Hd + X, De] = img_out weight stacked on other Linears that
        read the new deter, e.g. the actor's first layer): ONE GEMM writes bufs["cat"] [M, Hd + X] whose first Hd
        columns are bufs["x2pre"] (a view of it) -- the consumers of deter' share its launch.
        head=False stops after the GRU (deter' only): the acting step never reads the prior (dreamer.py:131-134)."""
        P = self.P
        M = stoch.shape[0]
        if idx is not None:
            wt = self.ws.get("rssm.img_in_wt", (P.img_in.W.shape[1], P.img_in.W.shape[0]))
            ops.onehot_linear_ln(idx, self.D, wt, bufs["x1pre"], x2=action, gamma=P.img_in.g, beta=P.img_in.b,
                                 y=bufs["x1"], mean=bufs["m1"], rstd=bufs["r1"])
        else:
            dense_ln_fwd(P.img_in, stoch, action, bufs["x1pre"], bufs["m1"], bufs["r1"], bufs["x1"])
        ops.gemm(bufs["x1"], P.gru.W, bufs["gpre"], A2=deter)
        ops.gru_fwd(bufs["gpre"], P.gru.g, P.gru.b, deter, bufs["deter"], bufs["mg"], bufs["rg"])
        if not head:
            return
        fuse_smp = _FUSE_SAMPLE and ops.gemm_sample_ok(M, self.SD, self.D)
        if wcat is not None:
            ops.gemm(bufs["deter"], wcat, bufs["cat"])
        else:
            ops.gemm(bufs["deter"], P.img_out.W, bufs["x2pre"])
        ops.ln_act_fwd(bufs["x2pre"], P.img_out.g, P.img_out.b, bufs["x2"], bufs["m2"], bufs["r2"], act=True)
        io = None if idx_out is None else idx_out.view(-1)
        if fuse_smp:
            ops.gemm_sample(bufs["x2"], P.ims.W, bufs["logit"].view(M, self.SD), bufs["stoch"], bias=P.ims.b,
                            noise=noise, rng=rng, idx=io, forced=forced, flips=flips, unimix=self.unimix,
                            mode=not sample)
        else:
            ops.gemm(bufs["x2"], P.ims.W, bufs["logit"].view(M, self.SD), bias=P.ims.b)
            ops.onehot_sample(bufs["logit"], bufs["stoch"], noise=noise, rng=rng, unimix=self.unimix, mode=not sample,
                              forced=forced, flips=flips, idx=io)

    def pack_bwd(self, defer=None):
        """Transposed copies of the four img_step weights for its data gradients: dX = dY W is then the y = x B^T form
        with B = W^T [K_in, N_out] (k-contiguous rows: 16-byte operand loads in the register-direct kernel, and the
        6-column action gradient takes the narrow-output path).  Once per behaviour update: the world model's weights
        are frozen while it runs (models.py:335), the 14 steps of the reverse rollout share the copies."""
        P, ws = self.P, self.ws
        wt = {"in": self.pack_img_in(defer=defer)}
        for nm, W in (("gru", P.gru.W), ("out", P.img_out.W), ("ims", P.ims.W)):
            t = ws.get(f"rssm.{nm}_wt", (W.shape[1], W.shape[0]))
            if defer is not None:
                defer.append((W, t))
            else:
                ops.transpose2d(W, t)
            wt[nm] = t
        return wt

    def img_step_bwd(self, dstoch, ddeter, prev_deter, bufs, scratch, dprev_stoch, dprev_deter, daction,
                     accumulate_prev=False, wt=None):
        """Input gradients of img_step_fwd (world-model weights frozen: no wgrad; models.py:335).
        dstoch [M,SD] / ddeter [M,De]: total gradient on the step's outputs (ddeter is used as scratch).
        Writes (or, with accumulate_prev, adds into) dprev_stoch / dprev_deter; writes daction."""
        P = self.P
        M = dstoch.shape[0]
        S, D, SD, De, Hd = self.S, self.D, self.SD, self.De, self.Hd
        dlogit = scratch["dlogit"]
        ops.onehot_st_bwd(bufs["logit"], dstoch.view(M, S, D), dlogit.view(M, S, D), unimix=self.unimix)
        if wt is not None:  # transposed weights (pack_bwd): every data gradient in the y = x B^T form
            ops.gemm(dlogit, wt["ims"], scratch["dx2"])
            dense_ln_bwd_pre(P.img_out, scratch["dx2"], bufs["x2pre"], bufs["m2"], bufs["r2"], scratch["dx2pre"],
                             wgrad=False)
            ops.gemm(scratch["dx2pre"], wt["out"], ddeter, accumulate=True)
            ops.gru_bwd(ddeter, bufs["gpre"], P.gru.g, P.gru.b, prev_deter, bufs["mg"], bufs["rg"], scratch["dgpre"],
                        dprev_deter, accumulate_dh=accumulate_prev)
            if M > 128 and ops.gemm_split_ok(scratch["dgpre"], wt["gru"]):
                # [dx1 | dh] = dgpre W_gru in one launch, dh accumulating onto the direct path gru_bwd just wrote
                ops.gemm_split(scratch["dgpre"], wt["gru"], scratch["dx1"], dprev_deter, accumulate2=True)
            else:
                ops.gemm(scratch["dgpre"], wt["gru"][Hd:], dprev_deter, accumulate=True)
                ops.gemm(scratch["dgpre"], wt["gru"][:Hd], scratch["dx1"])
            dense_ln_bwd_pre(P.img_in, scratch["dx1"], bufs["x1pre"], bufs["m1"], bufs["r1"], scratch["dx1pre"],
                             wgrad=False)
            if daction is not None and M > 128 and SD % 16 == 0 and ops.gemm_split_ok(scratch["dx1pre"], wt["in"]):
                # [dstoch | daction] = dx1pre W_in in one launch (the action's few columns ride in the last column tile)
                ops.gemm_split(scratch["dx1pre"], wt["in"], dprev_stoch, daction, accumulate=accumulate_prev)
                return
            ops.gemm(scratch["dx1pre"], wt["in"][:SD], dprev_stoch, accumulate=accumulate_prev)
            if daction is not None:
                ops.gemm(scratch["dx1pre"], wt["in"][SD:], daction)
            return
        ops.gemm(dlogit, P.ims.W, scratch["dx2"], transB=False)
        dense_ln_bwd_pre(P.img_out, scratch["dx2"], bufs["x2pre"], bufs["m2"], bufs["r2"], scratch["dx2pre"],
                         wgrad=False)
        ops.gemm(scratch["dx2pre"], P.img_out.W, ddeter, transB=False, accumulate=True)
        ops.gru_bwd(ddeter, bufs["gpre"], P.gru.g, P.gru.b, prev_deter, bufs["mg"], bufs["rg"], scratch["dgpre"],
                    dprev_deter, accumulate_dh=accumulate_prev)
        ops.gemm(scratch["dgpre"], P.gru.W[:, Hd:], dprev_deter, transB=False, accumulate=True)
        ops.gemm(scratch["dgpre"], P.gru.W[:, :Hd], scratch["dx1"], transB=False)
        dense_ln_bwd_pre(P.img_in, scratch["dx1"], bufs["x1pre"], bufs["m1"], bufs["r1"], scratch["dx1pre"],
                         wgrad=False)
        ops.gemm(scratch["dx1pre"], P.img_in.W[:, :SD], dprev_stoch, transB=False, accumulate=accumulate_prev)
        if daction is not None:
            ops.gemm(scratch["dx1pre"], P.img_in.W[:, SD:], daction, transB=False)


# ---------------------------------------------------------------------------------------------
# 64x64 conv stacks (networks.ConvEncoder / ConvDecoder, networks.py:448-585), NHWC activations
# ---------------------------------------------------------------------------------------------
@dataclass
class PConvLayer:
    W: torch.Tensor  # Conv2d [Co,Ci,4,4] (encoder) / ConvTranspose2d [Ci,Co,4,4] (decoder)
    g: Optional[torch.Tensor] = None  # channel LayerNorm scale / shift (None on the decoder's last layer)
    b: Optional[torch.Tensor] = None
    bias: Optional[torch.Tensor] = None  # only the decoder's last layer has a bias


# acting path: conv layers with at most this many output pixels (all images) use im2col + GEMM
_IM2COL_ROWS = _dev.value("DV3_IM2COL_ROWS", 2048)


class ConvEncoderEngine:
    def __init__(self, layers: List[PConvLayer], ws: Workspace, size=64):
        self.L, self.ws, self.size = layers, ws, size

    def forward(self, image_u8=None, perm=None, x_f32=None, keep=True):
        """image_u8 [B,T,H,W,C] u8 (or x_f32 [N,H,W,C] already = image/255 - 0.5) -> embed [N, E] in the
        reference's (C,H,W) flatten order; rows time-major when perm=(B,T) (row t*B+b <- image b*T+t).
        keep=False (inference, no backward follows): layers with few output rows go through im2col + GEMM."""
        ws = self.ws
        H = self.size
        if x_f32 is not None:
            x, N = x_f32, x_f32.shape[0]
        else:
            N = image_u8.shape[0] * image_u8.shape[1]
            C = image_u8.shape[-1]
            x = ws.get("enc.x0", (N, H, H, C))
            ops.image_to_f32(image_u8, x, n_images=N, perm=perm)
        self._acts = []
        n_layers = len(self.L)
        for i, L in enumerate(self.L):
            Co, Ci = L.W.shape[0], L.W.shape[1]
            OH = H // 2
            pre = ws.get(f"enc.pre{i}", (N, OH, OH, Co))
            if N * OH * OH <= _IM2COL_ROWS and not keep:
                # few images (the acting step): explicit patches + a plain GEMM against the weight as stored
                cols = ws.get(f"enc.cols{i}", (N * OH * OH, 16 * Ci))
                ops.im2col_s2(x, cols)
                ops.gemm(cols, L.W.view(Co, 16 * Ci), pre.view(N * OH * OH, Co))
            elif Ci == 3 and Co in ops.C3_WIDTHS:
                ops.conv_s2_c3_fwd(x, L.W, pre, CW=Co)
            else:
                wp = ws.get(f"enc.wp{i}", (Co, 16 * Ci))
                ops.pack_conv_weight(L.W, wp, transposed=False)
                ops.conv_s2_fwd(x, wp, pre, Ci=Ci, Co=Co)
            R = N * OH * OH
            mean, rstd = ws.get(f"enc.m{i}", (R,)), ws.get(f"enc.r{i}", (R,))
            last = i == n_layers - 1
            y = ws.get(f"enc.y{i}", (N, Co * OH * OH) if last else (N, OH, OH, Co))
            ops.ln_act_fwd(pre.view(R, Co), L.g, L.b, y if last else y.view(R, Co), mean, rstd, act=True,
                           chw_group=OH * OH if last else 0)
            self._acts.append((x, pre, mean, rstd, y))
            x, H = y, OH
        return x

    def backward(self, dembed):
        ws = self.ws
        n_layers = len(self.L)
        dy = dembed
        for i in reversed(range(n_layers)):
            L = self.L[i]
            x, pre, mean, rstd, y = self._acts[i]
            Co, Ci = L.W.shape[0], L.W.shape[1]
            N, OH = pre.shape[0], pre.shape[1]
            R = N * OH * OH
            last = i == n_layers - 1
            dpre = ws.get(f"enc.dpre{i}", pre.shape)
            ops.ln_act_bwd(dy if last else dy.view(R, Co), pre.view(R, Co), L.g, L.b, mean, rstd, dpre.view(R, Co),
                           _g(L.g), _g(L.b), act=True, chw_group=OH * OH if last else 0)
            ops.conv_s2_wgrad(dpre, x, _g(L.W))
            if i > 0:
                wpt = ws.get(f"enc.wpt{i}", (4, Ci, 4 * Co))
                ops.pack_conv_weight(L.W, wpt, transposed=True)  # Conv2d weight read as its adjoint
                dx = ws.get(f"enc.dy{i - 1}", x.shape)
                ops.convT_s2_fwd(dpre, wpt, dx, Ci=Co, Co=Ci)
                dy = dx


class ConvDecoderEngine:
    def __init__(self, lin: PLin, layers: List[PConvLayer], ws: Workspace, minres=4):
        self.lin, self.L, self.ws, self.minres = lin, layers, ws, minres

    def forward(self, x1, x2):
        """feat = [x1|x2] rows -> mean image [R, 64, 64, 3] (+0.5 folded into the last layer)."""
        ws = self.ws
        R = x1.shape[0]
        E = self.lin.W.shape[0]
        mr = self.minres
        C0 = E // (mr * mr)
        h0 = ws.get("dec.h0", (R, mr, mr, C0))
        ops.gemm(x1, self.lin.W, h0.view(R, E), A2=x2, bias=self.lin.b)
        self._x = (x1, x2)
        self._acts = []
        x, H = h0, mr
        for i, L in enumerate(self.L):
            Ci, Co = L.W.shape[0], L.W.shape[1]
            OH = 2 * H
            c3 = L.g is None and Co == 3 and Ci in ops.C3_WIDTHS
            if not c3:
                wpt = ws.get(f"dec.wpt{i}", (4, Co, 4 * Ci))
                ops.pack_conv_weight(L.W, wpt, transposed=True)
            if L.g is not None:
                pre = ws.get(f"dec.pre{i}", (R, OH, OH, Co))
                ops.convT_s2_fwd(x, wpt, pre, Ci=Ci, Co=Co)
                rows = R * OH * OH
                mean, rstd = ws.get(f"dec.m{i}", (rows,)), ws.get(f"dec.r{i}", (rows,))
                y = ws.get(f"dec.y{i}", (R, OH, OH, Co))
                ops.ln_act_fwd(pre.view(rows, Co), L.g, L.b, y.view(rows, Co), mean, rstd, act=True)
                self._acts.append((x, pre, mean, rstd, y))
            else:
                y = ws.get("dec.recon", (R, OH, OH, Co))
                if c3:
                    ops.convT_s2_c3_fwd(x, L.W, y, CW=Ci, bias=L.bias, out_add=0.5)
                else:
                    ops.convT_s2_fwd(x, wpt, y, Ci=Ci, Co=Co, bias=L.bias, out_add=0.5)
                self._acts.append((x, None, None, None, y))
            x, H = y, OH
        return x

    def backward(self, drecon, dx1, dx2, *, acc_dx=False, defer=None):
        ws = self.ws
        run = defer.append if defer is not None else (lambda f: f())
        dy = drecon
        for i in reversed(range(len(self.L))):
            L = self.L[i]
            x, pre, mean, rstd, y = self._acts[i]
            Ci, Co = L.W.shape[0], L.W.shape[1]
            R, OH = y.shape[0], y.shape[1]
            rows = R * OH * OH
            if L.g is not None:
                dpre = ws.get(f"dec.dpre{i}", pre.shape)
                ops.ln_act_bwd(dy.view(rows, Co), pre.view(rows, Co), L.g, L.b, mean, rstd, dpre.view(rows, Co),
                               _g(L.g), _g(L.b), act=True)
            else:
                dpre = dy
                if L.bias is not None:
                    run(lambda dpre=dpre, L=L, rows=rows, Co=Co: ops.colsum(dpre.view(rows, Co), _g(L.bias),
                                                                         accumulate=True))
            # coarse = layer input, fine = output gradient
            (run if _DEFER_CONV_WGRAD else (lambda f: f()))(lambda x=x, dpre=dpre, L=L: ops.conv_s2_wgrad(x, dpre, _g(L.W)))
            dx = ws.get(f"dec.dx{i}", x.shape)
            if Co == 3 and Ci in ops.C3_WIDTHS:
                ops.conv_s2_c3_fwd(dpre, L.W, dx, CW=Ci)  # ConvTranspose2d weight [Ci,3,4,4] read as its adjoint
            else:
                wp = ws.get(f"dec.wp{i}", (Ci, 16 * Co))
                ops.pack_conv_weight(L.W, wp, transposed=False)  # ConvTranspose2d weight read as its adjoint
                ops.conv_s2_fwd(dpre, wp, dx, Ci=Co, Co=Ci)
            dy = dx
        R = dy.shape[0]
        E = self.lin.W.shape[0]
        dh0 = dy.view(R, E)
        x1, x2 = self._x

        def _lin_wg():
            lin_wgrad(self.lin.W, dh0, x1, x2)
            ops.colsum(dh0, _g(self.lin.b), accumulate=True)

        run(_lin_wg)
        k1 = x1.shape[1]
        ops.gemm(dh0, self.lin.W[:, :k1], dx1, transB=False, accumulate=acc_dx)
        ops.gemm(dh0, self.lin.W[:, k1:], dx2, transB=False, accumulate=acc_dx)
