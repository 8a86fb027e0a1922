"""What ``model(x)`` and ``block(x)`` dispatch to by themselves — the drop-in tiers: ``AutoFusion`` (whole model:
the fused executor, as eager launches / stem launch + HIP graph / two halves in flight), ``BlockFusion`` (one residual
block), and ``install_auto_fusion`` (the same dispatch for ResNets of other packages, from ``prepare_binary_model``).
Reference call being served: ``outputs = net(inputs)`` (examples/cifar10.py:71,140-149)."""
from __future__ import annotations

import collections
import contextlib
import itertools
import os
import threading
import warnings
import weakref
from typing import Optional

import torch
import torch.nn as nn

from . import fastpath, native
from . import tails as _tails
from .executor import FusedBlocks, FusedResNet, FusionError, is_native_model, resnet_shaped, tap_binary_inputs
from .pipeline import TwoHalves
from .models.resnet import ResNet


_NO_MODEL_FUSION = 0


@contextlib.contextmanager
def no_model_fusion():
    """While active, whole-model fusion (``AutoFusion``) is off but residual blocks still fuse themselves
    (``BlockFusion``): what a network that is NOT laid out like the reference's ResNet gets (``bench.py --engine
    blockwise``)."""
    global _NO_MODEL_FUSION
    _NO_MODEL_FUSION += 1
    try:
        yield
    finally:
        _NO_MODEL_FUSION -= 1


class BlockFusion:
    """The second tier of the drop-in dispatch: a residual block of ``bnn_amd.models`` (``BasicBlock``, ``Bottleneck``,
    ``PreBasicBlock``, ``HBlock``) called on its own — inside a network that is not laid out like the reference's
    ``ResNet`` (a CIFAR-style three-stage ResNet-20, a custom backbone), or behind a stem the whole-model executor does
    not cover — evaluates itself as ``FusedBlocks([block])``: fp32 NCHW in -> ``pack_act`` -> the block's convolutions
    with BatchNorm / activation / residual add in their epilogues (activations between them as bit planes) -> fp32 NCHW
    out; 3 launches and 3 fp32 passes over HBM for a ``BasicBlock`` instead of 8 kernels and 13 passes.  Same conditions
    as ``AutoFusion`` (eval, no autograd, fp32 on a HIP device, no hooks on inner modules, not a replica); one instance
    per block in ``block.__dict__['_bnn_auto_block']``."""

    def __init__(self) -> None:
        self.engine: Optional["FusedBlocks"] = None
        self.failed_sig = None
        self.calls = {"fused": 0, "declined": 0}
        self.lock = threading.Lock()

    def __deepcopy__(self, memo):
        return BlockFusion()

    def __reduce__(self):
        return (BlockFusion, ())

    def run(self, block: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
        if (block.training or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4
                or x.shape[0] == 0 or getattr(block, "_is_replica", False) or _tails._PER_LAYER
                or os.environ.get("BNN_AMD_AUTOFUSE", "1") == "0" or fastpath.strict_weights() or not native.available()):
            self.calls["declined"] += 1
            return None
        with self.lock:
            eng = self.engine
            # (inside a caller's own graph capture nothing may be built or re-derived: an executor that is ready runs —
            # its launches are plain kernels on the capturing stream — anything else falls to the per-layer path)
            if torch.cuda.is_current_stream_capturing() and (eng is None or not eng._unchanged()):
                self.calls["declined"] += 1
                return None
            if eng is None:
                sig = _param_signature(block)
                if self.failed_sig == sig:
                    self.calls["declined"] += 1
                    return None
                try:
                    seq = nn.Sequential(block)
                    seq.training = False            # (a new container starts in training mode; the block is in eval mode)
                    eng = self.engine = FusedBlocks(seq)
                except FusionError:
                    self.failed_sig = sig
                    self.calls["declined"] += 1
                    return None
            if AutoFusion._hooked(block) or next(block.parameters()).device != x.device:
                self.calls["declined"] += 1
                return None
        try:
            y = eng(x)
        except FusionError:
            with self.lock:
                self.engine, self.failed_sig = None, _param_signature(block)
            self.calls["declined"] += 1
            return None
        self.calls["fused"] += 1
        return y


def auto_block_forward(block: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """Called at the top of the residual blocks' ``forward``: the fused block's output, or None -> its own forward."""
    st = block.__dict__.get("_bnn_auto_block")
    if st is None:
        st = block.__dict__["_bnn_auto_block"] = BlockFusion()
    return st.run(block, x)


class AutoFusion:
    """What makes the reference's own call fast: ``net = prepare_binary_model(...)``, ``net.eval()``, ``net(x)`` under
    ``torch.no_grad()`` (examples/cifar10.py:71,140-149) runs the fused executor instead of one launch per layer plus
    torch BatchNorm / ReLU / add round trips through HBM.

    One instance lives in ``model.__dict__['_bnn_auto']`` (not a sub-module: ``state_dict`` and ``repr`` are those of
    the reference).  ``run(model, x)`` returns the logits, or ``None`` when the call has to take the model's own
    per-layer forward: training mode or autograd recording, CPU / non-fp32 input, forward hooks registered on inner
    modules (they would not fire), a
    model the executor does not cover (``FusionError``, remembered until the parameters change), or
    ``BNN_AMD_AUTOFUSE=0``.

    Policy: the first batch of a given shape runs the fused launches eagerly (19 for ResNet-18); from the second one
    on the stem reads the caller's tensor and a HIP graph replays the rest (``FusedResNet.forward_fresh``) — the last,
    ragged batch of an epoch never pays for a capture.  A model built from classes of another package (same names
    and layout: the reference's ``bnn.models``) is fused only after its first fused result has been checked against
    its own forward on the same input (logits within ``VERIFY_TOL`` relative to the largest one).

    ``nn.DataParallel`` (examples/cifar10.py:74-77) replicates the model on every forward; the replicas share this object
    (``replicate`` copies ``__dict__``) and get ONE executor per device, derived from the first replica seen there
    and valid until a parameter of the master changes — so the reference's multi-GPU script runs the fused executor on
    every GPU, not the per-layer path."""

    # First-call check of a foreign model (``_verify``): images whose sign() results in front of EVERY binary convolution
    # equal those of the model's own forward went through the same integers — only fp32 rounding of the real-valued
    # layers is left, and they must agree to VERIFY_TOL (measured: 1e-6 of the largest logit).  An image where a rounding
    # decided one sign() differently (about 2 % of ImageNet-sized images) may move a logit by a few
    # per cent: VERIFY_TOL_FLIPPED, and at most half of the checked images may be such.  A wrong graph (a missing or extra
    # operation) changes the discrete state of every image behind it and is refused.
    VERIFY_TOL = 1e-4
    VERIFY_TOL_FLIPPED = 2e-2
    VERIFY_IMAGES = 4
    CAPTURE_AFTER = 1       # eager calls of a shape before its graph is captured

    def __init__(self, owner: Optional[nn.Module] = None) -> None:
        self.engine: Optional[FusedResNet] = None
        self.failed_sig = None          # parameter signature for which fusion was refused
        self.reason: Optional[str] = None
        self.verified = False
        self.seen = collections.Counter()
        self.lock = threading.RLock()   # re-entrant: the first-call check runs the model's own forward under it
        self._verifying = False
        self.calls = {"graph": 0, "eager": 0, "declined": 0}
        # the model this state belongs to.  nn.DataParallel replicas (``replicate`` copies ``__dict__`` shallowly) share
        # the object with the model they were made from: their executors live here, one per device
        self.owner = None if owner is None else weakref.ref(owner)
        self.replica_engines = {}       # device -> (master parameter signature, FusedResNet of the first replica there)
        self.halves = {}                # id(engine) -> TwoHalves of the same model (large batches: two halves in flight)

    def __deepcopy__(self, memo):       # copy.deepcopy(model): the copy derives its own executor
        return AutoFusion()

    def __reduce__(self):               # pickling / torch.save(model): derived data is not saved
        return (AutoFusion, ())

    def reset(self) -> None:
        with self.lock:
            # replays of the executors' HIP graphs may still be in flight on their side streams (two halves of a batch,
            # the shortcut stream): wait for them before the graphs and their static buffers are released
            for e in [self.engine] + [v[1] for v in self.replica_engines.values()]:
                p = next(e.parameters(), None) if isinstance(e, nn.Module) else None
                if p is not None and p.is_cuda:
                    torch.cuda.synchronize(p.device)
            self.engine, self.failed_sig, self.reason, self.verified = None, None, None, False
            self.seen.clear()
            self.replica_engines.clear()
            self.halves.clear()

    @staticmethod
    def enabled() -> bool:
        return (_tails._PER_LAYER == 0 and _NO_MODEL_FUSION == 0 and os.environ.get("BNN_AMD_AUTOFUSE", "1") != "0"
                and not fastpath.strict_weights())       # (strict: nothing derived from the weights outlives a call)

    @staticmethod
    def _hooked(model: nn.Module) -> bool:
        import torch.nn.modules.module as _mm
        if _mm._global_forward_hooks or _mm._global_forward_pre_hooks:
            return True
        return any(m._forward_hooks or m._forward_pre_hooks for m in model.modules() if m is not model)

    def _decline(self):
        self.calls["declined"] += 1
        return None

    def _engine_for(self, model: nn.Module, x: torch.Tensor) -> Optional["FusedResNet"]:
        """The executor for this call (built on first use), or None.  Called with the lock held."""
        if getattr(model, "_is_replica", False):
            # a DataParallel replica (examples/cifar10.py:74-77): its parameters are broadcast copies that are new on
            # every forward, but their VALUES are the master's — one executor per device, derived from the first
            # replica seen there (which it keeps alive), valid until a master parameter changes
            master = self.owner() if self.owner is not None else None
            if master is None or master is model:
                return None
            sig = _param_signature(master)
            if self.failed_sig == sig:
                return None
            ent = self.replica_engines.get(x.device)
            if ent is not None and ent[0] == sig:
                return ent[1]
            try:
                eng = FusedResNet(model)
            except FusionError as exc:
                self.failed_sig, self.reason = sig, str(exc)
                return None
            self.replica_engines[x.device] = (sig, eng)
            self.verified = self.verified or is_native_model(master)
            if not self.verified:
                return None     # a foreign class is verified on the master first (one un-replicated call)
            return eng
        eng = self.engine
        if eng is None:
            sig = _param_signature(model)
            if self.failed_sig == sig:
                return None
            try:
                eng = FusedResNet(model)
            except FusionError as exc:
                self.failed_sig, self.reason = sig, str(exc)
                return None
            self.engine = eng
            self.verified = self.verified or is_native_model(model)
            if self.owner is None:
                self.owner = weakref.ref(model)
        return eng

    def run(self, model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
        if (model.training or torch.is_grad_enabled() or not isinstance(x, torch.Tensor) or not x.is_cuda
                or x.dtype != torch.float32 or x.dim() != 4 or x.shape[0] == 0
                or not self.enabled() or not native.available()):
            return self._decline()
        try:
            with self.lock:     # lookup / build / verification / graph capture; the steady-state launches run outside
                if self._verifying:         # the model's own forward, run by _verify: not a call to dispatch
                    return self._decline()
                eng = self._engine_for(model, x)
                if eng is None or eng.model.fc.weight.device != x.device or (
                        eng.hooked() if eng.model is model else self._hooked(model)):
                    return self._decline()
                if torch.cuda.is_current_stream_capturing():
                    # the caller is capturing a HIP graph of its own around `net(x)`: no graph replay inside a capture, no
                    # second stream, nothing that synchronises — the executor's eager launches are plain kernels on the
                    # capturing stream (an executor that is not built and checked yet cannot be built here: per layer)
                    if not self.verified or eng._sig is None or not eng._unchanged():
                        return self._decline()
                    self.calls["eager"] += 1
                    return eng._forward_impl(x)
                if not self.verified:
                    return self._verify(eng, model, x)
                key = (tuple(x.shape), torch.cuda.current_stream(x.device).cuda_stream)
                graph = eng.reads_caller_tensor and (key in eng._split or self.seen[(id(eng),) + key] >= self.CAPTURE_AFTER)
                if graph and TwoHalves.wanted(x):
                    eng._check_current()                     # (a parameter change drops the half-batch executors too)
                    two = self.halves.get(id(eng))
                    if two is None or two.engines[0]._sig != eng._sig:
                        two = self.halves[id(eng)] = TwoHalves(eng.model, x.device)
                    self.calls["graph"] += 1
                    if not two.captured(x):
                        return two(x)                        # captures: under the lock
                    graph = two
                elif graph:
                    self.calls["graph"] += 1
                    if key not in eng._split:
                        return eng.forward_fresh(x)          # captures: under the lock
                else:
                    self.seen[(id(eng),) + key] += 1
                    if len(self.seen) > 64:
                        self.seen.clear()
                    self.calls["eager"] += 1
            if isinstance(graph, TwoHalves):
                return graph(x)
            return eng.forward_fresh(x) if graph else eng(x)
        except FusionError as exc:      # e.g. parameters moved to the CPU since the executor was built
            with self.lock:
                self.engine, self.failed_sig, self.reason = None, _param_signature(model), str(exc)
                self.replica_engines.clear()
                self.halves.clear()
            return self._decline()

    def _verify(self, eng: "FusedResNet", model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
        """First fused call of a model built from another package's classes: check it against the class's own forward —
        logits AND the discrete state (sign checksums in front of every binary convolution)."""
        n = min(self.VERIFY_IMAGES, x.shape[0])
        fused_state, own_state = {}, {}
        with tap_binary_inputs(lambda name, act: fused_state.__setitem__(
                name, _sign_digest_planes(act.P[:n], act.M[:n], act.shape[1]))):
            y = eng(x)
        hooks = [mod.register_forward_pre_hook(
                     lambda m, inp, name=name: own_state.__setitem__(name, _sign_digest_tensor(inp[0])))
                 for name, mod in model.named_modules() if name in fused_state]
        self._verifying = True
        try:
            want = getattr(type(model), "_bnn_base", type(model)).forward(model, x[:n])   # the class's OWN forward
        finally:
            self._verifying = False
            for h in hooks:
                h.remove()
        scale = want.abs().max().clamp_min(1e-30)
        err = ((y[:n] - want).abs().amax(1) / scale).tolist()
        same = [all(name in own_state and int(own_state[name][i]) == int(fused_state[name][i]) for name in fused_state)
                for i in range(n)]
        ok = all(e <= (self.VERIFY_TOL if eq else self.VERIFY_TOL_FLIPPED) for e, eq in zip(err, same)) and \
            2 * sum(same) >= n and set(own_state) == set(fused_state)
        if not ok:
            self.engine, self.failed_sig = None, _param_signature(model)
            self.reason = ("fused result differs from the model's own forward (relative logit error per image "
                           f"{[float('%.3g' % e) for e in err]}, same sign() results everywhere: {same})")
            warnings.warn(f"bnn_amd: {type(model).__name__}: {self.reason}; keeping the per-layer path",
                          RuntimeWarning)
            return self._decline()
        self.verified = True
        self.calls["eager"] += 1
        self.seen[(id(eng), tuple(x.shape), torch.cuda.current_stream(x.device).cuda_stream)] += 1
        return y


def _digest_weights(n: int, device) -> torch.Tensor:
    i = torch.arange(n, dtype=torch.int64, device=device)
    return ((i * 2654435761 + 12345) & 0xFFFFFFFF) >> 8


def _sign_digest_tensor(x: torch.Tensor) -> torch.Tensor:
    """Position-weighted checksum of ``sign(x)`` per image (int64 ``[N]``): equal digests <=> the same sign() results."""
    flat = x.detach().reshape(x.shape[0], -1)
    w = _digest_weights(flat.shape[1], x.device)
    return ((flat > 0).to(torch.int64) * w).sum(1) - ((flat < 0).to(torch.int64) * w).sum(1)


def _sign_digest_planes(P: torch.Tensor, M: torch.Tensor, C: int) -> torch.Tensor:
    """The same checksum from bit planes ``[N, ceil(C/64), H, W]`` (include/bnn_hip.h): channel ``64 g + b`` is bit ``b``."""
    n, g, h, wd = P.shape
    w = _digest_weights(g * 64 * h * wd, P.device).view(1, g, 64, h, wd)
    shifts = torch.arange(64, device=P.device, dtype=torch.int64).view(1, 1, 64, 1, 1)
    out = torch.zeros(n, dtype=torch.int64, device=P.device)
    for plane, sgn in ((P, 1), (M, -1)):      # (pad channels >= C are 0 in both planes: their weights never count)
        out += sgn * (((plane.unsqueeze(2) >> shifts) & 1) * w).sum((1, 2, 3, 4))
    return out


def _param_signature(model: nn.Module):
    return tuple((id(t), t.data_ptr(), t._version) for t in itertools.chain(model.parameters(), model.buffers()))


def auto_fusion(model: nn.Module) -> AutoFusion:
    """The model's ``AutoFusion`` state (created on first use)."""
    st = model.__dict__.get("_bnn_auto")
    if st is None:
        st = model.__dict__["_bnn_auto"] = AutoFusion(model)
    return st


def auto_forward(model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """Called at the top of ``bnn_amd.models.ResNet.forward``: fused logits, or None -> the caller's own forward."""
    return auto_fusion(model).run(model, x)


_AUTO_CLASSES: dict = {}


def _auto_class(base: type) -> type:
    """``base`` with the dispatch of ``bnn_amd.models.ResNet.forward`` in front of its own ``forward``: a subclass made
    on the fly (one per base class), the way ``torch.nn.utils.parametrize`` injects behaviour into a module instance.
    A class — not an instance attribute — so that ``DataParallel`` replicas (``replicate`` copies ``__dict__``), deep
    copies and pickles of the model each dispatch on THEMSELVES; name, module and repr stay those of ``base``."""
    dyn = _AUTO_CLASSES.get(base)
    if dyn is None:
        def forward(self, x, *args, **kwargs):
            if not args and not kwargs and isinstance(x, torch.Tensor) and x.is_cuda and not self.training \
                    and not torch.is_grad_enabled():
                y = auto_forward(self, x)
                if y is not None:
                    return y
            return base.forward(self, x, *args, **kwargs)

        def train(self, mode=True):             # train() <-> eval() drops the executor (models/resnet.py: ResNet.train)
            if bool(mode) != self.training:
                st = self.__dict__.get("_bnn_auto")
                if st is not None:
                    st.reset()
            return base.train(self, mode)

        def __reduce_ex__(self, protocol):      # pickle / deepcopy: rebuilt from the importable base class
            return (_rebuild_auto, (base,), self.__dict__)

        def _replicate_for_data_parallel(self):  # replicas share the master's AutoFusion (one executor per device)
            auto_fusion(self)
            return base._replicate_for_data_parallel(self)

        dyn = type(base.__name__, (base,), {"forward": forward, "train": train, "__reduce_ex__": __reduce_ex__,
                                            "_bnn_base": base,
                                            "_replicate_for_data_parallel": _replicate_for_data_parallel,
                                            "__module__": base.__module__, "__qualname__": base.__qualname__,
                                            "__doc__": base.__doc__})
        _AUTO_CLASSES[base] = dyn
    return dyn


def _rebuild_auto(base: type):
    cls = _auto_class(base)
    return cls.__new__(cls)


def install_auto_fusion(model: nn.Module) -> bool:
    """Give a ResNet of ANOTHER package (laid out like the reference's ``bnn.models.resnet.ResNet``) the same
    dispatch ``bnn_amd.models.ResNet.forward`` has: "fused executor when it applies, else the class's own forward".
    ``prepare_binary_model`` calls this for the model it converted; returns whether the model was recognised (and was
    not dispatching already).  Undo with ``uninstall_auto_fusion``."""
    if isinstance(model, ResNet) or hasattr(type(model), "_bnn_base") or not resnet_shaped(model):
        return False
    model.__class__ = _auto_class(type(model))
    return True


def uninstall_auto_fusion(model: nn.Module) -> None:
    base = getattr(type(model), "_bnn_base", None)
    if base is not None:
        model.__class__ = base
    model.__dict__.pop("_bnn_auto", None)


def optimize_for_inference(model: nn.Module) -> nn.Module:
    """Return the fused executor for ``model`` when it is covered, else ``model`` unchanged."""
    try:
        return FusedResNet(model)
    except FusionError:
        return model
