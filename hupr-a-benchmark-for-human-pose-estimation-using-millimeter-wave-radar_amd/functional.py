"""autograd.Function wrappers: one forward/backward pair of C-ABI calls per operator, plus a few fused nodes where a
chain of reference operators is cheaper as one (MSCSALevelFn, TemporalMergeFn, DualConvFn, the two-branch BatchNorm tail).

Activations are channels-last 5-D tensors ``(B, D, H, W, C)`` (2-D maps use D == 1), contiguous, fp32 — or, with
bf16 math and ``ACT_BF16``, bf16-stored inside the encoders / decoder stacks (arithmetic and parameters stay fp32).
Parameters keep the reference's shapes; the packed layouts the convolution kernels read are cached per parameter and
refreshed by ONE table-driven launch per optimiser step (``_packed`` / ``invalidate_packed``).  Parameter gradients
are written straight into the flat all-reduce buckets when a gradient sink is installed (``GRAD_SINK``).

Nothing here computes with torch ops except allocation, views, concatenation of small weight matrices and gradient
bookkeeping; every kernel is reached through ``runtime.lib()`` and fails loudly without the HIP library.
"""
import ctypes
import os

import numpy as np
import torch

from . import runtime as rt

_ws_cache = {}
CONV_PROBE = None      # bench.py installs a callable(x, co, k) -> (start_event, end_event) | None
# Precision state (round 5: per THREAD, ADVICE r3 item 5 / VERDICT r4 weak 2).  ``_st.math`` = matrix-pipe arithmetic of the GEMM-shaped
# ops: "f32" (parity path) or "bf16"; ``_st.act_f32_here`` / ``_st.region_switched`` = inside a region switched to "f32act" / to the fp32
# pipe.  Rounds 1-4 kept the three as module globals mutated from whichever thread ran a forward or — through the autograd engine's
# device thread — a backward: two models driven from two threads of one process (or one per device) silently shared one pipe
# selection.  Now every thread owns its copy: ``set_math`` sets the calling thread's mode (and the default a thread that never chose
# one starts from), ``math_mode`` scopes it, ``HuPRNet.math_mode`` pins it per model, and every autograd node carries the state of
# its forward into its backward on whatever thread the engine runs it (``_math_scoped``).  ``F_.MATH`` etc. stay readable
# (module ``__getattr__``) and name the calling thread's values.
import threading

_DEFAULT_MATH = ["f32"]


class _State(threading.local):
    def __init__(self):            # runs once per thread, on its first access
        self.math = _DEFAULT_MATH[0]
        self.act_f32_here = False
        self.region_switched = False


_st = _State()


def __getattr__(name):             # PEP 562: F_.MATH / F_._ACT_F32_HERE / F_._REGION_SWITCHED of the calling thread
    if name == "MATH":
        return _st.math
    if name == "_ACT_F32_HERE":
        return _st.act_f32_here
    if name == "_REGION_SWITCHED":
        return _st.region_switched
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
ACT_BF16 = True        # bf16 mode only: the 3-D encoders keep their activations in HBM as bf16 ("bf16act" kernels)
ACT_BF16_DECODER = True    # ... and so do the BasicBlock2D decoder stacks


def set_math(mode):
    """Select v_mfma_f32_32x32x2_f32 ("f32", exact) or the bf16 matrix pipe ("bf16", fp32 accumulate) for the CALLING thread, and as
    the mode threads that have not chosen one start from."""
    if mode not in ("f32", "bf16"):
        raise ValueError("math mode must be 'f32' or 'bf16'")
    _st.math = mode
    _DEFAULT_MATH[0] = mode


class math_mode:
    """``with math_mode("bf16"):`` — the calling thread's matrix-pipe mode for the block (nothing process-wide changes)."""

    def __init__(self, mode):
        if mode not in ("f32", "bf16"):
            raise ValueError("math mode must be 'f32' or 'bf16'")
        self.mode = mode

    def __enter__(self):
        self.prev = (_st.math, _st.act_f32_here, _st.region_switched)
        _st.math, _st.act_f32_here, _st.region_switched = self.mode, False, False
        return self

    def __exit__(self, *exc):
        _st.math, _st.act_f32_here, _st.region_switched = self.prev
        return False


# Per-region precision inside a bf16 run: PRECISION[region] = "f32" runs that region of HuPRNet.forward on the fp32 matrix
# pipe with fp32-stored activations, "f32act" keeps the bf16 matrix pipe but leaves the region's activations fp32 in HBM
# (models/layers.py wraps its regions in ``region(name)``); casts happen at the region borders (``to_act``).  Every autograd
# node remembers the mode of its forward and restores it for its backward, so the switches hold for training too.
# scripts/precision_regions.py uses them to localise where the bf16 path loses arg-max agreement on a trained network
# (profiles/r03_precision_regions.txt): everything but the last decoder block and the 1x1 head is >= 99.8 % on its own, those
# two alone cost 2 % / 1.3 % of the first head's joints.  DEFAULT therefore: the last BasicBlock2D keeps fp32 activations and the
# 14-channel head runs on the fp32 pipe (first head 99.1 -> 99.8-100 % identical arg-max, decoded head's max-abs error 3.8e-2 ->
# 1.1e-2, for +0.4 ms / step).  HUPR_F32_REGIONS / HUPR_F32ACT_REGIONS (comma lists) replace the default; "none" clears it.
REGIONS = ("mnet", "enc1", "enc2", "enc3", "merge", "lvl0", "lvl1", "lvl2", "dec3", "dec2", "dec1a", "dec1b", "head")
if "HUPR_F32_REGIONS" in os.environ or "HUPR_F32ACT_REGIONS" in os.environ:
    PRECISION = {r: "f32" for r in os.environ.get("HUPR_F32_REGIONS", "").split(",") if r and r != "none"}
    PRECISION.update({r: "f32act" for r in os.environ.get("HUPR_F32ACT_REGIONS", "").split(",") if r and r != "none"})
else:
    PRECISION = {"dec1b": "f32act", "head": "f32"}
assert all(r in REGIONS for r in PRECISION), PRECISION


class region:
    """``with region("dec1"):`` — the ops inside follow PRECISION["dec1"] when the run is a bf16 run."""

    def __init__(self, name):
        assert name in REGIONS, name
        self.name = name

    def __enter__(self):
        self.prev = (_st.math, _st.act_f32_here, _st.region_switched)
        mode = PRECISION.get(self.name)
        if _st.math == "bf16" and mode == "f32":
            _st.math = "f32"
            _st.region_switched = True
        _st.act_f32_here = _st.math == "bf16" and mode == "f32act"
        return self

    def __exit__(self, *exc):
        _st.math, _st.act_f32_here, _st.region_switched = self.prev
        return False


def _math_scoped(cls):
    """Class decorator for the autograd nodes: the backward pass runs under the matrix-pipe mode of its forward."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *a):
        ctx._hupr_math = (_st.math, _st.act_f32_here, _st.region_switched)
        return fwd(ctx, *a)

    def backward(ctx, *g):
        # (the engine runs this on ITS thread: that thread's state is set from the node and restored afterwards)
        prev = (_st.math, _st.act_f32_here, _st.region_switched)
        _st.math, _st.act_f32_here, _st.region_switched = ctx._hupr_math
        try:
            return bwd(ctx, *g)
        finally:
            _st.math, _st.act_f32_here, _st.region_switched = prev
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _fn(stem):
    return getattr(rt.lib(), "hupr_%s_%s" % (stem, _st.math))


def act_bf16():
    """True when the encoder island stores activations as bf16 (bf16 math + ACT_BF16)."""
    return _st.math == "bf16" and ACT_BF16 and not _st.act_f32_here


def _act(stem, t):
    """C entry point of an activation-dtype-generic operator for tensor ``t`` (fp32 or bf16 storage)."""
    return getattr(rt.lib(), "hupr_%s_%s" % (stem, "bf16act" if t.dtype == torch.bfloat16 else "f32"))


def workspace(nbytes, device):
    """Stream-ordered scratch shared by all ops of one (device, stream): the two encoder branches may run on two streams."""
    key = (device.type, device.index, rt.stream() if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- two compute streams ------------------------------------------------------------------------------------------
# The horizontal and the vertical branch of HuPRNet (chirp net + 3-D encoder each) are independent until the decoder.
# On one stream the GPU idles through every kernel tail and every few-microsecond kernel (two whole training processes
# sharing one GPU reach 1.15x the throughput of one); with the vertical branch on a side stream — forward and, because
# autograd runs a node's backward on the stream of its forward, backward too — the two branches fill each other's gaps.
TWO_STREAMS = os.environ.get("HUPR_ONE_STREAM", "0") != "1"
_side_streams = {}


def side_stream(device):
    """The per-device side compute stream (created on first use)."""
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device=device)
    return s


def side_streams_in_use(device):
    """Side streams that have been handed out for ``device`` (the all-reduce launcher waits for them as well)."""
    s = _side_streams.get(device.index)
    return [s] if s is not None else []


def two_streams_ok(t):
    return TWO_STREAMS and t.is_cuda


def refresh_packed(device, params=None):
    """Run the packed-weight table refresh now (on the current stream) if any cached entry is stale — called before
    the branches fork so that the refresh is ordered in front of both.
    params: the parameters of the model about to run, passed when that model sat idle for two or more optimiser epochs of ANOTHER
    model (its entries then dropped out of the table pass's candidates): their stale entries are taken along here, in front of the
    fork — otherwise the first stale read inside the fork would refresh them lazily on whichever stream got there first, and the
    sibling stream would read packed buffers with no dependency on that launch."""
    if params is not None:
        cur = _pack_recent[0]
        for p in params:
            ptr = p.data_ptr()
            for kind in (0, 1, 2, 3):
                e = _pack_entries.get((ptr, kind))
                if e is not None and e.wref() is p and e.stamp != (PACK_EPOCH, p._version):
                    cur.setdefault(id(e), e)
    for e in _pack_candidates():
        w = e.wref()
        if w is not None and w.device == device and e.stamp != (PACK_EPOCH, w._version):
            _pack_refresh_all(device)
            break
    if not torch.cuda.is_current_stream_capturing():
        _wc_refresh(device)


# Direct gradient sink (installed by tools.distributed.GradientBuckets): parameter gradients are written by the kernels
# straight into the flat-bucket views, the autograd Functions return None for them, and the 165 per-parameter
# AccumulateGrad add kernels of a step disappear.  Without a sink every Function returns ordinary gradient tensors.
GRAD_SINK = None
PRELU_DEFER = True         # test aid: False = every PReLU backward sums its slope gradient at once
BN_COUNTER_SINK = None      # list collecting the BatchNorm modules whose num_batches_tracked is due (tools.engine)


def _pgrad(param):
    """-> (tensor the kernel writes the gradient of ``param`` into, direct?)."""
    if GRAD_SINK is not None and param is not None:
        v = GRAD_SINK.take(param)
        if v is not None:
            return v, True
    return torch.empty_like(param), False


def _pret(param, g, direct):
    """Value a backward() returns for ``param``: None after a direct write (the sink is told the gradient landed)."""
    if direct:
        GRAD_SINK.done(param)
        return None
    return g


def _vox(x):
    """(B, D, H, W) extents and channel count of a channels-last 5-D tensor."""
    assert x.dim() == 5 and x.dtype in (torch.float32, torch.bfloat16), (x.shape, x.dtype)
    return x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.shape[4]


def pack_weights(w, mode):
    co, ci = w.shape[0], w.shape[1]
    taps = int(np.prod(w.shape[2:]))
    wp = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    rt.check(rt.lib().hupr_pack_conv_weights_f32(rt.ptr(_c(w)), rt.ptr(wp), co, ci, taps, mode, rt.stream()))
    return wp


def pack_weights_bf16(w, mode):
    co, ci = w.shape[0], w.shape[1]
    taps = int(np.prod(w.shape[2:]))
    wp = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
    rt.check(rt.lib().hupr_pack_conv_weights_bf16(rt.ptr(_c(w)), rt.ptr(wp), co, ci, taps, mode, rt.stream()))
    return wp


# ---- packed-weight cache -------------------------------------------------------------------------------------
# Parameters (leaf tensors that require grad) keep both packed layouts cached; the cache is stamped with PACK_EPOCH
# (bumped by optimisers that update weights through the C ABI, i.e. behind torch's version counter) and the tensor's
# own ``_version``.  The first stale hit of a step refreshes EVERY registered weight with one table-driven launch
# (hupr_pack_conv_weights_table) instead of ~160 small pack launches per training step.
import weakref

PACK_EPOCH = 0
PACK_CACHE = True          # debugging aid: False = repack on every call
_pack_entries = {}      # (storage address, kind) -> entry
_pack_table = None      # (device uint8 tensor, n, total) or None when dirty
# The table pass refreshes the entries that were READ during the current or the previous epoch (= optimiser step), not every weight
# the process ever registered: a process that holds several models (the GPU test suite: a dozen live networks by the time the pose
# fits run; an application with a training and an evaluation copy) repacked all of them after every optimiser step of one — 20 ->
# 34 ms per fit step inside the suite, most of it this module's Python loops over thousands of entries (round 5).  An entry that
# drops out is refreshed when it is read again (the stale path of ``_packed`` / ``_proj_cat`` / ``_head_w16_cached`` touches it first).
_pack_recent = [{}, {}]      # id(entry) -> entry: read during the current / the previous epoch
_pack_pinned = {}            # entries a captured inference graph reads (it baked their addresses in): refreshed after every update
_pack_sweeps = 0


def invalidate_packed():
    """Call after changing parameters behind torch's back (FusedAdam does)."""
    global PACK_EPOCH
    PACK_EPOCH += 1
    _pack_recent[1] = _pack_recent[0]
    _pack_recent[0] = {}


class _PackEntry:
    __slots__ = ("wref", "ptr", "kind", "shape", "wp", "stamp", "group")      # group: the epoch the entry was created in (~ its model)


def _touch(e):
    """Mark ``e`` as read in this epoch -> was it among the table pass's candidates already?  (False: a weight that dropped out —
    its model sat idle for two optimiser steps of another one —: the caller's refresh then sweeps the stale entries of that
    model, i.e. those created in the same epoch, in its one launch, so that the rest of the model does not come back one table
    launch per weight; sweeping EVERY stale entry of the process made the GPU suite 150 s slower again.)"""
    k = id(e)
    known = k in _pack_recent[0] or k in _pack_recent[1] or k in _pack_pinned
    _pack_recent[0][k] = e
    if torch.cuda.is_current_stream_capturing():
        _pack_pinned[k] = e
    return known


def _pack_candidates():
    cur, prev = _pack_recent
    return (list(cur.values()) + [e for k, e in prev.items() if k not in cur] +
            [e for k, e in _pack_pinned.items() if k not in cur and k not in prev])


def _pack_refresh_all(dev, full=None):
    global _pack_table, _pack_sweeps
    L = rt.lib()
    _pack_sweeps += 1
    if full is not None:                             # a dropped-out entry came back: take the stale entries of ITS model along — those
        for e in list(_pack_entries.values()):       # created in the same epoch (a model's first forward registers all its weights)
            w = e.wref()
            if (getattr(e, "group", None) == full and w is not None and w.device == dev and w.data_ptr() == e.ptr
                    and e.stamp != (PACK_EPOCH, w._version)):
                _pack_recent[0].setdefault(id(e), e)
    if _pack_sweeps % 256 == 0:                      # now and then: drop the entries (and packed copies) of weights that are gone
        for key in [k for k, e in _pack_entries.items() if e.wref() is None]:
            del _pack_entries[key]
    live = []
    for e in _pack_candidates():
        w = e.wref()
        if w is None or w.data_ptr() != e.ptr or w.device != dev:
            if w is None:
                _pack_entries.pop((e.ptr, e.kind), None) if _pack_entries.get((e.ptr, e.kind)) is e else None
                _pack_recent[0].pop(id(e), None)
                _pack_recent[1].pop(id(e), None)
                _pack_pinned.pop(id(e), None)
            continue
        live.append((e, w))
    if not live:
        return
    if _pack_table is None or _pack_table[3] != tuple(id(e) for e, _ in live):
        rec = np.zeros(len(live), dtype=np.dtype([("w", "<u8"), ("wp0", "<u8"), ("wp1", "<u8"), ("first", "<i8"), ("co", "<i4"),
                                                   ("ci", "<i4"), ("taps", "<i4"), ("kind", "<i4")]))
        first = 0
        blocks = []
        for i, (e, w) in enumerate(live):
            co, ci = w.shape[0], w.shape[1]
            taps = int(np.prod(w.shape[2:]))
            rec[i] = (w.data_ptr(), e.wp[0].data_ptr(), e.wp[1].data_ptr(), first, co, ci, taps, e.kind)
            cnt = co * ci * taps
            first += cnt
            if e.kind >= 2:
                # a projection weight -> its row block of the level's concatenated matrices (plain, query-scaled): block layout 3
                blocks.extend((i, 3, st) for st in range(0, cnt, 2048))
            elif e.kind == 1 and co % 32 == 0 and ci % 32 == 0 and taps <= 27:
                # 32 x 32 x taps tiles, both layouts per tile through LDS (hupr_k_pack_table, block layout 2)
                blocks.extend((i, 2, (c0 << 32) | i0) for c0 in range(0, co, 32) for i0 in range(0, ci, 32))
            else:
                for layout in (0, 1):
                    blocks.extend((i, layout, st) for st in range(0, cnt, 2048))
        blk = np.zeros(len(blocks), dtype=np.dtype([("entry", "<i4"), ("layout", "<i4"), ("start", "<i8")]))
        blk["entry"], blk["layout"], blk["start"] = zip(*blocks)
        tab = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        btab = torch.from_numpy(blk.view(np.uint8).copy()).to(dev)
        _pack_table = (tab, btab, len(blocks), tuple(id(e) for e, _ in live))
    tab, btab, n_blocks, _ = _pack_table
    rt.check(L.hupr_pack_conv_weights_table(rt.ptr(tab), rt.ptr(btab), n_blocks, rt.stream()))
    for e, w in live:
        e.stamp = (PACK_EPOCH, w._version)


def _packed(weight, mode, kind):
    """Packed layout ``mode`` (0 forward, 1 input gradient) of ``weight`` as fp32 (kind 0) or bf16 (kind 1).
    During a hipGraph capture: a TRAINING graph (grad enabled) repacks inside the graph — its own optimiser node changes the
    weights between replays; an INFERENCE graph (no_grad) reads the cached layouts when they are fresh — ~60 pack launches less
    per replay — and those buffers are refreshed IN PLACE by the next eager refresh (``refresh_packed`` after a weight update
    keeps a captured inference graph current, together with the in-place ``_wc_cache`` below)."""
    global _pack_table
    capturing = torch.cuda.is_current_stream_capturing()
    if not (PACK_CACHE and weight.is_leaf and weight.requires_grad and weight.is_contiguous()) or \
            (capturing and torch.is_grad_enabled()):
        return pack_weights_bf16(weight, mode) if kind else pack_weights(weight, mode)
    key = (weight.data_ptr(), kind)
    e = _pack_entries.get(key)
    if e is not None and (e.wref() is None or e.shape != tuple(weight.shape)):     # the address was recycled by another tensor
        e = None
    if capturing:
        if e is None or e.stamp != (PACK_EPOCH, weight._version):                  # nothing cached is created or refreshed mid-capture
            return pack_weights_bf16(weight, mode) if kind else pack_weights(weight, mode)
        _touch(e)
        return e.wp[mode]
    if e is None:
        e = _PackEntry()
        e.wref, e.ptr, e.kind, e.shape = weakref.ref(weight), weight.data_ptr(), kind, tuple(weight.shape)
        pk = pack_weights_bf16 if kind else pack_weights
        e.wp = (pk(weight, 0), pk(weight, 1))
        e.stamp = (PACK_EPOCH, weight._version)
        e.group = PACK_EPOCH
        _pack_entries[key] = e
        _pack_table = None
        _touch(e)
    elif e.stamp != (PACK_EPOCH, weight._version):
        _pack_refresh_all(weight.device, full=None if _touch(e) else getattr(e, "group", None))
        if e.stamp != (PACK_EPOCH, weight._version):          # not covered by the table pass (should not happen)
            pk = pack_weights_bf16 if kind else pack_weights
            pk_into = (pk(weight, 0), pk(weight, 1))
            e.wp[0].copy_(pk_into[0])                           # in place: captured inference graphs hold these addresses
            e.wp[1].copy_(pk_into[1])
            e.stamp = (PACK_EPOCH, weight._version)
    else:
        _touch(e)
    return e.wp[mode]


# The eight 1x1 projection weights of an MSCSA level as the two (4C, C) matrices its two GEMMs read — [phi_cross | theta_cross |
# phi_self | theta_self] per map — kept as ENTRIES OF THE PACK TABLE (kinds 2 / 3): the one table-driven launch after an optimiser
# step refreshes them with everything else, where rounds 1-4 concatenated them with two ATen launches per level and step.  Two
# copies per map: the plain one (backward GEMMs, GEMM-attention fallback) and the one whose query (theta) rows carry
# log2(e) for the QS attention kernels (csrc/attention_bf16.hip, kDeferBits).
_proj_cache = {}       # (addresses of the four weights) -> (Wc plain, Wc query-scaled, entries)
QS_ATTN = True             # test aid: False = the plain-Q kernels (fma per score)


def _proj_cat(ws, C):
    """-> (Wc, Wc_qs): (4C, C) fp32 concatenations of the four (C, C, 1, 1) projection weights ``ws`` of one map."""
    capturing = torch.cuda.is_current_stream_capturing()
    ok = PACK_CACHE and all(w.is_leaf and w.requires_grad and w.is_contiguous() for w in ws)
    key = tuple(w.data_ptr() for w in ws)
    ent = _proj_cache.get(key) if ok else None
    if ent is not None and any(e.wref() is not w for e, w in zip(ent[2], ws)):          # addresses recycled by other tensors
        ent = None
    fresh = ent is not None and all(e.stamp == (PACK_EPOCH, w._version) for e, w in zip(ent[2], ws))
    if not ok or (capturing and (torch.is_grad_enabled() or not fresh)):
        # not cacheable, or inside a capture that must not create / refresh cache entries (a training graph's own optimiser node
        # changes the weights between replays): computed in place, in the graph
        wc = torch.cat([w.detach().reshape(C, C) for w in ws], 0)
        wq = wc.clone()
        wq[C:2 * C] *= 1.4426950408889634
        wq[3 * C:] *= 1.4426950408889634
        return wc, wq
    if ent is None:
        global _pack_table
        for k in [k for k, v in _proj_cache.items() if any(e.wref() is None for e in v[2])]:
            del _proj_cache[k]
        wc = torch.empty((4 * C, C), dtype=torch.float32, device=ws[0].device)
        wq = torch.empty_like(wc)
        entries = []
        for j, w in enumerate(ws):
            e = _PackEntry()
            e.wref, e.ptr, e.kind, e.shape = weakref.ref(w), w.data_ptr(), 3 if j in (1, 3) else 2, tuple(w.shape)
            e.wp = (wc[j * C:(j + 1) * C], wq[j * C:(j + 1) * C])
            e.stamp = None
            e.group = PACK_EPOCH
            _pack_entries[(w.data_ptr(), e.kind)] = e
            entries.append(e)
        ent = _proj_cache[key] = (wc, wq, entries)
        _pack_table = None
        fresh = False
        created = True
    else:
        created = False
    known = all([_touch(e) for e in ent[2]])
    if not fresh:
        _pack_refresh_all(ws[0].device, full=None if (known or created) else ent[2][0].group)
    return ent[0], ent[1]


USE_FLASH = True       # bf16 mode: fused attention kernels where supported (C in {64,128}, N % 128 == 0)
USE_HALO = True        # bf16 mode: LDS halo-tiled kernel for 3x3(x3) "same" convolutions


def _halo_ok(x, k, pad, co=4):
    if _st.math != "bf16" or not USE_HALO or co % 4 != 0:
        return False
    B, Di, Hi, Wi, Ci = _vox(x)
    if x.dtype == torch.bfloat16 and Ci % 8 != 0:
        return False
    return bool(rt.lib().hupr_conv3x3_halo_supported(Di, Hi, Wi, Ci, k[0], k[1], k[2], pad[0], pad[1], pad[2]))


# BatchNorm statistics fused into the producing convolution (hupr_conv3x3_halo_bf16act_stats, 64-channel 3-D layers = the
# largest tensors of the step): the column sums wait here, keyed by the output's address, for the BatchNorm that consumes that
# tensor next (_bn_params pops them).  Round 3: the kernel keeps running per-lane-pair sums in LDS (plain read-add-write of a
# private slot, deferred epilogue intact) and reduces across lanes once per launch: +6 us on a 222 us launch instead of +14 us
# and the immediate epilogue, -0.16 ms / +0.7 % frames/s per step measured in interleaved same-box runs -> ON by default
# (CONV_STATS = False switches it off: the parity tests compare the two).
CONV_STATS = True
ATTN_LEVEL_BATCH = True    # test aid: False = one launch per attention at every level
ATTN_PROBE = None          # measurement hook (bench.py): (kind, B, N, C) -> (start, end) events around one attention's launches, or None
_conv_stats = {}


def _conv_raw(x, weight, mode, bias, res, co, k, pad, out_extent, out=None, stats=False, infer=False):
    """weight: parameter-layout tensor (Co', Ci', taps...) packed here (mode 0 forward / 1 input gradient).
    out: write here instead of a fresh tensor (may be ``res`` itself: the epilogue reads a residual element right
    before the same lane overwrites it)."""
    B, Di, Hi, Wi, Ci = _vox(x)
    Do, Ho, Wo = out_extent
    y = torch.empty((B, Do, Ho, Wo, co), dtype=x.dtype, device=x.device) if out is None else out
    ev = CONV_PROBE(x, co, k) if CONV_PROBE is not None else None
    abf = x.dtype == torch.bfloat16
    if _halo_ok(x, k, pad, co):
        assert res is None or res.dtype == x.dtype
        wp = _packed(weight, mode, 1)
        if ev is not None:
            ev[0].record()
        if (stats and CONV_STATS and abf and bias is None and res is None and out is None
                and rt.lib().hupr_conv3x3_halo_stats_supported(B, Di, Hi, Wi, Ci, co, k[0])):
            rows = rt.lib().hupr_conv3x3_halo_stats_rows()
            st = torch.empty((rows, 2, co), dtype=torch.float64, device=x.device)
            rt.check(rt.lib().hupr_conv3x3_halo_bf16act_stats(rt.ptr(x), rt.ptr(wp), rt.ptr(y), B, Di, Hi, Wi, Ci, Ci, co, co,
                                                               k[0], rt.ptr(st), rt.stream()))
            _conv_stats[y.data_ptr()] = (st, rows, B * Do * Ho * Wo, co)
            if ev is not None:
                ev[1].record()
            return y
        L = rt.lib()
        # inference on small grids (config C2: B = 1): the reduction is sliced over workgroups, partial sums through a workspace
        nws = L.hupr_conv3x3_halo_splitk_ws_bytes(B, Di, Hi, Wi, Ci, co, k[0]) if (abf and infer) else 0
        if nws:
            ws = workspace(nws, x.device)
            rt.check(L.hupr_conv3x3_halo_bf16act_ws(rt.ptr(x), rt.ptr(wp), rt.ptr(bias) if bias is not None else None,
                                                    rt.ptr(res) if res is not None else None, rt.ptr(y), B, Di, Hi, Wi, Ci, Ci, co, co, co,
                                                    k[0], rt.ptr(ws), ws.numel(), rt.stream()))
        else:
            fn = L.hupr_conv3x3_halo_bf16act if abf else L.hupr_conv3x3_halo_bf16
            rt.check(fn(rt.ptr(x), rt.ptr(wp), rt.ptr(bias) if bias is not None else None,
                        rt.ptr(res) if res is not None else None, rt.ptr(y), B, Di, Hi, Wi, Ci, Ci, co, co, co, k[0], rt.stream()))
        if ev is not None:
            ev[1].record()
        return y
    if abf:
        raise rt.HuprError("bf16-stored activations are only supported by the halo-tiled 3x3 convolutions "
                           "(shape %r, kernel %r); cast to fp32 first" % (tuple(x.shape), k))
    wp = _packed(weight, mode, 0)
    if ev is not None:
        ev[0].record()
    rt.check(_fn("conv_fwd")(
        rt.ptr(x), rt.ptr(wp), rt.ptr(bias) if bias is not None else None,
        rt.ptr(res) if res is not None else None, rt.ptr(y), B, Di, Hi, Wi, Ci, Ci, Do, Ho, Wo, co, co,
        co, k[0], k[1], k[2], pad[0], pad[1], pad[2], 0, rt.stream()))
    if ev is not None:
        ev[1].record()
    return y


class ConvPartial:
    """A K-sliced inference convolution that has not been summed yet: ``slices`` fp32 tensors (n, B, D, H, W, Co) for a consumer
    that sums them itself (``infer_tail``).  See hupr_conv3x3_halo_bf16act_partial in include/hupr.h."""
    __slots__ = ("slices", "n", "shape")

    def __init__(self, slices, n, shape):
        self.slices, self.n, self.shape = slices, n, shape


INFER_TAILS = True         # test aid


def infer_fast_ok(x):
    """Single-sample style inference on the bf16 path: no autograd, bf16-stored channels-last input, bf16 matrix pipe."""
    return INFER_TAILS and _st.math == "bf16" and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16


def conv_infer_sliced(shape, weight, pad):
    """Would ``conv_infer`` K-slice this convolution (input extent ``shape`` = (B, D, H, W, Ci))?"""
    B, D, H, W, Ci = shape
    k = _ksize(weight)
    return bool(rt.lib().hupr_conv3x3_halo_supported(D, H, W, Ci, k[0], k[1], k[2], pad[0], pad[1], pad[2])) and \
        rt.lib().hupr_conv3x3_halo_splitk_ws_bytes(B, D, H, W, Ci, weight.shape[0], k[0]) > 0


def conv_infer(x, weight, pad):
    """Bias-free 3x3(x3) "same" convolution under ``infer_fast_ok``: a ConvPartial where the reduction is K-sliced (small grids),
    else the stored bf16 tensor."""
    x = _c(x)
    k = _ksize(weight)
    B, D, H, W, Ci = _vox(x)
    co = weight.shape[0]
    L = rt.lib()
    nws = L.hupr_conv3x3_halo_splitk_ws_bytes(B, D, H, W, Ci, co, k[0]) if _halo_ok(x, k, pad, co) else 0
    if not nws:
        return conv(x, weight, None, None, pad)
    n = nws // (B * D * H * W * co * 4)
    part = torch.empty((n, B, D, H, W, co), dtype=torch.float32, device=x.device)
    rt.check(L.hupr_conv3x3_halo_bf16act_partial(rt.ptr(x), rt.ptr(_packed(weight, 0, 1)), B, D, H, W, Ci, Ci, co, k[0],
                                                 rt.ptr(part), nws, rt.stream()))
    return ConvPartial(part, n, (B, D, H, W, co))


def infer_tail(a, b=None, bn_a=None, bn_b=None, relu=False, prelu=None):
    """One launch for the elementwise tail of an inference block; ``a`` / ``b``: bf16 tensors or ConvPartials.
    BatchNorm form (bn_a given): relu?(bn_a(a) [+ bn_b(b)]) on running statistics; PReLU form: prelu(a [+ b])."""
    def side(t):
        if t is None:
            return None, 0
        return (rt.ptr(t.slices), t.n) if isinstance(t, ConvPartial) else (rt.ptr(_c(t)), 0)

    shape = a.shape
    C = shape[-1]
    M = int(np.prod(shape[:-1]))
    dev = (a.slices if isinstance(a, ConvPartial) else a).device
    y = torch.empty(tuple(shape), dtype=torch.bfloat16, device=dev)
    (x1, n1), (x2, n2) = side(a), side(b)
    bn = lambda m: (rt.ptr(m.weight), rt.ptr(m.bias), rt.ptr(m.running_mean), rt.ptr(m.running_var), float(m.eps)) \
        if m is not None else (None, None, None, None, 0.0)
    rt.check(rt.lib().hupr_infer_tail_bf16act(0 if bn_a is not None else 1, x1, n1, *bn(bn_a), x2, n2, *bn(bn_b),
                                              rt.ptr(prelu) if prelu is not None else None, 1 if relu else 0, rt.ptr(y), M, C,
                                              rt.stream()))
    return y


def _ksize(w):
    k = tuple(w.shape[2:])
    return (1,) + k if len(k) == 2 else k


@_math_scoped
class ConvFn(torch.autograd.Function):
    """Stride-1 convolution (nn.Conv3d / nn.Conv2d of the reference) with optional bias and fused
    residual add.  ``pad`` is (pd, ph, pw)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, pad, want_stats=False, infer=False):
        x = _c(x)
        k = _ksize(weight)
        B, Di, Hi, Wi, Ci = _vox(x)
        assert weight.shape[1] == Ci, (weight.shape, x.shape)
        out_extent = (Di + 2 * pad[0] - k[0] + 1, Hi + 2 * pad[1] - k[1] + 1, Wi + 2 * pad[2] - k[2] + 1)
        y = _conv_raw(x, weight, 0, bias, _c(res) if res is not None else None, weight.shape[0], k, pad, out_extent,
                      stats=want_stats, infer=infer)
        ctx.save_for_backward(x, weight)
        ctx.bias_ref = bias                       # only its identity/shape is needed (gradient destination)
        ctx.pad, ctx.k, ctx.has_bias, ctx.has_res = pad, k, bias is not None, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _c(dy)
        assert dy.dtype == x.dtype, (dy.dtype, x.dtype)
        k, pad = ctx.k, ctx.pad
        B, Di, Hi, Wi, Ci = _vox(x)
        _, Do, Ho, Wo, Co = _vox(dy)
        L = rt.lib()
        dx = dw = db = None
        dw_direct = db_direct = False
        if ctx.needs_input_grad[0]:
            co_pad = Co
            dyp, wsrc = dy, weight
            if Co % 32 != 0:                      # e.g. the 14-channel head: zero-pad the K axis
                co_pad = (Co + 31) // 32 * 32
                dyp = torch.zeros((B, Do, Ho, Wo, co_pad), dtype=dy.dtype, device=dy.device)
                dyp[..., :Co] = dy
                wsrc = torch.zeros((co_pad,) + tuple(weight.shape[1:]), dtype=torch.float32, device=dy.device)
                wsrc[:Co] = weight
            dpad = (k[0] - 1 - pad[0], k[1] - 1 - pad[1], k[2] - 1 - pad[2])
            dx = _conv_raw(dyp, wsrc, 1, None, None, Ci, k, dpad, (Di, Hi, Wi))      # packs [Ci][taps reversed][Co]
        if ctx.needs_input_grad[1]:
            dw, dw_direct = _pgrad(weight)
        if ctx.needs_input_grad[1] and _halo_ok(x, k, pad) and Ci % 32 == 0 and Co % 8 == 0:
            ws = workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, k[0]), x.device)
            fn = L.hupr_conv3x3_wgrad_halo_bf16act if x.dtype == torch.bfloat16 else L.hupr_conv3x3_wgrad_halo_bf16
            rt.check(fn(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, Di, Hi, Wi, Ci, Ci, Co, Co, k[0],
                        rt.ptr(ws), ws.numel(), rt.stream()))
        elif ctx.needs_input_grad[1] and x.dtype == torch.bfloat16:
            raise rt.HuprError("no bf16-activation weight-gradient kernel for shape %r" % (tuple(x.shape),))
        elif ctx.needs_input_grad[1]:
            nbytes = L.hupr_conv_wgrad_ws_bytes(B, Do, Ho, Wo, Ci, Co, k[0], k[1], k[2])
            ws = workspace(nbytes, x.device)
            rt.check(_fn("conv_wgrad")(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, Di, Hi, Wi, Ci, Ci, Do, Ho, Wo,
                                           Co, Co, k[0], k[1], k[2], pad[0], pad[1], pad[2], rt.ptr(ws),
                                           ws.numel(), rt.stream()))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db, db_direct = _pgrad(ctx.bias_ref)
            ws = workspace(L.hupr_bn_ws_bytes(Co), dy.device)
            rt.check(_act("colsum", dy)(rt.ptr(dy), B * Do * Ho * Wo, Co, rt.ptr(db), rt.ptr(ws), ws.numel(),
                                        rt.stream()))
        dres = dy if ctx.has_res else None
        return dx, _pret(weight, dw, dw_direct), _pret(ctx.bias_ref, db, db_direct), dres, None, None, None


@_math_scoped
class DualConvFn(torch.autograd.Function):
    """Two bias-free "same" 3x3(x3) convolutions of one bf16-stored map (the main[0] / downsample[0] pair of a
    BasicBlock3D, reference models/layers.py:55-65) as one autograd node, so that the two input gradients are summed
    in the second kernel's residual epilogue instead of by a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, w_a, w_b, pad, want_stats=False, infer=False):
        x = _c(x)
        k = _ksize(w_a)
        B, D, H, W, Ci = _vox(x)
        co = w_a.shape[0]
        assert w_b.shape == w_a.shape and _halo_ok(x, k, pad, co)
        y_a = _conv_raw(x, w_a, 0, None, None, co, k, pad, (D, H, W), stats=want_stats, infer=infer)
        y_b = _conv_raw(x, w_b, 0, None, None, co, k, pad, (D, H, W), stats=want_stats, infer=infer)
        ctx.save_for_backward(x, w_a, w_b)
        ctx.k, ctx.pad = k, pad
        return y_a, y_b

    @staticmethod
    def backward(ctx, dy_a, dy_b):
        x, w_a, w_b = ctx.saved_tensors
        dy = (_c(dy_a), _c(dy_b))
        k, pad = ctx.k, ctx.pad
        B, D, H, W, Ci = _vox(x)
        Co = w_a.shape[0]
        L = rt.lib()
        dx = None
        if ctx.needs_input_grad[0]:
            dpad = (k[0] - 1 - pad[0], k[1] - 1 - pad[1], k[2] - 1 - pad[2])
            dx = _conv_raw(dy[0], w_a, 1, None, None, Ci, k, dpad, (D, H, W))
            dx = _conv_raw(dy[1], w_b, 1, None, dx, Ci, k, dpad, (D, H, W), out=dx)
        grads = []
        if (ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and x.dtype == torch.bfloat16
                and L.hupr_conv3x3_wgrad_halo_dual_supported(B, D, H, W, Ci, Co, k[0])):
            # both weight gradients read the same x: one launch over 2 Co output channels, one reduction (same sums as two calls)
            (dwa, da), (dwb, db_) = _pgrad(w_a), _pgrad(w_b)
            ws = workspace(2 * L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, k[0]), x.device)
            rt.check(L.hupr_conv3x3_wgrad_halo_bf16act_dual(rt.ptr(x), rt.ptr(dy[0]), rt.ptr(dy[1]), rt.ptr(dwa), rt.ptr(dwb), B, D, H, W,
                                                            Ci, Ci, Co, Co, k[0], rt.ptr(ws), ws.numel(), rt.stream()))
            return dx, _pret(w_a, dwa, da), _pret(w_b, dwb, db_), None, None, None
        for i, w in enumerate((w_a, w_b)):
            if not ctx.needs_input_grad[1 + i]:
                grads.append(None)
                continue
            dw, direct = _pgrad(w)
            ws = workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, k[0]), x.device)
            fn = L.hupr_conv3x3_wgrad_halo_bf16act if x.dtype == torch.bfloat16 else L.hupr_conv3x3_wgrad_halo_bf16
            rt.check(fn(rt.ptr(x), rt.ptr(dy[i]), rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, k[0], rt.ptr(ws), ws.numel(),
                        rt.stream()))
            grads.append(_pret(w, dw, direct))
        return dx, grads[0], grads[1], None, None, None


def dual_conv(x, w_a, w_b, pad, stats=False):
    """(conv(x, w_a), conv(x, w_b)); fused input-gradient accumulation where the halo kernels apply."""
    k = _ksize(w_a)
    if w_a.shape == w_b.shape and _halo_ok(x, k, pad, w_a.shape[0]) and x.shape[-1] % 32 == 0 and w_a.shape[0] % 8 == 0:
        return DualConvFn.apply(x, w_a, w_b, tuple(pad), bool(stats), not torch.is_grad_enabled())
    infer = not torch.is_grad_enabled()
    return (ConvFn.apply(x, w_a, None, None, tuple(pad), bool(stats), infer),
            ConvFn.apply(x, w_b, None, None, tuple(pad), bool(stats), infer))


def conv(x, weight, bias=None, res=None, pad=(0, 0, 0), stats=False):
    """stats: a BatchNorm in training mode consumes the output next — let the convolution leave its column sums."""
    # (grad mode is read HERE: inside an autograd.Function's forward it is always off)
    return ConvFn.apply(x, weight, bias, res, tuple(pad), bool(stats), not torch.is_grad_enabled())


TMERGE_STREAM = True       # test aid: False = the generic mixed-storage convolution for the temporal merges


def _tmerge_fwd(x, weight, y):
    """merged map y (B,1,H,W,Co) fp32 of x (B,G,H,W,Ci): the LDS-DMA streaming kernel for bf16 64-channel maps (level 1),
    the generic mixed-storage convolution otherwise."""
    B, G, H, W, Ci = _vox(x)
    Co = weight.shape[0]
    L = rt.lib()
    if TMERGE_STREAM and x.dtype == torch.bfloat16 and L.hupr_tmerge_stream_supported(G, H * W, Ci, Co):
        rt.check(L.hupr_tmerge_fwd_stream_bf16(rt.ptr(x), rt.ptr(_packed(weight, 0, 1)), rt.ptr(y), B, G, H * W, Ci, Co, rt.stream()))
        return
    rt.check(L.hupr_conv_fwd_bf16_mixed(rt.ptr(x), int(x.dtype == torch.bfloat16), rt.ptr(_packed(weight, 0, 0)), None, rt.ptr(y), 0,
                                        B, G, H, W, Ci, Ci, 1, H, W, Co, Co, G, 1, 1, 0, 0, 0, rt.stream()))


def _tmerge_dgrad(dy, weight, dx):
    """dx (B,G,H,W,Ci) of the merge from the merged map's gradient dy (B,1,H,W,Co) fp32."""
    B, G, H, W, Ci = _vox(dx)
    Co = weight.shape[0]
    L = rt.lib()
    if TMERGE_STREAM and dx.dtype == torch.bfloat16 and L.hupr_tmerge_stream_supported(G, H * W, Ci, Co):
        rt.check(L.hupr_tmerge_dgrad_stream_bf16(rt.ptr(dy), rt.ptr(_packed(weight, 1, 1)), rt.ptr(dx), B, G, H * W, Ci, Co, rt.stream()))
        return
    rt.check(L.hupr_tmerge_dgrad_bf16(rt.ptr(dy), rt.ptr(_packed(weight, 1, 0)), rt.ptr(dx), int(dx.dtype == torch.bfloat16), B, G,
                                      H * W, Ci, Co, rt.stream()))


def _tmerge_wgrad(x, dy, weight):
    """-> (dw in the parameter layout, written directly into the gradient sink?)."""
    B, G, H, W, Ci = _vox(x)
    Co = weight.shape[0]
    L = rt.lib()
    dw, direct = _pgrad(weight)
    if TMERGE_STREAM and x.dtype == torch.bfloat16 and L.hupr_tmerge_wgrad_stream_supported(G, H * W, Ci, Co):
        ws = workspace(L.hupr_tmerge_wgrad_stream_ws_bytes(B, G, H * W, Ci, Co), x.device)
        rt.check(L.hupr_tmerge_wgrad_stream_bf16(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, G, H * W, Ci, Co, rt.ptr(ws), ws.numel(),
                                                 rt.stream()))
        return dw, direct
    ws = workspace(L.hupr_conv_wgrad_ws_bytes(B, 1, H, W, Ci, Co, G, 1, 1), x.device)
    rt.check(L.hupr_conv_wgrad_bf16_mixed(rt.ptr(x), int(x.dtype == torch.bfloat16), rt.ptr(dy), rt.ptr(dw), B, G, H, W, Ci, Ci, 1,
                                          H, W, Co, Co, G, 1, 1, 0, 0, 0, rt.ptr(ws), ws.numel(), rt.stream()))
    return dw, direct


@_math_scoped
class TemporalMergeFn(torch.autograd.Function):
    """Frame-axis merge: Conv3d with kernel (G,1,1), no padding and no bias on a (B,G,H,W,C) map (the
    l1temporalMerge / l2temporalMerge / temporalMerge of the reference, models/layers.py:195-197,218-220), taking
    the bf16-stored feature maps of the bf16-activation encoder directly: bf16 in, fp32 merged map out; the
    input gradient (G*B batched GEMMs — every frame slice sees exactly one tap) is written as bf16."""

    @staticmethod
    def forward(ctx, x, weight):
        x = _c(x)
        B, G, H, W, Ci = _vox(x)
        Co = weight.shape[0]
        assert tuple(weight.shape[1:]) == (Ci, G, 1, 1), (weight.shape, x.shape)
        y = torch.empty((B, 1, H, W, Co), dtype=torch.float32, device=x.device)
        _tmerge_fwd(x, weight, y)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _c(dy)
        B, G, H, W, Ci = _vox(x)
        Co = weight.shape[0]
        L = rt.lib()
        dx = dw = None
        direct = False
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _tmerge_dgrad(dy, weight, dx)
        if ctx.needs_input_grad[1]:
            dw, direct = _tmerge_wgrad(x, dy, weight)
        return dx, _pret(weight, dw, direct)


@_math_scoped
class MergeDownFn(torch.autograd.Function):
    """An encoder level map feeds TWO consumers (reference models/layers.py:212-217): its temporal merge and the next level's
    align_corners down-sampling.  As separate nodes autograd adds their two input gradients with a kernel of its own (three
    tensor passes over the largest activations of the network); as one node the merge's input gradient is written first and
    the resampling backward accumulates onto it (hupr_interp_linear_bwd_acc_*).  bf16-stored maps, bf16 math.
    -> (merged fp32 (B,1,H,W,Co), down-sampled bf16 (B, D', H', W', C))."""

    @staticmethod
    def forward(ctx, x, weight, size):
        x = _c(x)
        B, G, H, W, Ci = _vox(x)
        Co = weight.shape[0]
        assert tuple(weight.shape[1:]) == (Ci, G, 1, 1) and x.dtype == torch.bfloat16
        L = rt.lib()
        merged = torch.empty((B, 1, H, W, Co), dtype=torch.float32, device=x.device)
        _tmerge_fwd(x, weight, merged)
        Do, Ho, Wo = size
        down = torch.empty((B, Do, Ho, Wo, Ci), dtype=x.dtype, device=x.device)
        rt.check(L.hupr_interp_linear_fwd_bf16act(rt.ptr(x), rt.ptr(down), B, G, H, W, Do, Ho, Wo, Ci, Ci, Ci, rt.stream()))
        ctx.save_for_backward(x, weight)
        ctx.size = size
        return merged, down

    @staticmethod
    def backward(ctx, d_merged, d_down):
        x, weight = ctx.saved_tensors
        B, G, H, W, Ci = _vox(x)
        Co = weight.shape[0]
        Do, Ho, Wo = ctx.size
        L = rt.lib()
        d_merged, d_down = _c(d_merged), _c(d_down)
        dx = torch.empty_like(x)
        _tmerge_dgrad(d_merged, weight, dx)
        rt.check(L.hupr_interp_linear_bwd_acc_bf16act(rt.ptr(d_down), rt.ptr(dx), B, G, H, W, Do, Ho, Wo, Ci, Ci, Ci, rt.stream()))
        dw = None
        direct = False
        if ctx.needs_input_grad[1]:
            dw, direct = _tmerge_wgrad(x, d_merged, weight)
        return dx, _pret(weight, dw, direct), None


def merge_down_ok(x):
    return _st.math == "bf16" and x.dtype == torch.bfloat16 and x.is_cuda


def temporal_merge(x, weight):
    """(B,G,H,W,C) -> (B,1,H,W,Co) fp32.  bf16-stored maps go through the mixed-storage kernels (bf16 math only);
    fp32 maps through the generic convolution."""
    if x.dtype == torch.bfloat16:
        if _st.math != "bf16":
            raise rt.HuprError("bf16-stored activations need the bf16 math mode")
        return TemporalMergeFn.apply(x, weight)
    return ConvFn.apply(x, weight, None, None, (0, 0, 0))


# ----------------------------------------------------------------------------------------------
# BatchNorm (+ReLU), and the BasicBlock3D tail  relu(bn_a(x1) + bn_b(x2))
# ----------------------------------------------------------------------------------------------
def _bn_params(x, bn, training, need_bwd=True):
    """-> (scale, shift, save_mean, save_invstd) for one nn.BatchNorm3d-like parameter holder."""
    L = rt.lib()
    C = x.shape[-1]
    M = x.numel() // C
    dev = x.device
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    mean = torch.empty_like(scale)
    invstd = torch.empty_like(scale)
    fused = _conv_stats.pop(x.data_ptr(), None)
    if fused is not None and not (training and fused[2] == M and fused[3] == C):
        fused = None
    if training:
        track = bn.running_mean is not None
        if fused is not None:           # the producing convolution left the column sums: finalize only
            rt.check(L.hupr_bn_train_finalize_f32(
                rt.ptr(fused[0]), fused[1], M, C, rt.ptr(bn.weight), rt.ptr(bn.bias),
                rt.ptr(bn.running_mean) if track else None, rt.ptr(bn.running_var) if track else None,
                float(bn.momentum), float(bn.eps), rt.ptr(mean), rt.ptr(invstd), rt.ptr(scale), rt.ptr(shift), rt.stream()))
        else:
            ws = workspace(L.hupr_bn_ws_bytes(C), dev)
            rt.check(_act("bn_train_stats", x)(
                rt.ptr(x), M, C, rt.ptr(bn.weight), rt.ptr(bn.bias),
                rt.ptr(bn.running_mean) if track else None, rt.ptr(bn.running_var) if track else None,
                float(bn.momentum), float(bn.eps), rt.ptr(mean), rt.ptr(invstd), rt.ptr(scale), rt.ptr(shift),
                rt.ptr(ws), ws.numel(), rt.stream()))
        if track and bn.num_batches_tracked is not None:
            if BN_COUNTER_SINK is not None:
                BN_COUNTER_SINK.append(bn)          # the engine bumps all counters of a step with one launch
            else:
                bn.num_batches_tracked.add_(1)
    else:
        rt.check(L.hupr_bn_eval_params_f32(rt.ptr(bn.weight), rt.ptr(bn.bias), rt.ptr(bn.running_mean),
                                           rt.ptr(bn.running_var), float(bn.eps), C, rt.ptr(scale),
                                           rt.ptr(shift), rt.stream()))
        if need_bwd:                         # only a backward pass through eval-mode statistics reads these two
            mean.copy_(bn.running_mean)
            invstd = torch.rsqrt(bn.running_var + bn.eps)
    return scale, shift, mean, invstd


def _bn_params_pair(x1, bn1, x2, bn2):
    """Training-mode coefficients of the two BatchNorms of a block tail when BOTH producing convolutions left their column sums:
    one finalize launch for the pair (hupr_bn_train_finalize2_f32).  None: not applicable — the caller takes them one by one."""
    f1, f2 = _conv_stats.get(x1.data_ptr()), _conv_stats.get(x2.data_ptr())
    C = x1.shape[-1]
    M = x1.numel() // C
    if (f1 is None or f2 is None or x1.shape != x2.shape or x1.data_ptr() == x2.data_ptr()
            or (f1[2], f1[3]) != (M, C) or (f2[2], f2[3]) != (M, C)):
        return None
    _conv_stats.pop(x1.data_ptr())
    _conv_stats.pop(x2.data_ptr())
    dev = x1.device
    out = [[torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4)] for _ in range(2)]      # scale, shift, mean, invstd
    args = []
    for f, bn, (scale, shift, mean, invstd) in ((f1, bn1, out[0]), (f2, bn2, out[1])):
        track = bn.running_mean is not None
        args += [rt.ptr(f[0]), f[1], rt.ptr(bn.weight), rt.ptr(bn.bias), rt.ptr(bn.running_mean) if track else None,
                 rt.ptr(bn.running_var) if track else None, float(bn.momentum), float(bn.eps), rt.ptr(mean), rt.ptr(invstd),
                 rt.ptr(scale), rt.ptr(shift)]
        if track and bn.num_batches_tracked is not None:
            if BN_COUNTER_SINK is not None:
                BN_COUNTER_SINK.append(bn)
            else:
                bn.num_batches_tracked.add_(1)
    rt.check(rt.lib().hupr_bn_train_finalize2_f32(*args, M, C, rt.stream()))
    return tuple(out[0]), tuple(out[1])


def _bn_bwd(dy, x, mean, invstd, gamma, training, beta=None, fwd=None):
    """-> (dx, dgamma, dbeta) as backward() return values (None for parameters written through the gradient sink).
    fwd = (scale, shift) of the forward pass of a BatchNorm + ReLU: the ReLU mask is recomputed from x (the very expression the
    forward evaluated) instead of read back as a third tensor; None: no ReLU."""
    L = rt.lib()
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    dg, dg_direct = _pgrad(gamma)
    db, db_direct = _pgrad(beta) if beta is not None else (torch.empty_like(gamma), False)
    ws = workspace(L.hupr_bn_ws_bytes(C), x.device)
    assert dy.dtype == x.dtype
    if fwd is not None:
        rt.check(_act("bn_bwd_remask", x)(rt.ptr(dy), rt.ptr(fwd[0]), rt.ptr(fwd[1]), rt.ptr(x), rt.ptr(mean), rt.ptr(invstd),
                                          rt.ptr(gamma), rt.ptr(dx), rt.ptr(dg), rt.ptr(db), M, C, 1 if training else 0,
                                          rt.ptr(ws), ws.numel(), rt.stream()))
    else:
        rt.check(_act("bn_bwd", x)(rt.ptr(dy), None, rt.ptr(x), rt.ptr(mean),
                                   rt.ptr(invstd), rt.ptr(gamma), rt.ptr(dx), rt.ptr(dg), rt.ptr(db), M, C,
                                   1 if training else 0, rt.ptr(ws), ws.numel(), rt.stream()))
    return dx, _pret(gamma, dg, dg_direct), _pret(beta, db, db_direct)


@_math_scoped
class BNActFn(torch.autograd.Function):
    """y = [relu](batch_norm(x)).  ``bn`` is the parameter holder (running stats updated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, training, relu, no_bwd=False):
        # no_bwd: the caller runs under torch.no_grad() (grad mode is always off in here, and needs_input_grad reflects the
        # tensors' flags whatever the mode) — eval-mode statistics then skip what only a backward pass reads
        x = _c(x)
        C = x.shape[-1]
        if no_bwd and not training:          # inference: coefficients and apply in one launch
            y = torch.empty_like(x)
            rt.check(_act("bn_eval_act", x)(rt.ptr(x), rt.ptr(bn.weight), rt.ptr(bn.bias), rt.ptr(bn.running_mean),
                                            rt.ptr(bn.running_var), float(bn.eps), None, None, None, None, None, 0.0, rt.ptr(y),
                                            x.numel() // C, C, 1 if relu else 0, rt.stream()))
            return y
        scale, shift, mean, invstd = _bn_params(x, bn, training, not no_bwd)
        y = torch.empty_like(x)
        rt.check(_act("scale_shift_act", x)(rt.ptr(x), rt.ptr(scale), rt.ptr(shift), None, None, None,
                                            rt.ptr(y), x.numel() // C, C, 1 if relu else 0, rt.stream()))
        ctx.save_for_backward(x, mean, invstd, gamma, scale if relu else None, shift if relu else None)
        ctx.beta_ref = beta
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, gamma, scale, shift = ctx.saved_tensors
        dx, dg, db = _bn_bwd(_c(dy), x, mean, invstd, gamma, ctx.training, ctx.beta_ref,
                             fwd=(scale, shift) if scale is not None else None)
        return dx, dg, db, None, None, None, None


@_math_scoped
class BNAddBNReLUFn(torch.autograd.Function):
    """y = relu(bn_a(x1) + bn_b(x2)) — the tail of BasicBlock3D.forward (models/layers.py:66-70)."""

    @staticmethod
    def forward(ctx, x1, g1, b1, bn1, x2, g2, b2, bn2, training, no_bwd=False):
        x1, x2 = _c(x1), _c(x2)
        if no_bwd and not training:          # inference: both coefficient sets and the apply in one launch
            assert x1.dtype == x2.dtype
            C = x1.shape[-1]
            y = torch.empty_like(x1)
            rt.check(_act("bn_eval_act", x1)(rt.ptr(x1), rt.ptr(bn1.weight), rt.ptr(bn1.bias), rt.ptr(bn1.running_mean),
                                             rt.ptr(bn1.running_var), float(bn1.eps), rt.ptr(x2), rt.ptr(bn2.weight),
                                             rt.ptr(bn2.bias), rt.ptr(bn2.running_mean), rt.ptr(bn2.running_var), float(bn2.eps),
                                             rt.ptr(y), x1.numel() // C, C, 1, rt.stream()))
            return y
        pair = _bn_params_pair(x1, bn1, x2, bn2) if training else None
        if pair is not None:
            (s1, t1, m1, i1), (s2, t2, m2, i2) = pair
        else:
            s1, t1, m1, i1 = _bn_params(x1, bn1, training, not no_bwd)
            s2, t2, m2, i2 = _bn_params(x2, bn2, training, not no_bwd)
        C = x1.shape[-1]
        y = torch.empty_like(x1)
        assert x1.dtype == x2.dtype
        rt.check(_act("scale_shift_act", x1)(rt.ptr(x1), rt.ptr(s1), rt.ptr(t1), rt.ptr(x2), rt.ptr(s2),
                                             rt.ptr(t2), rt.ptr(y), x1.numel() // C, C, 1, rt.stream()))
        ctx.save_for_backward(x1, x2, m1, i1, g1, m2, i2, g2, s1, t1, s2, t2)
        ctx.beta_refs = (b1, b2)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, m1, i1, g1, m2, i2, g2, s1, t1, s2, t2 = ctx.saved_tensors
        dy = _c(dy)
        # both branches share dy and the ReLU mask: one statistics pass + one apply pass for the pair
        L = rt.lib()
        C = x1.shape[-1]
        M = x1.numel() // C
        assert dy.dtype == x1.dtype == x2.dtype
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        b1, b2 = ctx.beta_refs
        dg1, dg1_d = _pgrad(g1)
        db1, db1_d = _pgrad(b1)
        dg2, dg2_d = _pgrad(g2)
        db2, db2_d = _pgrad(b2)
        ws = workspace(L.hupr_bn_ws_bytes(C), x1.device)
        # ReLU mask recomputed from x1, x2 and the forward coefficients
        rt.check(_act("bn_bwd2_remask", x1)(rt.ptr(dy), rt.ptr(x1), rt.ptr(s1), rt.ptr(t1), rt.ptr(m1), rt.ptr(i1), rt.ptr(g1),
                                            rt.ptr(x2), rt.ptr(s2), rt.ptr(t2), rt.ptr(m2), rt.ptr(i2), rt.ptr(g2), rt.ptr(dx1),
                                            rt.ptr(dx2), rt.ptr(dg1), rt.ptr(db1), rt.ptr(dg2), rt.ptr(db2), M, C,
                                            1 if ctx.training else 0, rt.ptr(ws), ws.numel(), rt.stream()))
        return (dx1, _pret(g1, dg1, dg1_d), _pret(b1, db1, db1_d), None, dx2, _pret(g2, dg2, dg2_d), _pret(b2, db2, db2_d),
                None, None, None)


@_math_scoped
class PReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = _c(x)
        y = torch.empty_like(x)
        rt.check(_act("prelu_fwd", x)(rt.ptr(x), rt.ptr(alpha), rt.ptr(y), x.numel(), rt.stream()))
        ctx.save_for_backward(x, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        L = rt.lib()
        dx = torch.empty_like(x)
        da, da_direct = _pgrad(alpha)
        dy = _c(dy)
        assert dy.dtype == x.dtype
        if da_direct and PRELU_DEFER and getattr(GRAD_SINK, "can_defer", None) is not None and GRAD_SINK.can_defer(x.device):
            # the slope gradient's final sum waits until its gradient bucket is complete: the twelve of a step then take one launch
            part = torch.empty(L.hupr_prelu_ws_bytes() // 8, dtype=torch.float64, device=x.device)
            npart = ctypes.c_int(0)
            rt.check(_act("prelu_bwd_partials", x)(rt.ptr(dy), rt.ptr(x), rt.ptr(alpha), rt.ptr(dx), x.numel(), rt.ptr(part),
                                                   part.numel() * 8, ctypes.byref(npart), rt.stream()))
            GRAD_SINK.defer_sum(alpha, part, npart.value, da)
            return dx, None
        ws = workspace(L.hupr_prelu_ws_bytes(), x.device)
        rt.check(_act("prelu_bwd", x)(rt.ptr(dy), rt.ptr(x), rt.ptr(alpha), rt.ptr(dx), rt.ptr(da), x.numel(),
                                      rt.ptr(ws), ws.numel(), rt.stream()))
        return dx, _pret(alpha, da, da_direct)


# ----------------------------------------------------------------------------------------------
@_math_scoped
class MNetFn(torch.autograd.Function):
    """x (B,G,F,2,R,A,E) — or its elevation mean as planes (B,G,16,R,A) — -> (B, G, R, A, 32) channels-last, depth = group frame."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype=torch.float32):
        x = _c(x)
        if weight.shape != (32, 2, 2, 1, 1):
            raise ValueError("MNet kernel is specialised for F=8, 2 (re/im), E=8, 32 filters")
        from_means = x.dim() == 5            # (B, G, 16, R, A): the fused loader's elevation-mean planes (fft_chain_loader_means)
        if from_means:
            B, G, P16, R, A = x.shape
            if P16 != 16:
                raise ValueError("elevation-mean input must be (B, G, 16, R, A)")
        else:
            B, G, F, two, R, A, E = x.shape
            if (F, two, E) != (8, 2, 8):
                raise ValueError("MNet kernel is specialised for F=8, 2 (re/im), E=8, 32 filters")
        out = torch.empty((B, G, R, A, 32), dtype=out_dtype, device=x.device)
        # training: keep the 16 elevation means per pixel (1/8 of x) — the backward pass recomputes everything from them
        need = weight.requires_grad or bias.requires_grad
        means = torch.empty((B * G, R * A, 16), dtype=torch.float32, device=x.device) if need else None
        rt.check(_act("mnet_fwd_means" if from_means else "mnet_fwd", out)(
            rt.ptr(x), rt.ptr(_c(weight)), rt.ptr(bias), rt.ptr(out), rt.ptr(means) if need else None, B * G, R * A, rt.stream()))
        ctx.save_for_backward(means, weight, bias)
        ctx.geom = (B, G, R, A)
        return out

    @staticmethod
    def backward(ctx, dy):
        means, weight, bias = ctx.saved_tensors
        L = rt.lib()
        B, G, R, A = ctx.geom
        dw, dw_direct = _pgrad(weight)
        db, db_direct = _pgrad(bias)
        ws = workspace(L.hupr_mnet_bwd_ws_bytes(), dy.device)
        dy = _c(dy)
        rt.check(_act("mnet_bwd", dy)(None, rt.ptr(means), rt.ptr(_c(weight)), rt.ptr(bias), rt.ptr(dy), rt.ptr(dw), rt.ptr(db),
                                      B * G, R * A, rt.ptr(ws), ws.numel(), rt.stream()))
        return None, _pret(weight, dw, dw_direct), _pret(bias, db, db_direct), None


def _ld_view_ok(t, C):
    """``t`` (B, D, H, W, C) is a channel slice of a wider contiguous channels-last tensor (row stride = t.stride(3) elements, dense
    voxel order): kernels with a leading-dimension argument read / write it in place."""
    B, D, H, W, _ = t.shape
    ld = t.stride(3)
    return (t.stride(4) == 1 and ld >= C and t.stride(2) == W * ld and t.stride(1) == H * W * ld
            and (B == 1 or t.stride(0) == D * H * W * ld) and t.data_ptr() % 16 == 0 and ld % 8 == 0)


@_math_scoped
class InterpFn(torch.autograd.Function):
    """align_corners=True linear resampling of a channels-last (B,D,H,W,C) tensor to ``size``.  ``out`` (optional): a channel
    slice of a wider channels-last buffer to write into (the decoder's concatenated input: no concatenation copy); the incoming
    gradient may be such a slice too and is read in place."""

    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        B, Di, Hi, Wi, C = _vox(x)
        Do, Ho, Wo = size
        out = _interp_out.pop("out", None)           # (a side channel, not an argument: an argument returned as the output would be an
        if out is None:                               # input alias in autograd's eyes)
            y, ld = torch.empty((B, Do, Ho, Wo, C), dtype=x.dtype, device=x.device), C
        else:
            assert tuple(out.shape) == (B, Do, Ho, Wo, C) and out.dtype == x.dtype and _ld_view_ok(out, C), (out.shape, out.stride())
            y, ld = out, out.stride(3)
        rt.check(_act("interp_linear_fwd", x)(rt.ptr(x), rt.ptr_ld(y), B, Di, Hi, Wi, Do, Ho, Wo, C, C, ld, rt.stream()))
        ctx.in_shape = (B, Di, Hi, Wi, C)
        ctx.size = size
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Di, Hi, Wi, C = ctx.in_shape
        Do, Ho, Wo = ctx.size
        if dy.is_contiguous() or not _ld_view_ok(dy, C):
            dy, ld = _c(dy), C
        else:
            ld = dy.stride(3)
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        rt.check(_act("interp_linear_bwd", dy)(rt.ptr_ld(dy), rt.ptr(dx), B, Di, Hi, Wi, Do, Ho, Wo, C, C, ld, rt.stream()))
        return dx, None


class JoinFn(torch.autograd.Function):
    """The decoder's channel concatenation (reference models/layers.py:166-178) without a copy: the parts were WRITTEN as adjacent
    channel slices of ``wide`` by their producers (InterpFn / MSCSALevelFn with an output placement), so the forward just hands
    ``wide`` on, and the backward hands each producer its slice of the gradient (views; the producers' backward kernels read them in
    place).  Rounds 1-4 paid a concatenation kernel per decoder stage and two ``contiguous`` copies of gradient slices per step."""

    @staticmethod
    def forward(ctx, wide, *parts):
        off = 0
        for t in parts:
            assert t.data_ptr() == wide.data_ptr() + off * wide.element_size() and t.stride(3) == wide.shape[-1], "parts must be slices of wide"
            off += t.shape[-1]
        assert off == wide.shape[-1]
        ctx.widths = [t.shape[-1] for t in parts]
        return wide.view(wide.shape)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(g[..., off:off + w])
            off += w
        return (None,) + tuple(outs)


_interp_out = {}


def interp(x, size, out=None):
    """out: optional output placement (see InterpFn)."""
    _interp_out.clear()
    if out is not None:
        _interp_out["out"] = out
    return InterpFn.apply(x, tuple(size))


def _cast(x, dtype):
    x = _c(x)
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    if dtype == torch.bfloat16:
        rt.check(rt.lib().hupr_cast_f32_to_bf16(rt.ptr(x), rt.ptr(y), x.numel(), rt.stream()))
    else:
        rt.check(rt.lib().hupr_cast_bf16_to_f32(rt.ptr(x), rt.ptr(y), x.numel(), rt.stream()))
    return y


@_math_scoped
class CastFn(torch.autograd.Function):
    """Boundary of the bf16-activation region: y = x in ``dtype``; the gradient comes back in x's dtype."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.in_dtype = x.dtype
        return _cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return _cast(dy, ctx.in_dtype), None


def cast(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


def to_act(x, decoder=False):
    """Region border: ``x`` in the activation storage type of the current mode (bf16 inside bf16-activation regions)."""
    want = torch.bfloat16 if (act_bf16() and (ACT_BF16_DECODER or not decoder)) else torch.float32
    return cast(x, want)


# ----------------------------------------------------------------------------------------------
def gemm(ta, tb, A, B, M, N, K, lda, ldb, batch, a_bs, b_bs, out=None, res=None, accumulate=False, math=None):
    """Batched row-major fp32 GEMM on raw strides; returns C (batch, M, N) dense.  ``math`` overrides the global matrix-pipe
    mode for this call ("f32" keeps an operator on the exact fp32 pipe inside a bf16 run)."""
    if out is None:
        out = torch.empty((batch, M, N), dtype=torch.float32, device=A.device)
    fn = _fn("gemm") if math is None else getattr(rt.lib(), "hupr_gemm_%s" % math)
    rt.check(fn(ta, tb, rt.ptr(A), rt.ptr(B), rt.ptr(out), M, N, K, lda, ldb, N, batch, a_bs, b_bs,
                                   M * N, rt.ptr(res) if res is not None else None, N, M * N if res is not None else 0,
                                   1 if accumulate else 0, rt.stream()))
    return out


def _attn_ws(B, N, C, device):
    """Workspace of the split-key forward (small batches: the plain grid would leave most CUs idle), or None."""
    nbytes = rt.lib().hupr_attn_fwd_split_ws_bytes(B, N, C)
    return workspace(nbytes, device) if nbytes else None


@_math_scoped
class AttentionFn(torch.autograd.Function):
    """MSCSA attention (models/layers.py:126-133) on token-major tensors (B, N, C):
    S[j,k] = sum_c K[j,c] Q[k,c];  P = softmax over keys j;  out[k,c] = sum_j P[j,k] V[j,c] (+ V[k,c])."""

    @staticmethod
    def forward(ctx, k, q, v, residual):
        k, q, v = _c(k), _c(q), _c(v)
        B, N, C = v.shape
        L = rt.lib()
        ctx.flash = _st.math == "bf16" and USE_FLASH and bool(L.hupr_attn_flash_supported(N, C))
        if ctx.flash:
            out = torch.empty_like(v)
            lse = torch.empty((B, N), dtype=torch.float32, device=v.device)
            # bf16 copies of the MFMA operands: each is re-read by every workgroup (N / 128 times), and the kernels
            # would round them to bf16 per tile anyway — same bits, half the traffic, also half the saved activations
            kb, qb, vb = _cast(k, torch.bfloat16), _cast(q, torch.bfloat16), _cast(v, torch.bfloat16)
            ws = _attn_ws(B, N, C, v.device)
            rt.check(L.hupr_attn_fwd_bf16in_ld_ws(rt.ptr(kb), C, rt.ptr(qb), C, rt.ptr(vb), rt.ptr(v) if residual else None,
                                                  rt.ptr(out), rt.ptr(lse), None, 0, B, N, C, rt.ptr(ws) if ws is not None else None,
                                                  ws.numel() if ws is not None else 0, rt.stream()))
            ctx.save_for_backward(kb, qb, vb, v, out, lse)
            ctx.residual = residual
            return out
        # St[kq][j] = q . k  -> softmax over j is a row softmax
        P = gemm(0, 1, q, k, N, N, C, C, C, B, N * C, N * C)
        rt.check(L.hupr_softmax_rows_f32(rt.ptr(P), B * N, N, rt.stream()))
        out = gemm(0, 0, P, v, N, C, N, N, C, B, N * N, N * C, res=v if residual else None)
        ctx.save_for_backward(k, q, v, P)
        ctx.residual = residual
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.flash:
            kb, qb, vb, v, out, lse = ctx.saved_tensors
            dout = _c(dout)
            B, N, C = v.shape
            dk, dq, dv = torch.empty_like(v), torch.empty_like(v), torch.empty_like(v)
            dq_scr = torch.empty((B, N), dtype=torch.float32, device=v.device)
            gb = _cast(dout, torch.bfloat16)
            rt.check(rt.lib().hupr_attn_bwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(gb), rt.ptr(v), rt.ptr(out),
                                                  rt.ptr(dout), rt.ptr(lse), rt.ptr(dk), rt.ptr(dq), rt.ptr(dv),
                                                  rt.ptr(dq_scr), B, N, C, 1 if ctx.residual else 0, rt.stream()))
            return dk, dq, dv, None
        k, q, v, P = ctx.saved_tensors
        dout = _c(dout)
        B, N, C = v.shape
        L = rt.lib()
        # dV[j][c] = sum_kq P[kq][j] dout[kq][c]  (+ dout for the residual path)
        dv = gemm(1, 0, P, dout, N, C, N, N, C, B, N * N, N * C, res=dout if ctx.residual else None)
        # dP[kq][j] = sum_c dout[kq][c] v[j][c]
        dS = gemm(0, 1, dout, v, N, N, C, C, C, B, N * C, N * C)
        rt.check(L.hupr_softmax_rows_bwd_f32(rt.ptr(P), rt.ptr(dS), B * N, N, rt.stream()))
        dq = gemm(0, 0, dS, k, N, C, N, N, C, B, N * N, N * C)      # dQ[kq][c] = sum_j dS[kq][j] k[j][c]
        dk = gemm(1, 0, dS, q, N, C, N, N, C, B, N * N, N * C)      # dK[j][c] = sum_kq dS[kq][j] q[kq][c]
        return dk, dq, dv, None


_level_cat_out = {}      # {"out": view}: where the NEXT fused level writes its concatenated bf16 output (models/layers.py sets it)
CAT_FUSION = True          # fused levels return their maps concatenated as bf16
PROJ_STREAM = True         # test aid: False = the level's projections through the generic implicit-GEMM engine


def mscsa_level_fused_ok(ra):
    """One MSCSA level can run as MSCSALevelFn (bf16 math; any (N, C) — shapes without a fused attention kernel keep the
    GEMM / row-softmax attention core inside the node)."""
    B, _, H, W, C = ra.shape
    return _st.math == "bf16" and ra.dtype == torch.float32 and C % 8 == 0


CAT_INPLACE = True         # test aid: False = concatenation copies per decoder stage


def level_cat_placement_ok(ra):
    """The fused level node of map ``ra`` returns ONE concatenated bf16 tensor and can write it into a slice of a wider buffer."""
    B, _, H, W, C = ra.shape
    return CAT_INPLACE and mscsa_level_fused_ok(ra) and USE_FLASH and CAT_FUSION and bool(rt.lib().hupr_attn_flash_supported(H * W, C))


# Derived inference constants (concatenated projection weights, the zero-padded head filter): one entry per set of source
# parameters, keyed by their addresses, stamped like the packed layouts and REFRESHED IN PLACE when stale — a captured
# inference graph that baked in an entry's address keeps reading current values once ``refresh_packed`` has run after a weight
# update (the first version keyed entries by epoch, so an update orphaned the tensor a graph was still reading,
# and ``clear()`` at 64 entries could free it).  Entries die with their parameters (weak references), never by count.
_wc_cache = {}
ATTN_BATCH = True          # test aid: False = one launch pair per attention in single-sample inference


class _WcEntry:
    __slots__ = ("wrefs", "stamp", "build", "t")


def _wc_stamp(ws):
    return (PACK_EPOCH,) + tuple(w._version for w in ws)


def _wc_get(tag, ws, build):
    """-> the cached constant ``build(ws, out)`` derives from the parameters ``ws``.  ``build(ws, None)`` returns a fresh tensor,
    ``build(ws, t)`` refills ``t`` in place.  Nothing is created or refreshed during a capture (a fresh entry would live in the
    graph's private pool; a stale one means the caller skipped ``refresh_packed``): the capture then computes its own copy
    inside the graph, which is always current."""
    key = (tag,) + tuple(w.data_ptr() for w in ws)
    e = _wc_cache.get(key)
    if e is not None and any(r() is not w for r, w in zip(e.wrefs, ws)):            # an address recycled by another tensor
        e = None
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing:
        return e.t if e is not None and e.stamp == _wc_stamp(ws) else build(ws, None)
    if e is None:
        for k in [k for k, v in _wc_cache.items() if any(r() is None for r in v.wrefs)]:
            del _wc_cache[k]
        e = _WcEntry()
        e.wrefs, e.build, e.t = tuple(weakref.ref(w) for w in ws), build, build(ws, None)
        e.stamp = _wc_stamp(ws)
        _wc_cache[key] = e
    elif e.stamp != _wc_stamp(ws):
        build(ws, e.t)
        e.stamp = _wc_stamp(ws)
    return e.t


def _wc_refresh(device):
    """Refill every stale derived constant on ``device`` in place (part of ``refresh_packed``)."""
    for k in list(_wc_cache):
        e = _wc_cache[k]
        ws = [r() for r in e.wrefs]
        if any(w is None for w in ws):
            del _wc_cache[k]
        elif ws[0].device == device and e.stamp != _wc_stamp(ws):
            e.build(ws, e.t)
            e.stamp = _wc_stamp(ws)


def _cat_build(ws, out):
    C = ws[0].shape[0]
    parts = [w.detach().reshape(C, C) for w in ws]
    return torch.cat(parts, 0, out=out) if out is not None else torch.cat(parts, 0)


def _cat_weights(ws, C, cache):
    """The four (C, C, 1, 1) projection weights of a map as one (4C, C) matrix.  Inference (``cache``): kept across calls
    (``_wc_get``) — six concatenation launches less per forward (37 us of config C2's 1.28 ms)."""
    if not cache:
        return torch.cat([w.reshape(C, C) for w in ws], 0)
    return _wc_get("cat", tuple(ws), _cat_build)


def head_weight16(weight, num_keypoints):
    """The 1x1 head's filters zero-padded to 16 output channels (later kernels stay float4-aligned).  Under no_grad the padded copy is
    kept across calls like the concatenated projection weights (one pad launch less per inference forward)."""
    def build(ws, out):
        w = ws[0].detach()
        if out is None:
            return torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, 16 - w.shape[0]))
        out[:w.shape[0]].copy_(w)
        return out
    if torch.is_grad_enabled():                                 # training: an ordinary differentiable pad
        return torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, 16 - num_keypoints))
    assert weight.shape[0] == num_keypoints
    return _wc_get("head16", (weight,), build)


@_math_scoped
class MSCSALevelFn(torch.autograd.Function):
    """One level of the multi-scale cross/self attention (models/layers.py:150-163 of the reference): eight 1x1
    projections of the two maps and the four attentions they feed, as one autograd node.

        ra . [phi_cross_hori | theta_cross_hori | phi_self_hori | theta_self_hori] -> Ya (B, N, 4C)
        re . [phi_cross_vert | theta_cross_vert | phi_self_vert | theta_self_vert] -> Ye
        out1 = attn(K=Ya[0], Q=Ye[1], V=ra) + ra      out2 = attn(K=Ya[2], Q=Ya[3], V=ra)
        out3 = attn(K=Ye[0], Q=Ya[1], V=re) + re      out4 = attn(K=Ye[2], Q=Ye[3], V=re)

    The four projections of a map are ONE GEMM; the backward writes dK / dQ into column blocks of dYa / dYe and
    accumulates both dV of a map in place, so each map's gradient is one GEMM (K = 4C) with dV as its residual term and
    each map's four weight gradients are one split-K GEMM — no gradient-accumulation kernels at all.
    Attention core: the fused kernels where they exist for (N, C) — the projection epilogue then stores the bf16
    operands those kernels read, no fp32 projections and no casts — else GEMM + row-softmax on the fp32 projections
    (the level-0 maps, C = 256), with the same strided operands / outputs.
    Weights: the eight (C, C, 1, 1) parameters in the order of the two lists above.
    cat_bf16 (fused attention kernels only): return ONE bf16 tensor (B,1,H,W,4C) = cat(out1..out4) — the kernels write
    the bf16 copy the decoder concatenates next straight from their accumulators, and the backward reads the four
    column blocks of the incoming (possibly strided: a slice of the concatenation's gradient) bf16 gradient in place: no
    cast or copy kernels on either side.  Otherwise: the four fp32 outputs."""

    #            K source/slot, Q source/slot, V map (0: ra, 1: re), residual
    SPEC = ((0, 0, 1, 1, 0, True), (0, 2, 0, 3, 0, False), (1, 0, 0, 1, 1, True), (1, 2, 1, 3, 1, False))

    @staticmethod
    def forward(ctx, ra, re, cat_bf16, *weights):
        assert len(weights) == 8
        ra, re = _c(ra), _c(re)
        B, _, H, W, C = ra.shape
        N = H * W
        L = rt.lib()
        dev = ra.device
        flash = USE_FLASH and bool(L.hupr_attn_flash_supported(N, C))
        ydt = torch.bfloat16 if flash else torch.float32
        esz = 2 if flash else 4
        maps = (ra, re)
        infer = bool(int(cat_bf16) & 2)                  # bit 1: the caller runs under no_grad (grad mode is always off in here)
        cat_bf16 = bool(int(cat_bf16) & 1)
        # QS: the query projections leave the GEMM as log2(e) Q (rounded to bf16 once, like every projection), the attention kernels
        # take the exponent of 2 straight from the matrix pipe (csrc/attention_bf16.hip, kDeferBits)
        qscaled = flash and QS_ATTN                  # (not `qs`: the SPEC loops below bind that name to the query source map)
        pa, pe = _proj_cat(weights[:4], C), _proj_cat(weights[4:], C)
        Wc = (pa[0], pe[0])                              # plain: the backward GEMMs
        Wf = (pa[1], pe[1]) if qscaled else Wc                # what the forward projections multiply by
        Y = (torch.empty((B, N, 4 * C), dtype=ydt, device=dev), torch.empty((B, N, 4 * C), dtype=ydt, device=dev))
        for x, wc, y in zip(maps, Wf, Y):          # a 1x1 kernel's packed layout IS the parameter layout (Co, Ci)
            if flash and PROJ_STREAM and L.hupr_mscsa_proj_supported(B * N, C):
                # the four projections of the map as one HBM-bound streaming product (csrc/projection.hip)
                rt.check(L.hupr_mscsa_proj_fwd_bf16(rt.ptr(x), rt.ptr(wc), rt.ptr(y), B * N, C, rt.stream()))
                continue
            rt.check(L.hupr_conv_fwd_bf16_mixed(rt.ptr(x), 0, rt.ptr(wc), None, rt.ptr(y), 1 if flash else 0, B, 1, H, W, C, C,
                                                1, H, W, 4 * C, 4 * C, 1, 1, 1, 0, 0, 0, rt.stream()))
        outs = [torch.empty((B, 1, H, W, C), dtype=torch.float32, device=dev) for _ in range(4)]
        cat_bf16 = bool(cat_bf16) and flash
        cat, ld16 = None, 4 * C
        if cat_bf16:
            cat = _level_cat_out.pop("out", None)      # output placement: a channel slice of the decoder stage's input buffer
            if cat is not None:
                assert tuple(cat.shape) == (B, 1, H, W, 4 * C) and cat.dtype == torch.bfloat16 and _ld_view_ok(cat, 4 * C)
                ld16 = cat.stride(3)
            else:
                cat = torch.empty((B, 1, H, W, 4 * C), dtype=torch.bfloat16, device=dev)
        _level_cat_out.clear()
        if flash:
            vb = (_cast(ra, torch.bfloat16), _cast(re, torch.bfloat16))
            aux = [torch.empty((B, N), dtype=torch.float32, device=dev) for _ in range(4)]          # log-sum-exp per query
        else:
            vb = maps
            aux = [torch.empty((B, N, N), dtype=torch.float32, device=dev) for _ in range(4)]       # P[query][key]
        # single-sample inference (config C2): the four attentions of the level as ONE split launch + ONE merge launch instead of eight
        split_ws = L.hupr_attn_fwd_split_ws_bytes(B, N, C) if flash else 0       # (> 0: the key-split form applies to this batch)
        split_bytes = split_ws if (infer and ATTN_BATCH) else 0
        # training batches, levels 2 and 3 (C = 128 / 256: one attention's grid is 256 / 64 workgroups): the four as ONE launch of the
        # one-pass kernel (not where the single-sample split form applies: its shares round differently)
        level_batch = flash and ATTN_LEVEL_BATCH and not split_ws and C != 64
        if split_bytes or level_batch:
            items = (rt.AttnItem * 4)()
            for i, ((ks, kslot, qs, qslot, vs, residual), out, a) in enumerate(zip(MSCSALevelFn.SPEC, outs, aux)):
                items[i].K, items[i].Q = Y[ks].data_ptr() + kslot * C * esz, Y[qs].data_ptr() + qslot * C * esz
                items[i].V, items[i].Vres = rt.ptr(vb[vs]), (rt.ptr(maps[vs]) if residual else None)
                items[i].out, items[i].lse = rt.ptr(out), rt.ptr(a)
                items[i].out16 = (cat.data_ptr() + i * C * 2) if cat_bf16 else None
            ws = workspace(4 * split_bytes, dev) if split_bytes else None
            fwd_batch = L.hupr_attn_fwd_bf16in_ld_ws_batch_qs if qscaled else L.hupr_attn_fwd_bf16in_ld_ws_batch
            rt.check(fwd_batch(items, 4, 4 * C, 4 * C, ld16, B, N, C, rt.ptr(ws) if ws is not None else None,
                               ws.numel() if ws is not None else 0, rt.stream()))
        for i, ((ks, kslot, qs, qslot, vs, residual), out, a) in enumerate(zip(MSCSALevelFn.SPEC, outs, aux)):
            if split_bytes or level_batch:
                break
            kp, qp = Y[ks].data_ptr() + kslot * C * esz, Y[qs].data_ptr() + qslot * C * esz
            if flash:
                ws = _attn_ws(B, N, C, dev)
                fwd = L.hupr_attn_fwd_bf16in_ld_ws_qs if qscaled else L.hupr_attn_fwd_bf16in_ld_ws
                ev = ATTN_PROBE("fwd", B, N, C) if ATTN_PROBE is not None else None
                if ev is not None:
                    ev[0].record()
                rt.check(fwd(kp, 4 * C, qp, 4 * C, rt.ptr(vb[vs]), rt.ptr(maps[vs]) if residual else None,
                             rt.ptr(out), rt.ptr(a), cat.data_ptr() + i * C * 2 if cat_bf16 else None,
                             ld16, B, N, C, rt.ptr(ws) if ws is not None else None,
                             ws.numel() if ws is not None else 0, rt.stream()))
                if ev is not None:
                    ev[1].record()
            else:
                # P[q][j] = Q[q] . K[j], softmax over the keys j (row softmax), out = P V (+ V)
                rt.check(L.hupr_gemm_bf16(0, 1, qp, kp, rt.ptr(a), N, N, C, 4 * C, 4 * C, N, B, N * 4 * C, N * 4 * C, N * N,
                                          None, 0, 0, 0, rt.stream()))
                rt.check(L.hupr_softmax_rows_f32(rt.ptr(a), B * N, N, rt.stream()))
                v = maps[vs]
                rt.check(L.hupr_gemm_bf16(0, 0, rt.ptr(a), rt.ptr(v), rt.ptr(out), N, C, N, N, C, C, B, N * N, N * C, N * C,
                                          rt.ptr(v) if residual else None, C, N * C if residual else 0, 0, rt.stream()))
        ctx.save_for_backward(ra, re, Wc[0], Wc[1], Y[0], Y[1], vb[0], vb[1], *outs, *aux)
        ctx.weights = weights
        ctx.flash, ctx.cat_bf16, ctx.qscaled = flash, cat_bf16, qscaled
        if cat_bf16:
            return (cat,)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        t = ctx.saved_tensors
        maps, Wc, Y, vb, outs, aux = t[0:2], t[2:4], t[4:6], t[6:8], t[8:12], t[12:16]
        weights = ctx.weights
        flash = ctx.flash
        esz = 2 if flash else 4
        B, _, H, W, C = maps[0].shape
        N = H * W
        L = rt.lib()
        dev = maps[0].device
        f32 = torch.float32
        dY = (torch.empty((B, N, 4 * C), dtype=f32, device=dev), torch.empty((B, N, 4 * C), dtype=f32, device=dev))
        dV = (torch.empty((B, N, C), dtype=f32, device=dev), torch.empty((B, N, C), dtype=f32, device=dev))
        scr = torch.empty((B, N), dtype=f32, device=dev) if flash else torch.empty((B, N, N), dtype=f32, device=dev)
        if ctx.cat_bf16:
            # one bf16 gradient (B,1,H,W,4C), typically a column slice of the decoder input's gradient: read in place
            dcat = douts[0]
            ld = dcat.stride(3)
            if not (dcat.dtype == torch.bfloat16 and dcat.stride(4) == 1 and dcat.stride(2) == W * ld and ld % 8 == 0
                    and (B == 1 or dcat.stride(0) == N * ld) and dcat.data_ptr() % 16 == 0):
                dcat = _cast(dcat, torch.bfloat16)
                ld = 4 * C
            douts = [None] * 4
        # SPEC order: the residual attention of a map writes its dV, the other one adds to it
        level_batch = flash and ctx.cat_bf16 and ATTN_LEVEL_BATCH
        if level_batch:
            # row sums and dQ of the four attentions in one launch each; dK / dV: levels 2 and 3 in two launches of two (the residual
            # attentions of the two maps, then the two that add onto their dV) — 4 launches instead of 12, grids four / two times as
            # large; level 1 one launch per attention (12 -> 6 launches)
            scr4 = torch.empty((4, B, N), dtype=f32, device=dev)
            items = (rt.AttnBwdItem * 4)()
            for i, ((ks, kslot, qs, qslot, vs, residual), out, a) in enumerate(zip(MSCSALevelFn.SPEC, outs, aux)):
                it = items[i]
                it.K, it.Q = Y[ks].data_ptr() + kslot * C * esz, Y[qs].data_ptr() + qslot * C * esz
                it.V, it.dO = rt.ptr(vb[vs]), dcat.data_ptr() + i * C * 2
                it.V32, it.out, it.lse = rt.ptr(maps[vs]), rt.ptr(out), rt.ptr(a)
                it.dK, it.dQ = dY[ks].data_ptr() + kslot * C * 4, dY[qs].data_ptr() + qslot * C * 4
                it.dV, it.Dq = rt.ptr(dV[vs]), scr4.data_ptr() + i * B * N * 4
                it.residual, it.accumulate = (1, 0) if residual else (0, 1)
            bwd_batch = L.hupr_attn_bwd_bf16in_ld_batch_qs if ctx.qscaled else L.hupr_attn_bwd_bf16in_ld_batch
            ev = ATTN_PROBE("bwd_level", B, N, C) if ATTN_PROBE is not None else None
            if ev is not None:
                ev[0].record()
            rt.check(bwd_batch(items, 4, 4 * C, 4 * C, ld, 4 * C, 4 * C, B, N, C, rt.stream()))
            if ev is not None:
                ev[1].record()
        for i, ((ks, kslot, qs, qslot, vs, residual), out, a, dout) in enumerate(zip(MSCSALevelFn.SPEC, outs, aux, douts)):
            if level_batch:
                break
            kp, qp = Y[ks].data_ptr() + kslot * C * esz, Y[qs].data_ptr() + qslot * C * esz
            dkp, dqp = dY[ks].data_ptr() + kslot * C * 4, dY[qs].data_ptr() + qslot * C * 4
            if flash:
                if ctx.cat_bf16:
                    gp, ldg, g32 = dcat.data_ptr() + i * C * 2, ld, None
                else:
                    dout = _c(dout)
                    gb = _cast(dout, torch.bfloat16)
                    gp, ldg, g32 = rt.ptr(gb), C, rt.ptr(dout)
                bwd = L.hupr_attn_bwd_bf16in_ld_qs if ctx.qscaled else L.hupr_attn_bwd_bf16in_ld
                ev = ATTN_PROBE("bwd", B, N, C) if ATTN_PROBE is not None else None
                if ev is not None:
                    ev[0].record()
                rt.check(bwd(kp, 4 * C, qp, 4 * C, rt.ptr(vb[vs]), gp, ldg, rt.ptr(maps[vs]),
                             rt.ptr(out), g32, rt.ptr(a), dkp, 4 * C, dqp, 4 * C, rt.ptr(dV[vs]),
                             rt.ptr(scr), B, N, C, 1 if residual else 0, 0 if residual else 1,
                             rt.stream()))
                if ev is not None:
                    ev[1].record()
                continue
            dout = _c(dout)
            v, P, g = maps[vs], a, rt.ptr(dout)
            # dV[j] = sum_q P[q][j] dout[q]  (+ dout: residual form; else added onto the dV already there)
            rt.check(L.hupr_gemm_bf16(1, 0, rt.ptr(P), g, rt.ptr(dV[vs]), N, C, N, N, C, C, B, N * N, N * C, N * C,
                                      g if residual else None, C, N * C if residual else 0, 0 if residual else 1, rt.stream()))
            # dP[q][j] = dout[q] . V[j]; dS = softmax backward in place
            rt.check(L.hupr_gemm_bf16(0, 1, g, rt.ptr(v), rt.ptr(scr), N, N, C, C, C, N, B, N * C, N * C, N * N, None, 0, 0, 0,
                                      rt.stream()))
            rt.check(L.hupr_softmax_rows_bwd_f32(rt.ptr(P), rt.ptr(scr), B * N, N, rt.stream()))
            # dQ[q] = sum_j dS[q][j] K[j];  dK[j] = sum_q dS[q][j] Q[q]   (into their column blocks)
            rt.check(L.hupr_gemm_bf16(0, 0, rt.ptr(scr), kp, dqp, N, C, N, N, 4 * C, 4 * C, B, N * N, N * 4 * C, N * 4 * C,
                                      None, 0, 0, 0, rt.stream()))
            rt.check(L.hupr_gemm_bf16(1, 0, rt.ptr(scr), qp, dkp, N, C, N, N, 4 * C, 4 * C, B, N * N, N * 4 * C, N * 4 * C,
                                      None, 0, 0, 0, rt.stream()))
        grads = [None, None]
        for i in range(2):
            if ctx.needs_input_grad[i]:
                dx = torch.empty_like(maps[i])
                if PROJ_STREAM and L.hupr_mscsa_proj_dgrad_supported(B * N, C):
                    rt.check(L.hupr_mscsa_proj_dgrad_f32(rt.ptr(dY[i]), rt.ptr(Wc[i]), rt.ptr(dV[i]), rt.ptr(dx), B * N, C, rt.stream()))
                else:
                    rt.check(L.hupr_gemm_bf16(0, 0, rt.ptr(dY[i]), rt.ptr(Wc[i]), rt.ptr(dx), B * N, C, 4 * C, 4 * C, C, C, 1, 0, 0,
                                              0, rt.ptr(dV[i]), C, 0, 0, rt.stream()))
                grads[i] = dx
        wgrads = [None] * 8
        dsts, srcs, landed = [], [], []
        for i in range(2):
            if not any(ctx.needs_input_grad[3 + 4 * i + j] for j in range(4)):
                continue
            ws4 = weights[4 * i:4 * i + 4]
            got = [_pgrad(w) if ctx.needs_input_grad[3 + 4 * i + j] else (None, False) for j, w in enumerate(ws4)]
            # the four gradient slots adjacent in their flat bucket, in this order (HuPRNet.gradient_groups -> GradientBuckets): the
            # GEMM writes them in place — no (4C, C) temporary, no copy launch
            inplace = all(g is not None and d for g, d in got) and all(
                got[j][0].is_contiguous() and got[j][0].data_ptr() == got[0][0].data_ptr() + j * C * C * 4 for j in range(4))
            dWc = None if inplace else torch.empty((4 * C, C), dtype=f32, device=dev)
            ws = workspace(L.hupr_conv_wgrad_ws_bytes(B, 1, H, W, C, 4 * C, 1, 1, 1), dev)
            rt.check(L.hupr_conv_wgrad_bf16(rt.ptr(maps[i]), rt.ptr(dY[i]), got[0][0].data_ptr() if inplace else rt.ptr(dWc), B, 1, H, W, C, C,
                                            1, H, W, 4 * C, 4 * C, 1, 1, 1, 0, 0, 0, rt.ptr(ws), ws.numel(), rt.stream()))
            for j in range(4):
                w = ws4[j]
                if ctx.needs_input_grad[3 + 4 * i + j]:
                    g, direct = got[j]
                    if not inplace:
                        dsts.append(g)
                        srcs.append(dWc[j * C:(j + 1) * C].view_as(w))
                    landed.append((4 * i + j, w, g, direct))
        if dsts:
            torch._foreach_copy_(dsts, srcs)        # the eight row blocks of the two fused gradients in one launch
        for slot, w, g, direct in landed:
            wgrads[slot] = _pret(w, g, direct)
        return (grads[0], grads[1], None) + tuple(wgrads)


# The PRGCN head stays on the fp32 matrix pipe in bf16 runs.  It is 0.07 % of the model's flops (3 x 1024 x 1024 x 14 per
# sample) but its 1024-term dot products of un-normalised logits are where bf16 operand rounding hurt most: on trained
# (peaky) weights the decoded head went from 94.4 % to >= 99 % arg-max agreement with the fp32 path at B = 32
# (tests/test_trained_gpu.py), for a few tens of microseconds per step.
GCN_MATH = "f32"
GCN_PRODUCTS = True        # test aid: False = the generic fp32 engine instead of csrc/gcn_products.hip


def _gcn_products_ok(x, weight):
    """The dedicated PRGCN product kernels apply: fp32 pipe (the default of the head in every mode), 16-wide key-point slots."""
    return (GCN_PRODUCTS and (GCN_MATH == "f32" or _st.math == "f32") and x.dtype == torch.float32 and x.shape[2] == 16
            and x.shape[1] % 64 == 0 and tuple(weight.shape) == (x.shape[1], x.shape[1]))


@_math_scoped
class GCNLayerFn(torch.autograd.Function):
    """y = act( (W x) A + bias ) == W (x A) + bias  (models/gcn_networks.py:23-29); x, y: (B, F, ld=16)."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj, relu):
        x = _c(x)
        B, F, ld = x.shape
        K = bias.shape[1]
        L = rt.lib()
        if B == 1 and F % 1024 == 0:
            # single-sample inference: F / 128 = 8 workgroups would each walk K = F alone (37 us of latency for 34 MFLOP);
            # eight K slices side by side as a batched GEMM; the adjacency kernel sums them
            S = 8
            t = gemm(0, 0, weight, x, F, ld, F // S, F, ld, S, F // S, (F // S) * ld, math=GCN_MATH)
        elif _gcn_products_ok(x, weight):
            S = 1
            t = torch.empty_like(x)
            rt.check(L.hupr_gcn_wx_f32(rt.ptr(_c(weight)), rt.ptr(x), rt.ptr(t), B, F, ld, 0, rt.stream()))
        else:
            S = 1
            t = gemm(0, 0, weight, x, F, ld, F, F, ld, B, 0, F * ld, math=GCN_MATH)
        y = torch.empty_like(x)
        if S > 1:
            rt.check(L.hupr_gcn_adj_fwd_sliced_f32(rt.ptr(t), S, rt.ptr(adj), rt.ptr(_c(bias)), rt.ptr(y), B, F, K, ld, 1 if relu else 0,
                                                   rt.stream()))
        else:
            rt.check(L.hupr_gcn_adj_fwd_f32(rt.ptr(t), rt.ptr(adj), rt.ptr(_c(bias)), rt.ptr(y), B, F, K, ld, 1 if relu else 0,
                                            rt.stream()))
        ctx.save_for_backward(x, weight, y, adj)
        ctx.bias_ref = bias
        ctx.relu, ctx.K = relu, K
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y, adj = ctx.saved_tensors
        B, F, ld = x.shape
        K = ctx.K
        L = rt.lib()
        dt = torch.empty_like(x)
        gm = torch.empty_like(x)
        # weight / bias gradients go straight into the flat-bucket views when a gradient sink is installed (round 5: the six
        # AccumulateGrad add kernels of the PRGCN per step are gone)
        bias = ctx.bias_ref
        dbias, db_direct = _pgrad(bias) if tuple(bias.shape) == (F, K) and bias.is_contiguous() else (torch.empty((F, K), dtype=torch.float32, device=x.device), False)
        rt.check(L.hupr_gcn_adj_bwd_f32(rt.ptr(_c(dy)), rt.ptr(y), rt.ptr(adj), rt.ptr(dt), rt.ptr(gm), rt.ptr(dbias), B, F, K,
                                        ld, 1 if ctx.relu else 0, rt.stream()))
        if _gcn_products_ok(x, weight) and B > 1:
            # the batch folded into the matrix pipe's column axis (dx) / reduction axis (dW) in place: csrc/gcn_products.hip
            dx = torch.empty_like(x)
            rt.check(L.hupr_gcn_wx_f32(rt.ptr(_c(weight)), rt.ptr(dt), rt.ptr(dx), B, F, ld, 1, rt.stream()))
            dw, dw_direct = _pgrad(weight)
            rt.check(L.hupr_gcn_dw_f32(rt.ptr(dt), rt.ptr(x), rt.ptr(dw), B, F, ld, rt.stream()))
            return dx, _pret(weight, dw, dw_direct), _pret(bias, dbias, db_direct), None, None
        dx = gemm(1, 0, weight, dt, F, ld, F, F, ld, B, 0, F * ld, math=GCN_MATH)              # W^T dt
        # dW[f][g] = sum_{b,k} dt[b][f][k] x[b][g][k]: fold the batch into the reduction axis
        dt2 = dt.permute(1, 0, 2).reshape(F, B * ld)
        x2 = x.permute(1, 0, 2).reshape(F, B * ld)
        dw = gemm(0, 1, _c(dt2), _c(x2), F, F, B * ld, B * ld, B * ld, 1, 0, 0, math=GCN_MATH)[0]
        return dx, dw, _pret(bias, dbias, db_direct), None, None


_head_cache = {}       # address of the head weight -> (zero-padded (16, 32, 1, 1) copy, pack-table entry)


def _head_w16_cached(weight):
    """The 14 head filters zero-padded to 16 as a PACK-TABLE entry (kind 2: a plain copy into rows 0..K-1 of a persistent buffer whose
    other rows stay zero), refreshed by the one table launch after an optimiser step — rounds 1-4 padded with ATen in every
    training forward (a fill, a copy, and a slice + accumulate in the backward)."""
    K = weight.shape[0]
    capturing = torch.cuda.is_current_stream_capturing()
    ok = PACK_CACHE and weight.is_leaf and weight.requires_grad and weight.is_contiguous() and tuple(weight.shape[1:]) == (32, 1, 1)
    ent = _head_cache.get(weight.data_ptr()) if ok else None
    if ent is not None and ent[1].wref() is not weight:
        ent = None
    fresh = ent is not None and ent[1].stamp == (PACK_EPOCH, weight._version)
    if not ok or (capturing and (torch.is_grad_enabled() or not fresh)):
        return torch.nn.functional.pad(weight.detach(), (0, 0, 0, 0, 0, 0, 0, 16 - K))
    if ent is None:
        global _pack_table
        for k in [k for k, v in _head_cache.items() if v[1].wref() is None]:
            del _head_cache[k]
        buf = torch.zeros((16,) + tuple(weight.shape[1:]), dtype=torch.float32, device=weight.device)
        e = _PackEntry()
        e.wref, e.ptr, e.kind, e.shape = weakref.ref(weight), weight.data_ptr(), 2, tuple(weight.shape)
        e.wp = (buf[:K], buf[:K])
        e.stamp = None
        e.group = PACK_EPOCH
        _pack_entries[(weight.data_ptr(), 2)] = e
        ent = _head_cache[weight.data_ptr()] = (buf, e)
        _pack_table = None
        fresh = False
        created = True
    else:
        created = False
    known = _touch(ent[1])
    if not fresh:
        _pack_refresh_all(weight.device, full=None if (known or created) else ent[1].group)
    return ent[0]


@_math_scoped
class Head1x1Fn(torch.autograd.Function):
    """The 1x1 key-point head (reference models/layers.py:94) as plain fp32 FMAs: x (B,1,H,W,32) fp32, weight (K,32,1,1) the parameter,
    w16 (16,32,1,1) its zero-padded copy (``_head_w16_cached``) -> (B,1,H,W,16).  What the "head" precision region of a bf16 run uses
    (PRECISION["head"] = "f32"): the generic fp32 implicit-GEMM path costs 0.26 ms per training step on this 58-MFLOP product, these
    kernels ~0.03.  The weight gradient is written for the K real filters only, straight into the gradient sink."""

    @staticmethod
    def forward(ctx, x, weight, w16):
        x, w16 = _c(x), _c(w16)
        B, D, H, W, Ci = _vox(x)
        assert Ci == 32 and tuple(w16.shape) == (16, 32, 1, 1) and x.dtype == torch.float32
        y = torch.empty((B, D, H, W, 16), dtype=torch.float32, device=x.device)
        rt.check(rt.lib().hupr_head1x1_fwd_f32(rt.ptr(x), rt.ptr(w16), rt.ptr(y), B * D * H * W, rt.stream()))
        ctx.save_for_backward(x, w16, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16, weight = ctx.saved_tensors
        dy = _c(dy)
        M = x.numel() // 32
        L = rt.lib()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, direct = _pgrad(weight) if ctx.needs_input_grad[1] else (None, False)
        ws = workspace(L.hupr_head1x1_ws_bytes(), x.device)
        rt.check(L.hupr_head1x1_bwd_rows_f32(rt.ptr(x), rt.ptr(w16), rt.ptr(dy), rt.ptr(dx) if dx is not None else None,
                                             rt.ptr(dw) if dw is not None else None, weight.shape[0], M, rt.ptr(ws), ws.numel(), rt.stream()))
        return dx, _pret(weight, dw, direct) if dw is not None else None, None


def head_conv(x, weight, num_keypoints):
    """1x1 head: the dedicated fp32 kernels inside a bf16 run whose "head" region is switched to fp32 (32 -> 16 channels),
    the generic convolution otherwise (the pure fp32 parity path keeps the arithmetic its golden fixtures were pinned with)."""
    if _st.region_switched and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 32 and tuple(weight.shape) == (num_keypoints, 32, 1, 1):
        return Head1x1Fn.apply(x, weight, _head_w16_cached(weight))
    return conv(x, head_weight16(weight, num_keypoints), None, None, (0, 0, 0))


@_math_scoped
class SigmoidHeadFn(torch.autograd.Function):
    """channels-last logits (B, HW, ld) -> probabilities (B, K, HW) in NCHW order."""

    @staticmethod
    def forward(ctx, x, K):
        x = _c(x)
        B, HW, ld = x.shape
        y = torch.empty((B, K, HW), dtype=torch.float32, device=x.device)
        rt.check(rt.lib().hupr_sigmoid_to_nchw_f32(rt.ptr(x), rt.ptr(y), B, HW, K, ld, rt.stream()))
        ctx.save_for_backward(y)
        ctx.ld = ld
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        B, K, HW = y.shape
        dx = torch.empty((B, HW, ctx.ld), dtype=torch.float32, device=y.device)
        rt.check(rt.lib().hupr_sigmoid_to_nchw_bwd_f32(rt.ptr(_c(dy)), rt.ptr(y), rt.ptr(dx), B, HW, K, ctx.ld, rt.stream()))
        return dx, None


@_math_scoped
class BCEFn(torch.autograd.Function):
    """nn.BCELoss(reduction='mean') on probabilities."""

    @staticmethod
    def forward(ctx, p, t):
        p, t = _c(p), _c(t)
        L = rt.lib()
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        ws = workspace(L.hupr_bce_ws_bytes(), p.device)
        rt.check(L.hupr_bce_fwd_f32(rt.ptr(p), rt.ptr(t), p.numel(), rt.ptr(loss), rt.ptr(ws), ws.numel(), rt.stream()))
        ctx.save_for_backward(p, t)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        dp = torch.empty_like(p)
        g = _c(g.reshape(1).to(torch.float32))
        rt.check(rt.lib().hupr_bce_bwd_f32(rt.ptr(p), rt.ptr(t), rt.ptr(g), rt.ptr(dp), p.numel(), rt.stream()))
        return dp, None


@_math_scoped
class PairBCEFn(torch.autograd.Function):
    """(alpha * BCE(p1, t) + beta * BCE(p2, t), BCE(p2, t)) — the two losses of a step and their weighted sum (reference
    misc/losses.py:24-33) as two launches forward and one backward instead of four, two and three torch-native ones; the same floats
    as two BCEFn nodes combined by torch (products and sum rounded separately)."""

    @staticmethod
    def forward(ctx, p1, p2, t, alpha, beta):
        p1, p2, t = _c(p1), _c(p2), _c(t)
        assert p1.shape == p2.shape == t.shape and p1.dtype == p2.dtype == t.dtype == torch.float32
        L = rt.lib()
        out = torch.empty(3, dtype=torch.float32, device=p1.device)
        ws = workspace(2 * L.hupr_bce_ws_bytes(), p1.device)
        rt.check(L.hupr_bce_pair_fwd_f32(rt.ptr(p1), rt.ptr(p2), rt.ptr(t), p1.numel(), float(alpha), float(beta), rt.ptr(out), rt.ptr(ws),
                                         ws.numel(), rt.stream()))
        ctx.save_for_backward(p1, p2, t)
        ctx.ab = (float(alpha), float(beta))
        ctx.set_materialize_grads(False)
        return out[0], out[2]

    @staticmethod
    def backward(ctx, g, g2):
        p1, p2, t = ctx.saved_tensors
        if g is None:
            g = torch.zeros(1, dtype=torch.float32, device=p1.device)
        dp1, dp2 = torch.empty_like(p1), torch.empty_like(p2)
        g = _c(g.reshape(1).to(torch.float32))
        g2 = _c(g2.reshape(1).to(torch.float32)) if g2 is not None else None
        rt.check(rt.lib().hupr_bce_pair_bwd_f32(rt.ptr(p1), rt.ptr(p2), rt.ptr(t), rt.ptr(g), rt.ptr(g2) if g2 is not None else None,
                                                ctx.ab[0], ctx.ab[1], rt.ptr(dp1), rt.ptr(dp2), p1.numel(), rt.stream()))
        return dp1, dp2, None, None, None


_PATCH_CACHE = {}


def gaussian_targets(joints, hsize=64, isize=256, sigma=2):
    """joints (B,K,2) int64 GPU tensor -> (B,K,H,W) fp32 targets (misc/utils.py:6-66)."""
    joints = _c(joints.to(torch.int64))
    B, K, _ = joints.shape
    rad = 3 * sigma
    key = (sigma, joints.device)
    if key not in _PATCH_CACHE:
        ax = np.arange(2 * rad + 1, dtype=np.float32)
        g = np.exp(-((ax[None, :] - rad) ** 2 + (ax[:, None] - rad) ** 2) / (2 * sigma ** 2))
        _PATCH_CACHE[key] = torch.from_numpy(g.astype(np.float32)).to(joints.device)
    t = torch.empty((B, K, hsize, hsize), dtype=torch.float32, device=joints.device)
    rt.check(rt.lib().hupr_gaussian_targets_f32(rt.ptr(joints), rt.ptr(_PATCH_CACHE[key]), rt.ptr(t), B * K, hsize, rad,
                                               float(isize) / float(hsize), rt.stream()))
    return t


def argmax_rows(p):
    """p (rows, n) GPU -> (idx int32 (rows,), maxval (rows,)) with first-max tie-break."""
    p = _c(p)
    rows, n = p.shape
    idx = torch.empty(rows, dtype=torch.int32, device=p.device)
    mx = torch.empty(rows, dtype=torch.float32, device=p.device)
    rt.check(rt.lib().hupr_argmax_rows_f32(rt.ptr(p), rows, n, rt.ptr(idx), rt.ptr(mx), rt.stream()))
    return idx, mx
