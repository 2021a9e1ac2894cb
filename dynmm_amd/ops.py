"""torch.autograd.Function shells over the C ABI (include/dynmm_hip.h).

PyTorch supplies device memory (caching allocator), the current HIP stream and the autograd tape;
every FLOP and every byte moved on the hot path is done by the kernels in libdynmm_hip.so.  All
functions require contiguous fp32 tensors on a HIP device and raise otherwise — there is no eager
fallback.
"""
import contextlib
import ctypes as C

import torch
from torch.autograd import Function

from . import lib as L

ACT = L.ACT


def _lib():
    return L.load()


def _stream():
    if not torch.cuda.is_available():
        raise L.DynmmHipError('no HIP device is available: the DynMM hot path runs only on its HIP kernels '
                              '(there is no CPU / eager-PyTorch fallback)')
    h = torch.cuda.current_stream().cuda_stream
    _CENSUS.add(h)
    return h


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, name='tensor'):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.DynmmHipError(f'{name} is on {t.device}: the DynMM HIP path needs a HIP (cuda) device; '
                              'there is no CPU fallback')
    if t.dtype != torch.float32:
        raise L.DynmmHipError(f'{name} must be float32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


# Winograd convolutions on the fp32 matrix cores: the stride-1 1x3 / 3x1 convolutions by 1-D F(2,3) (csrc/conv_wino.hip: 2/3 of the
# direct matrix work), the 3x3 ones by 2-D F(2x2,3x3) (csrc/conv_wino2d.hip: 4/9) — forward (training and inference) and input
# gradients of every convolution the kernels' geometry rules admit; the weight gradients by the three-tap form of
# csrc/conv_wgrad_v6.hip / conv_wgrad_wino_vt.hip.  Everything else (and a pass switched off here) runs on the operand-ring / tile
# kernels, which therefore stay in the library either way.  These are MODULE ATTRIBUTES — caller options used by the parity tests
# (tests/test_hip_ops.py: test_conv2d_winograd runs every mode) — not environment switches:
#   WINO        'all' (default) | 'dgrad' (input gradients only: direct training forward) | 'fwd' | '0'.  The training forward in
#               the Winograd form is closer to the fp64 result than the direct kernels on every shape tried (1.0e-6 vs 1.35e-6 on
#               the decoder module's outputs); inference always takes it when WINO != '0'.
#   WINO_DGRAD  '43h' (default): F(4,3) (csrc/conv_wino43.hip: 1/2 of the work, 1e-6 .. 4e-6 from fp64) for the input gradients of
#               the 1x3 filters, F(2,3) for the 3x1 ones | '23': F(2,3) for both.  (A vertical F(4,3) form existed in round 4 — six
#               input rows per four output rows through b32 reads, 72.37 ms against 71.63 — and was removed in round 5.  3x3
#               filters take the 2-D form under either value.)
#   CONV_BN_STATS  BatchNorm batch statistics from the epilogue of the convolution that feeds the BatchNorm (conv_wino.hip STATS,
#               conv_wino2d.hip, and the two 7x7 stems: conv_small.hip conv_stem_fwd_kernel<CI, STATS>) instead of a bn_stats launch
#               + a pass over the conv output.
WINO = 'all'
WINO_DGRAD = '43h'
CONV_BN_STATS = True
_WINO_OK = {}
_WINO43_OK = {}


def _wino43(g):
    """input gradient of this convolution on the F(4,3) kernel?  (only consulted where _wino(g, True) holds)"""
    if WINO_DGRAD == '23' or g.KW != 3:
        return False
    # (round 5, alternating runs: F(4,3) only where its quad tiles fill the chip — C <= 128 / C <= 256 — 63.94 / 64.07 ms against
    # 63.75 with F(4,3) on every horizontal launch: where it is slower in isolation (C = 512: 124 against 111 us) the half of the
    # matrix pipe it leaves goes to the weight-gradient streams)
    key = (g.N, g.Ci, g.H, g.W, g.Co, g.KH, g.KW, g.SH, g.SW, g.PH, g.PW, g.c_split)
    ok = _WINO43_OK.get(key)
    if ok is None:
        ok = _WINO43_OK[key] = bool(_lib().dynmm_conv2d_wino43_supported(C.byref(g)))
    return ok


_WINO2D_OK = {}


def _wino2d(g, dgrad, x2=None, infer=False):
    """does this pass of this (3x3) convolution run on the 2-D Winograd kernel (csrc/conv_wino2d.hip: 4/9 of the direct matrix work)?
    Same WINO switch as _wino (round 4 ran 3x3 filters on the 1-D forms with the vertical taps looped: 2/3 and 1/2; removed)."""
    if WINO == '0' or x2 is not None or g.KH != 3 or g.KW != 3:
        return False
    if not infer and ((WINO == 'dgrad' and not dgrad) or (WINO == 'fwd' and dgrad)):
        return False
    key = (g.N, g.Ci, g.H, g.W, g.Co, g.SH, g.SW, g.PH, g.PW, g.c_split, bool(dgrad))
    ok = _WINO2D_OK.get(key)
    if ok is None:
        ok = _WINO2D_OK[key] = bool(_lib().dynmm_conv2d_wino2d_supported(C.byref(g), int(bool(dgrad))))
    return ok


def _wino(g, dgrad, x2=None, infer=False):
    """does this pass of this convolution run on the Winograd kernels?"""
    if WINO == '0' or x2 is not None:
        return False
    if not infer and ((WINO == 'dgrad' and not dgrad) or (WINO == 'fwd' and dgrad)):
        return False
    key = (g.N, g.Ci, g.H, g.W, g.Co, g.KH, g.KW, g.SH, g.SW, g.PH, g.PW, g.c_split, bool(dgrad))
    ok = _WINO_OK.get(key)
    if ok is None:
        # 1: stride-1 three-tap / 3x3 (Winograd); 2 (input gradient only): stride-2 three-tap (polyphase form of the same kernel)
        ok = _WINO_OK[key] = int(_lib().dynmm_conv2d_wino_supported(C.byref(g), int(bool(dgrad))))
    return ok


# Opt-in (TrainStep / bench): when a parameter already owns a contiguous `.grad` buffer (a view into the
# flat gradient buffer of dp.GradBucketReducer), backward kernels write the parameter gradient
# STRAIGHT into it and return None to autograd — no temporary, no per-parameter accumulate kernel
# (607 tiny adds per step).  Semantics: overwrite, so valid for one backward per zero().
DIRECT_GRAD = False


# Opt-in with DIRECT_GRAD: launch every conv weight-gradient kernel on a dedicated stream.  dgrad and
# wgrad of a layer depend only on the incoming gradient, so the wgrad queue runs concurrently with
# the main backward chain and fills its ramp-up / tail bubbles.  Tensors it reads are kept alive in
# _INFLIGHT until join_async() (call it after backward, on the stream that consumes the gradients).
ASYNC_WGRAD = False
WGRAD_STREAMS = 2        # weight-gradient side streams.  With the null stream and the depth-encoder stream that makes the FOUR busy
                         # streams the runtime's hardware queues hold: a third one costs 10 ms per step (profiles/r05_ab_runs.md:
                         # 74.5 against 63.7 ms; one stream: 64.6)
_WGRAD_RR = [0]
_INFLIGHT = []


# ---- the stream plan: a PROCESS-WIDE invariant ----------------------------------------------------------------------------------
# The optimisation step runs on FOUR busy streams per device: the caller's stream (torch's current stream), ONE depth-encoder
# stream and WGRAD_STREAMS = 2 weight-gradient streams of the least priority.  A fifth busy stream costs the step ~10 ms whatever
# runs on it (profiles/r05_ab_runs.md, five sightings), and round 5's per-model-instance depth stream (`torch.cuda.Stream()` = the
# NEXT stream of torch's round-robin pool for every model built in a process) made the second, third ... model of a process run
# 17 % slower than the first (profiles/r06_stream_plan.md holds the experiment).  So the streams are created
# ONCE per device, through the runtime (never from torch's pool), in a fixed order, and every user — nn/net.py encoder_stage_pair,
# the weight-gradient queues, dp.GradBucketReducer's exchange, engine.TrainStep's capture warm-up — takes them from here.
MAX_BUSY_STREAMS = 4
VERIFY_STREAM_PLAN = True      # check at creation that the plan's streams run concurrently (_StreamPlan.verify)
_PLANS = {}                    # device index -> _StreamPlan
_HIP_RT = []
_CENSUS = set()                # stream handles kernels were enqueued on since stream_census_reset()


def _hip_runtime():
    """the HIP runtime ALREADY mapped in this process (torch's), by its path: a bare dlopen('libamdhip64.so') could bring a second
    runtime in, whose streams mean nothing to the first.  None when that is ambiguous."""
    if not _HIP_RT:
        mapped = sorted(L._mapped_hip_runtimes())
        hip = C.CDLL(mapped[0]) if len(mapped) == 1 else None
        if hip is not None:
            hip.hipDeviceGetStreamPriorityRange.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
            hip.hipDeviceGetStreamPriorityRange.restype = C.c_int
            hip.hipStreamCreateWithPriority.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_int]
            hip.hipStreamCreateWithPriority.restype = C.c_int
            hip.hipStreamGetPriority.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            hip.hipStreamGetPriority.restype = C.c_int
        _HIP_RT.append(hip)
    return _HIP_RT[0]


def _runtime_stream(device, least_priority):
    """A non-blocking HIP stream on `device` created through the runtime and wrapped (torch.cuda.ExternalStream): of the LEAST
    priority the device offers (torch.cuda.Stream reaches only normal and high, priority <= 0) or of the normal one.  Falls back
    to a torch pool stream only when the runtime cannot be reached.  Never destroyed: the plan lives as long as the process."""
    hip = _hip_runtime()
    with torch.cuda.device(device):
        torch.cuda.current_stream()                 # (makes sure the device context exists)
        if hip is None:
            return torch.cuda.Stream(device=device), None
        prio = 0
        if least_priority:
            least, greatest = C.c_int(0), C.c_int(0)
            if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) != 0 or least.value <= 0:
                return torch.cuda.Stream(device=device), None      # no priority below normal on this device
            prio = least.value
        h = C.c_void_p()
        if hip.hipStreamCreateWithPriority(C.byref(h), 1, prio) != 0 or not h.value:      # 1 = hipStreamNonBlocking
            return torch.cuda.Stream(device=device), None
        return torch.cuda.ExternalStream(h.value, device=device), h


class _StreamPlan:
    def __init__(self, device):
        if torch.cuda.is_current_stream_capturing():
            raise L.DynmmHipError('the stream plan must exist before a stream capture starts (hipStreamCreate is not '
                                  'capturable): build engine.TrainStep / call ops.stream_plan() first')
        self.device = device
        self.handles = []              # the runtime streams behind the ExternalStream objects (kept: never destroyed)
        self.side = self._make(False)
        self.wgrad = [self._make(True) for _ in range(max(1, WGRAD_STREAMS))]
        self.report = {'checked': False}
        if VERIFY_STREAM_PLAN:
            self.verify()

    def _make(self, least_priority):
        s, h = _runtime_stream(self.device, least_priority)
        if h is not None:
            self.handles.append(h)
        return s

    def grow(self):
        while len(self.wgrad) < max(1, WGRAD_STREAMS):
            self.wgrad.append(self._make(True))

    # -- do the four streams really run side by side? ------------------------------------------------------------------------
    # The runtime multiplexes a process's streams onto a few hardware queues in creation order, and two streams that share one
    # serialise: the +10 ms cliff of profiles/r05_ab_runs.md / r06_stream_plan.md.  Which queue a new stream lands on depends on
    # every stream the process created before (torch's pools, RCCL's, a data loader's) — so the plan is CHECKED, not assumed:
    # two spin kernels (torch.cuda._sleep, one workgroup each) on a pair of streams take 1.2x one spin when the streams are
    # independent and 2.1x when they share a queue (scratch/r6/queue_probe.py: the ratio separates the classes cleanly; a third
    # class at 1.45 - 1.65 — neighbouring queues of one pipe — is rejected too when a cleaner stream can be had).  A stream that
    # collides with the caller's stream or with an earlier stream of the plan is replaced by a newly created one, a bounded
    # number of times; what was found is kept in `self.report` (bench.py prints it).
    SPIN_CYCLES = 1000000               # ~0.5 ms per spin: long against the ~20 us of launch overhead both measurements carry
    CLEAN, TRIES = 1.35, 6

    def _pair_ratio(self, a, b, single_ms):
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.synchronize()
            b.synchronize()
            e0.record(a)
            b.wait_event(e0)
            with torch.cuda.stream(a):
                torch.cuda._sleep(self.SPIN_CYCLES)
            with torch.cuda.stream(b):
                torch.cuda._sleep(self.SPIN_CYCLES)
            a.wait_stream(b)
            e1.record(a)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best / single_ms

    def verify(self):
        self.report = {'checked': False}
        if not hasattr(torch.cuda, '_sleep') or torch.cuda.is_current_stream_capturing():
            return self.report
        with torch.cuda.device(self.device):
            caller = torch.cuda.current_stream()
            with torch.cuda.stream(caller):
                torch.cuda._sleep(self.SPIN_CYCLES)                 # (first launch of the spin kernel: module load)
            caller.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(caller)
            with torch.cuda.stream(caller):
                torch.cuda._sleep(self.SPIN_CYCLES)
            e1.record(caller)
            e1.synchronize()
            single = max(e0.elapsed_time(e1), 1e-3)
            replaced, worst = 0, {}
            fixed = [('caller', caller)]
            order = [('depth', False)] + [(f'wgrad{i}', True) for i in range(len(self.wgrad))]
            for name, low in order:
                for attempt in range(self.TRIES + 1):
                    s = self.side if name == 'depth' else self.wgrad[int(name[5:])]
                    ratios = {fn: round(self._pair_ratio(fs, s, single), 2) for fn, fs in fixed}
                    if max(ratios.values()) <= self.CLEAN or attempt == self.TRIES:
                        break
                    fresh = self._make(low)                          # (the rejected stream stays alive, idle: never destroyed)
                    if name == 'depth':
                        self.side = fresh
                    else:
                        self.wgrad[int(name[5:])] = fresh
                    replaced += 1
                worst.update({f'{fn}|{name}': r for fn, r in ratios.items()})
                fixed.append((name, s))
            self.report = {'checked': True, 'spin_ms': round(single, 3), 'pair_ratio': worst, 'replaced': replaced,
                           'clean': bool(max(worst.values()) <= self.CLEAN)}
        return self.report

    def streams(self):
        return [self.side] + self.wgrad[:max(1, WGRAD_STREAMS)]


def stream_plan(device=None):
    """The side streams of `device` (default: the current one), created on first call.  engine.TrainStep and the data-parallel
    reducer call this in their constructors, i.e. before any capture."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    plan = _PLANS.get(dev)
    if plan is None:
        plan = _PLANS[dev] = _StreamPlan(dev)
    elif len(plan.wgrad) < max(1, WGRAD_STREAMS):
        plan.grow()
    return plan


def low_priority_stream():
    """A NEW stream of the least priority the device offers (tests / experiments; the step takes its streams from stream_plan()).
    Weight-gradient launches fill the matrix pipe behind the backward's dependent chain (input gradient -> BatchNorm backward ->
    input gradient), whose workgroups the dispatcher takes first — 7 alternating pairs of runs on one box: -0.18 ms per step on
    average, never slower (round 5; the opposite, high-priority chains, cost +0.75 ms in round 3)."""
    s, h = _runtime_stream(torch.cuda.current_device(), True)
    if h is not None:
        stream_plan().handles.append(h)
    return s


def side_stream():
    """THE depth-encoder stream of the current device (normal priority)."""
    return stream_plan().side


def _wgrad_stream():
    pool = stream_plan().wgrad
    _WGRAD_RR[0] = (_WGRAD_RR[0] + 1) % max(1, WGRAD_STREAMS)
    return pool[_WGRAD_RR[0]]


def exchange_stream():
    """The stream the data-parallel gradient exchange is enqueued on (dp.GradBucketReducer): the LAST weight-gradient stream —
    not a stream of its own, see MAX_BUSY_STREAMS."""
    return stream_plan().wgrad[max(1, WGRAD_STREAMS) - 1]


def stream_census_reset():
    _CENSUS.clear()


def busy_streams():
    """Handles of the streams this library's launches were enqueued on since stream_census_reset() (every launch takes its
    stream from _stream() or from the plan)."""
    return set(_CENSUS)


def stream_census(check=True):
    """{'streams': n, 'roles': [...]} of the launches since stream_census_reset(); raises when the step went beyond
    MAX_BUSY_STREAMS (a stream outside the plan: some caller created one of its own)."""
    plan = stream_plan()
    names = {plan.side.cuda_stream: 'depth'}
    for i, s in enumerate(plan.wgrad):
        names[s.cuda_stream] = f'wgrad{i}'
    roles = sorted(names.get(h, 'caller' if h == torch.cuda.current_stream().cuda_stream else f'other:{h:#x}') for h in _CENSUS)
    if check and len(_CENSUS) > MAX_BUSY_STREAMS:
        raise L.DynmmHipError(f'this step enqueued work on {len(_CENSUS)} streams ({roles}); the schedule holds '
                              f'{MAX_BUSY_STREAMS} (ops.stream_plan): a further busy stream costs it ~10 ms')
    return {'streams': len(_CENSUS), 'roles': roles}


def join_async():
    flush_wgrad_groups()
    if _INFLIGHT:
        cur = torch.cuda.current_stream()
        for s in stream_plan().wgrad:
            cur.wait_stream(s)
    _INFLIGHT.clear()


# Grouped weight gradients (dynmm_conv2d_wgrad_group): under the in-place gradient protocol a convolution's weight
# gradient is queued by geometry instead of launched; WGRAD_GROUP same-shape problems (the factorised convs of
# consecutive residual blocks, RGB and depth encoder alike) go out as ONE launch.  Queues are flushed by join_async()
# at the end of backward at the latest; the parameters are reported to the gradient reducer when their launch is
# actually enqueued.
WGRAD_GROUP = 4
WGRAD_GROUP_AGE = 4
_WGRAD_QUEUES = {}
_CAPTURE_EVENTS = []       # events recorded while a stream capture was in progress (kept alive: see _queue_wgrad)
_WGRAD_TICK = [0]          # conv backward calls seen; a queue that got nothing for WGRAD_GROUP_AGE of them is flushed
_WGRAD_LAST = {}           # (the backward has left the run of layers with that geometry: do not hold its gradients back)


def _age_wgrad_queues():
    _WGRAD_TICK[0] += 1
    if _WGRAD_QUEUES:
        now = _WGRAD_TICK[0]
        for key in [k for k in _WGRAD_QUEUES if now - _WGRAD_LAST.get(k, now) > WGRAD_GROUP_AGE]:
            _flush_wgrad_queue(key)


def _queue_wgrad(g, x, gy, w_param, b_param):
    st = torch.cuda.current_stream()                    # gy is produced on this stream
    use_async = ASYNC_WGRAD and PROFILE is None
    # without the weight-gradient streams a group is launched on the stream its members were produced on: one queue per
    # stream, no events (the branches of the ModalityDynMM step each run on their own stream)
    key = (g.N, g.Ci, g.H, g.W, g.Co, g.KH, g.KW, g.SH, g.SW, g.PH, g.PW, b_param is not None,
           0 if use_async else st.cuda_stream)
    ev = None
    if use_async:
        ev = torch.cuda.Event()
        ev.record(st)
        if torch.cuda.is_current_stream_capturing():
            # an event recorded into a capture must outlive the capture: torch destroys an Event object the moment its last
            # reference goes (here: when its group is flushed, still inside the capture), and hipStreamEndCapture then
            # intermittently faults on the freed node (seen as a segfault in capture_end, order- and timing-dependent)
            _CAPTURE_EVENTS.append(ev)
    q = _WGRAD_QUEUES.setdefault(key, [])
    q.append((g, x, gy, w_param, b_param, ev, st))
    _WGRAD_LAST[key] = _WGRAD_TICK[0]
    if len(q) >= WGRAD_GROUP:
        _flush_wgrad_queue(key)


def _flush_wgrad_queue(key):
    q = _WGRAD_QUEUES.pop(key, None)
    _WGRAD_LAST.pop(key, None)
    if not q:
        return
    lib = _lib()
    g = q[0][0]
    n = len(q)
    use_async = ASYNC_WGRAD and PROFILE is None
    stream = _wgrad_stream() if use_async else q[0][6]
    for item in q:
        if item[5] is not None:
            stream.wait_event(item[5])
        elif item[6] != stream:
            stream.wait_stream(item[6])
    # parameters registered by the backward that triggered this flush belong to ITS streams: set them aside
    held = _PENDING[:]
    del _PENDING[:]
    dws = [_grad_dst(item[3])[0] for item in q]
    dbs = [_grad_dst(item[4])[0] for item in q] if q[0][4] is not None else None
    nbytes = lib.dynmm_conv2d_wgrad_group_workspace_bytes(C.byref(g), n)
    _CENSUS.add(stream.cuda_stream)
    with torch.cuda.stream(stream):
        ws = torch.empty(max(nbytes // 4, 1), device=q[0][1].device, dtype=torch.float32)
        xs = (C.c_void_p * n)(*[item[1].data_ptr() for item in q])
        dys = (C.c_void_p * n)(*[item[2].data_ptr() for item in q])
        dwa = (C.c_void_p * n)(*[t.data_ptr() for t in dws])
        dba = (C.c_void_p * n)(*[t.data_ptr() for t in dbs]) if dbs is not None else None

        def call():
            return lib.dynmm_conv2d_wgrad_group(n, xs, dys, dwa, dba, _p(ws), nbytes, C.byref(g), stream.cuda_stream)
        if PROFILE is not None:
            L.check(_timed('wgrad', g, call, n), 'conv2d_wgrad_group')
        else:
            L.check(call(), 'conv2d_wgrad_group')
    if use_async:
        _INFLIGHT.append(tuple(t for item in q for t in (item[1], item[2])))
    _grads_enqueued(stream)
    _PENDING.extend(held)


def flush_wgrad_groups():
    for key in list(_WGRAD_QUEUES):
        _flush_wgrad_queue(key)


# Bookkeeping of the in-place gradient protocol.  Every backward that writes a parameter gradient straight
# into its `.grad` view registers the parameter in _PENDING (via _grad_dst) and, once the kernels that write
# it are enqueued, calls _grads_enqueued(streams): the parameter joins the set of parameters touched in this
# step (the flat optimizers update only those ranges, as torch.optim skips parameters whose grad is None) and
# GRAD_READY_HOOK — installed by dp.GradBucketReducer — learns that the gradient will be complete once the
# given streams reach this point, so a bucket's all-reduce can start while the rest of backward still runs.
GRAD_READY_HOOK = None
_PENDING = []
_TOUCHED = set()


def touched_reset():
    _TOUCHED.clear()
    _PENDING.clear()


def touched_ids():
    return frozenset(_TOUCHED)


def _direct_ok(param):
    """_grad_dst(param) would hand out the parameter's own .grad view"""
    return DIRECT_GRAD and param is not None and getattr(param, 'grad', None) is not None and \
        param.grad.is_contiguous() and param.grad.dtype == torch.float32


def _grad_dst(param, like=None):
    """(tensor to write the gradient into, value to return to autograd)."""
    if DIRECT_GRAD and param is not None and getattr(param, 'grad', None) is not None and param.grad.is_contiguous() \
            and param.grad.dtype == torch.float32:
        _PENDING.append(param)
        return param.grad, None
    t = torch.empty_like(param if like is None else like)
    return t, t


def _grads_enqueued(*streams):
    """The kernels writing the gradients handed out by _grad_dst since the last call are enqueued on `streams`
    (default: the current stream)."""
    if not _PENDING:
        return
    hook = GRAD_READY_HOOK
    if hook is not None:
        ss = [s for s in streams if s is not None] or [torch.cuda.current_stream()]
        for prm in _PENDING:
            hook(prm, ss)
    for prm in _PENDING:
        _TOUCHED.add(id(prm))
    _PENDING.clear()


@contextlib.contextmanager
def capture_scope():
    """Around a stream capture: Python's cyclic collector is switched off.  torch.cuda.graph() collects BEFORE capture_begin, but
    a generation-0 collection can still start at any allocation inside the captured step, and what it finalises (Event / Stream
    / CUDAGraph objects of earlier steps, captures and tests) calls into the HIP runtime in the middle of the capture — seen as
    an intermittent segfault in hipStreamEndCapture whose frequency depended on how much garbage the process had accumulated."""
    import gc
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


# Tests (tests/test_hip_blocks.py): when ACT_TRACE is a list, every ReLU output of conv2d / batch_norm_act is appended to it in
# call order — the parity tests read the ReLU DECISIONS the HIP pass took and impose them on the fp64 oracle wherever the
# oracle's own pre-activation is within rounding of zero (two correct fp32 evaluations disagree there).
ACT_TRACE = None

# Optional per-launch timing of the implicit-GEMM kernels (bench.py's roofline leg): when PROFILE is a
# list, every conv launch is bracketed by events recorded on the stream it is launched on.
PROFILE = None


def _tile(co, m=1 << 30):
    """Tile the library picks (conv_igemm.hip: launch_igemm) for a GEMM with `co` rows and `m` pixels."""
    if co > 64:
        return '128x32' if -(-co // 128) * -(-m // 64) < 3 * 256 else '128x64'
    return '64x128' if co > 32 else '32x256'




def _timed(kind, g, call, nprob=1, extra=0, wino=False):
    if PROFILE is None:
        return call()
    co = g.Ci if kind == 'dgrad' else g.Co
    gemm_ci = g.Co if kind == 'dgrad' else g.Ci
    rk = ((gemm_ci + 15) & ~15) if (gemm_ci >= 8 and gemm_ci % 16) else gemm_ci      # conv_igemm.hip: round_k
    generic = ''
    if kind != 'wgrad':
        if rk % 16 != 0 or (g.c_split < g.Ci and g.c_split % 16 != 0):
            generic = ',generic'
        elif rk != gemm_ci:
            generic = ',padded'
    m = g.N * (g.H * g.W if kind == 'dgrad' else g.Ho * g.Wo)
    name = f'conv_wgrad<co{_tile(co).split("x")[0]}>' if kind == 'wgrad' else f'conv_igemm_{kind}<{_tile(co, m)}{generic}>'
    if kind == 'wgrad':
        variant = _lib().dynmm_conv2d_wgrad_variant(C.byref(g))
        if variant == 6 and (g.SH == 2 or g.SW == 2):       # conv_wgrad_s2.hip: the stride-2 three-tap convolutions, direct form
            name = f'conv_wgrad_s2<co{128 if g.Co % 128 == 0 else 64},{g.KH}x{g.KW}>'
        elif variant == 6:                                  # conv_wgrad_v6.hip: one template instance per tile height and tap axis
            name = f'conv_wgrad_v6<co{128 if g.Co % 128 == 0 else 64},{g.KH}x{g.KW}>'
        elif variant == 4:
            name = 'conv_wgrad_v4<co128>'                   # the vectorised 128x128 kernel (conv_igemm.hip: wgrad_v4_shape_ok)
        elif variant == 8:
            name = 'conv_co8_wgrad<direct,valu>'            # conv_small.hip: the gate conv's weight + bias gradient
    if kind != 'wgrad' and not generic and _lib().dynmm_conv2d_uses_operand_ring(C.byref(g), int(kind == 'dgrad')):
        name = f'conv_igemm_v5_{kind}<{"128x64" if co % 128 == 0 else "64x128"},kw{g.KW}>'      # conv_igemm_v5.hip
    if wino == 22:                               # conv_wino2d.hip
        name = f'conv_wino2d_{kind}<co{128 if co % 128 == 0 else 64},3x3>'
    elif wino:                                   # conv_wino.hip: one template instance per tile height, tap axis and direction
        name = f'conv_wino{"43" if wino == 43 else ""}_{kind}<co{128 if co % 128 == 0 else 64},{g.KH}x{g.KW}{"s2" if g.SH * g.SW > 1 else ""}>'
    if kind == 'fwd':                            # conv_small.hip: *_eligible (the library's own dispatch rules)
        k5, k7 = (g.KH, g.KW, g.SH, g.SW, g.PH, g.PW) == (5, 5, 2, 2, 0, 0), (g.KH, g.KW, g.SH, g.SW, g.PH, g.PW) == (7, 7, 2, 2, 3, 3)
        if k5 and 5 <= g.Co <= 8 and g.Ci % 4 == 0 and g.Ci >= 16 and (g.c_split == g.Ci or g.c_split % (g.Ci // 4) == 0):
            name = 'conv_co8_fwd<direct,valu>'
        elif k7 and g.Ci in (1, 3) and g.Co == 64 and g.c_split == g.Ci:
            name = f'conv_stem_fwd<ci{g.Ci},mfma>'
    flops = 2.0 * g.N * g.Ho * g.Wo * g.KH * g.KW * g.Ci * g.Co * nprob      # algorithmic (= forward MACs x2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = call()
    e1.record()
    # extra: epilogue operands of the launch that are tensors of the OUTPUT's size (ReLU-mask source, residual / accumulated gradient)
    PROFILE.append((name, flops, e0, e1, (g.N, g.Ci, g.H, g.W, g.Co, g.KH, g.KW, g.SH, g.SW, nprob), (kind, int(extra))))
    return r


def _up4(v):
    return (v + 3) & ~3


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _same_shape(a, b, what):
    if b is not None and tuple(a.shape) != tuple(b.shape):
        raise L.DynmmHipError(f'{what}: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}')


def _geom(x, x2, weight, stride, padding):
    if x.dim() != 4 or weight.dim() != 4:
        raise L.DynmmHipError('conv2d expects NCHW input and OIHW weight')
    if x2 is not None and (x2.shape[0] != x.shape[0] or tuple(x2.shape[2:]) != tuple(x.shape[2:])):
        raise L.DynmmHipError(f'conv2d: second input {tuple(x2.shape)} does not match {tuple(x.shape)}')
    N, C0, H, W = x.shape
    Ci = C0 + (x2.shape[1] if x2 is not None else 0)
    Co, Ciw, KH, KW = weight.shape
    if Ciw != Ci:
        raise L.DynmmHipError(f'conv2d: weight expects {Ciw} input channels, got {Ci}')
    SH, SW = stride
    PH, PW = padding
    Ho = (H + 2 * PH - KH) // SH + 1
    Wo = (W + 2 * PW - KW) // SW + 1
    if N <= 0 or Ho <= 0 or Wo <= 0:
        raise L.DynmmHipError(f'conv2d: empty output for input {tuple(x.shape)}, kernel {(KH, KW)}, stride {stride}, '
                              f'padding {padding}')
    g = L.ConvGeom(N, Ci, H, W, Co, Ho, Wo, KH, KW, SH, SW, PH, PW, C0)
    return g


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
class GradLink:
    """Hands the residual-branch gradient of a block from the op that produces it (the BatchNorm that
    adds the identity) to the op that can absorb it for free (the epilogue of the first conv's dgrad),
    instead of letting autograd materialise `dgrad + dres` with a separate add pass."""
    __slots__ = ('dres',)

    def __init__(self):
        self.dres = None


class BNLink:
    """Joins a training-mode BatchNorm + ReLU to the single convolution that consumes its output, for the backward pass: the
    convolution's input-gradient launch (csrc/conv_wino.hip BNRED) also leaves the BatchNorm backward's two reductions, and the
    BatchNorm's backward then starts at its apply pass.  batch_norm_act fills `x, mean, invstd, gamma, beta` in the forward; the
    convolution's backward fills `sums` when its kernel took the job (it runs first: gradients flow consumer -> producer).

    Second form (`bits` set; BNRED == 2): the BatchNorm adds an identity branch before the ReLU (bn2 of a residual block) and its
    output feeds the NEXT block and nothing else — that block's first convolution (this link's consumer) and its identity branch,
    whose gradient that convolution's input-gradient epilogue absorbs (GradLink).  The launch then holds the complete gradient g
    of the BatchNorm's output: it masks it with the forward's one-bit ReLU decisions, writes g.[out > 0] and leaves the two
    reductions; `premasked` tells the BatchNorm's backward that its incoming gradient already carries the mask (its apply pass
    runs without the bits, and the identity branch's gradient IS that tensor: no second masked copy is written)."""
    __slots__ = ('x', 'mean', 'invstd', 'gamma', 'beta', 'sums', 'bits', 'premasked')

    def __init__(self):
        self.x = self.mean = self.invstd = self.gamma = self.beta = self.sums = self.bits = None
        self.premasked = False


BN_BWD_FUSE = True       # (module attribute: tests switch it to compare with the bn_bwd_reduce path)
# BatchNorm + residual + ReLU (bn2 of every residual block): the ReLU decisions travel to the backward as one bit per element
# instead of the 4-byte output tensor (tests / A-B: ops.BN_RELU_BITS = False)
BN_RELU_BITS = True


class PackedWeights:
    """The operand layouts of every conv weight a training step uses — the implicit-GEMM layouts (dynmm_pack_weight) and the
    Winograd filter transforms (dynmm_wino_pack) — produced by ONE launch each per step instead of one per convolution (186
    for config P).  engine.TrainStep installs an instance as ops.PREPACK: the first step runs the ordinary per-conv packs
    and registers (weight, shape, which of the four operands its passes read); from then on pack() fills a static arena at
    the start of the step body and the convolutions look their operands up.  Entries are valid between pack() and
    invalidate() only (the optimizer rewrites the weights after the body)."""

    def __init__(self):
        self.reg = {}            # id(weight) -> [weight, Co, Ci, KH, KW, need_fwd, need_dgrad, wino_fwd, wino_dgrad, wino43_dgrad, wino2d_fwd, wino2d_dgrad]
        self.slots = {}          # id(weight) -> (wp, wpd, utf, utd, utd43, ut2f, ut2d), None where not needed
        self.arena = self.desc = self.wdesc = self.w43desc = self.w2desc = None
        self.blocks = self.wblocks = self.nwino = self.w43blocks = self.nw43 = self.w2blocks = self.nw2 = 0
        self.valid = False
        self.dirty = False       # registrations since the arena was laid out

    def register(self, weight, g, need_fwd=True, need_dgrad=False, wino_fwd=False, wino_dgrad=False, wino43_dgrad=False,
                 wino2d_fwd=False, wino2d_dgrad=False):
        if not isinstance(weight, torch.nn.Parameter):
            return
        flags = [bool(need_fwd), bool(need_dgrad), bool(wino_fwd), int(wino_dgrad), bool(wino43_dgrad),     # wino_dgrad: 0 | 1 | 2
                 bool(wino2d_fwd), bool(wino2d_dgrad)]
        e = self.reg.get(id(weight))
        if e is None:
            self.reg[id(weight)] = [weight, g.Co, g.Ci, g.KH, g.KW] + flags
            self.dirty = True
        else:
            for i, f in enumerate(flags):
                if f and e[5 + i] != f:
                    # (wino_dgrad is tri-state — 1: stride-1 transform, 2: stride-2 polyphase operand; the two are different
                    # operands of different kernels, so a weight is registered with exactly one of them: store the value)
                    if i == 3 and e[5 + i] and f:
                        raise L.DynmmHipError('a convolution weight was registered with both Winograd input-gradient operand kinds')
                    e[5 + i] = f
                    self.dirty = True

    def lookup(self, weight, need_fwd=True, need_dgrad=False, wino_fwd=False, wino_dgrad=False, wino43_dgrad=False,
               wino2d_fwd=False, wino2d_dgrad=False):
        if not self.valid:
            return None
        s = self.slots.get(id(weight))
        if s is None:
            return None
        for need, t in zip((need_fwd, need_dgrad, wino_fwd, wino_dgrad, wino43_dgrad, wino2d_fwd, wino2d_dgrad), s):
            if need and t is None:
                return None
        if wino_dgrad and self.reg[id(weight)][5 + 3] != int(wino_dgrad):
            return None                          # the slot holds the other operand kind (stride-1 transform vs stride-2 polyphase)
        return s

    def _layout(self):
        lib = _lib()
        ents = list(self.reg.values())
        dev = ents[0][0].device
        base = min(e[0].data_ptr() for e in ents)
        off, blk, wblk, w43blk, w2blk, rows, wrows, w43rows, w2rows = 0, 0, 0, 0, 0, [], [], [], []
        spans = {}
        for w, Co, Ci, KH, KW, nf_, nd, wf, wd, wd43, w2f, w2d in ents:
            src = (w.data_ptr() - base) // 4
            span = [None] * 7
            # the multi-tensor pack always writes the forward layout; the input-gradient layout only where a direct
            # kernel reads it (dynmm_pack_weight_multi's descriptor: dst_dgrad = -1 otherwise)
            nf = lib.dynmm_packed_weight_floats(Co, Ci, KH, KW, 0)
            ndg = lib.dynmm_packed_weight_floats(Co, Ci, KH, KW, 1) if nd else 0
            if nf_ or nd:
                dstf = off
                off += (nf + 3) & ~3             # rows stay 16-byte aligned (the kernels' dwordx4 path)
                dstd = off if nd else -1
                off += (ndg + 3) & ~3
                rows.append([src, dstf, dstd, Co | (Ci << 32), (KH * KW) | (blk << 32)])
                span[0] = (dstf, nf)
                span[1] = (dstd, ndg) if nd else None
                blk += lib.dynmm_pack_weight_multi_blocks(Co, Ci, KH, KW, int(nd))
            nu = lib.dynmm_wino_packed_floats(Co, Ci, KH, KW)
            for slot, dgrad, on in ((2, 0, wf), (3, 2 if wd == 2 else 1, wd)):     # (wd == 2: stride-2 polyphase operand)
                if on:
                    wrows.append([src, off, Co | (Ci << 32), KH | (KW << 8) | (dgrad << 16) | (wblk << 32)])
                    span[slot] = (off, nu)
                    off += nu                    # a multiple of 4 floats
                    wblk += lib.dynmm_wino_pack_multi_blocks(Co, Ci, KH, KW)
            if wd43:
                nu43 = lib.dynmm_wino43_packed_floats(Co, Ci, KH, KW)
                w43rows.append([src, off, Co | (Ci << 32), KH | (KW << 8) | (w43blk << 32)])
                span[4] = (off, nu43)
                off += (nu43 + 3) & ~3
                w43blk += lib.dynmm_wino43_pack_multi_blocks(Co, Ci, KH, KW)
            nu2 = lib.dynmm_wino2d_packed_floats(Co, Ci) if (w2f or w2d) else 0
            for slot, dgrad, on in ((5, 0, w2f), (6, 1, w2d)):
                if on:
                    w2rows.append([src, off, Co | (Ci << 32), (dgrad << 16) | (w2blk << 32)])
                    span[slot] = (off, nu2)
                    off += nu2                   # a multiple of 4 floats
                    w2blk += lib.dynmm_wino2d_pack_multi_blocks(Co, Ci, dgrad)
            spans[id(w)] = span
        self.arena = torch.empty(off, device=dev, dtype=torch.float32)
        self.desc = torch.tensor(rows, dtype=torch.int64).to(dev) if rows else None
        self.wdesc = torch.tensor(wrows, dtype=torch.int64).to(dev) if wrows else None
        self.w43desc = torch.tensor(w43rows, dtype=torch.int64).to(dev) if w43rows else None
        self.w2desc = torch.tensor(w2rows, dtype=torch.int64).to(dev) if w2rows else None
        self.ndesc, self.nwino, self.nw43, self.nw2 = len(rows), len(wrows), len(w43rows), len(w2rows)
        self.base, self.blocks, self.wblocks, self.w43blocks, self.w2blocks = base, blk, wblk, w43blk, w2blk
        self.slots = {k: tuple(self.arena[sp[0]:sp[0] + sp[1]] if sp is not None else None for sp in span)
                      for k, span in spans.items()}
        self.dirty = False

    def pack(self):
        """one launch per operand family: every registered weight -> the layouts its convolution's passes read"""
        self.valid = False
        if not self.reg:
            return
        if self.dirty or self.arena is None:
            if torch.cuda.is_current_stream_capturing():
                return                           # layout changes allocate: not inside a capture
            self._layout()
        if self.desc is not None:
            L.check(_lib().dynmm_pack_weight_multi(C.c_void_p(self.base), _p(self.arena), self.desc.data_ptr(), self.ndesc,
                                                   self.blocks, _stream()), 'pack_weight_multi')
        if self.wdesc is not None:
            L.check(_lib().dynmm_wino_pack_multi(C.c_void_p(self.base), _p(self.arena), self.wdesc.data_ptr(), self.nwino,
                                                 self.wblocks, _stream()), 'wino_pack_multi')
        if self.w43desc is not None:
            L.check(_lib().dynmm_wino43_pack_multi(C.c_void_p(self.base), _p(self.arena), self.w43desc.data_ptr(), self.nw43,
                                                   self.w43blocks, _stream()), 'wino43_pack_multi')
        if self.w2desc is not None:
            L.check(_lib().dynmm_wino2d_pack_multi(C.c_void_p(self.base), _p(self.arena), self.w2desc.data_ptr(), self.nw2,
                                                   self.w2blocks, _stream()), 'wino2d_pack_multi')
        self.valid = True

    def invalidate(self):
        self.valid = False


PREPACK = None


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, x2, weight, bias, stride, padding, act, mask_input, defer_mask, link, w_owner=None, stats=None,
                bn_link=None):
        lib = _lib()
        st = _stream()
        x, x2, weight, bias = _chk(x, 'x'), _chk(x2, 'x2'), _chk(weight, 'weight'), _chk(bias, 'bias')
        g = _geom(x, x2, weight, stride, padding)
        need_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
        y = torch.empty((g.N, g.Co, g.Ho, g.Wo), device=x.device, dtype=torch.float32)
        # (the Winograd kernels read their input with 16-byte loads; an input off that grid takes the direct kernels)
        w2f = _wino2d(g, False, x2) and x.data_ptr() % 16 == 0          # 3x3: the 2-D form; 1x3 / 3x1: the 1-D ones
        w2d = need_dx and _wino2d(g, True, x2)
        wino_f = not w2f and _wino(g, False, x2) and x.data_ptr() % 16 == 0
        wino_d = need_dx and not w2d and _wino(g, True, x2)
        wino_d43 = wino_d and _wino43(g)
        wino_d = 0 if wino_d43 else int(wino_d or 0)          # 0 | 1 (stride 1) | 2 (stride 2: polyphase form)
        need_wp, need_wpd = not (wino_f or w2f), need_dx and not (wino_d or wino_d43 or w2d)
        # (a Linear / Conv1d weight arrives as a [Co, Ci, 1, 1] alias of its parameter: same memory, so the parameter keys the pack)
        wkey = w_owner if w_owner is not None else weight
        wp = wpd = utf = utd = utd43 = ut2f = ut2d = None
        pre = PREPACK.lookup(wkey, need_wp, need_wpd, wino_f, wino_d, wino_d43, w2f, w2d) if PREPACK is not None else None
        if pre is not None:
            wp, wpd, utf, utd, utd43, ut2f, ut2d = pre           # packed by the step's multi-tensor launches
        else:
            if need_wp or need_wpd:
                wp = torch.empty(lib.dynmm_packed_weight_floats(g.Co, g.Ci, g.KH, g.KW, 0), device=x.device,
                                 dtype=torch.float32) if need_wp else None
                wpd = torch.empty(lib.dynmm_packed_weight_floats(g.Co, g.Ci, g.KH, g.KW, 1), device=x.device,
                                  dtype=torch.float32) if need_wpd else None
                L.check(lib.dynmm_pack_weight(_p(weight), _p(wp), _p(wpd), g.Co, g.Ci, g.KH, g.KW, st), 'pack_weight')
            nu = lib.dynmm_wino_packed_floats(g.Co, g.Ci, g.KH, g.KW) if (wino_f or wino_d) else 0
            if wino_f:
                utf = torch.empty(nu, device=x.device, dtype=torch.float32)
                L.check(lib.dynmm_wino_pack(_p(weight), _p(utf), None, g.Co, g.Ci, g.KH, g.KW, 0, st), 'wino_pack')
            if wino_d:
                utd = torch.empty(nu, device=x.device, dtype=torch.float32)
                L.check(lib.dynmm_wino_pack(_p(weight), _p(utd), None, g.Co, g.Ci, g.KH, g.KW, int(wino_d), st), 'wino_pack')
            if wino_d43:
                utd43 = torch.empty(lib.dynmm_wino43_packed_floats(g.Co, g.Ci, g.KH, g.KW), device=x.device, dtype=torch.float32)
                L.check(lib.dynmm_wino43_pack(_p(weight), _p(utd43), g.Co, g.Ci, g.KH, g.KW, st), 'wino43_pack')
            if w2f or w2d:
                nu2 = lib.dynmm_wino2d_packed_floats(g.Co, g.Ci)
                if w2f:
                    ut2f = torch.empty(nu2, device=x.device, dtype=torch.float32)
                    L.check(lib.dynmm_wino2d_pack(_p(weight), _p(ut2f), None, g.Co, g.Ci, 0, st), 'wino2d_pack')
                if w2d:
                    ut2d = torch.empty(nu2, device=x.device, dtype=torch.float32)
                    L.check(lib.dynmm_wino2d_pack(_p(weight), _p(ut2d), None, g.Co, g.Ci, 1, st), 'wino2d_pack')
            if PREPACK is not None:
                PREPACK.register(wkey, g, need_wp, need_wpd, wino_f, wino_d, wino_d43, w2f, w2d)
        if w2f:
            sums, ns = None, 0
            if stats is not None and act == L.ACT_NONE and g.Co % 64 == 0:
                # the consumer is a training-mode BatchNorm: its batch statistics come out of this launch
                ns = lib.dynmm_conv2d_wino2d_stats_slots(C.byref(g))
                sums, zeroed = _zero_sums(2 * g.Co * ns, x.device)
                if not zeroed:
                    sums.zero_()
                stats['sums'] = sums
            L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_wino2d_fwd(_p(x), _p(ut2f), _p(bias), None, _p(y), _p(sums), ns,
                                                                         C.byref(g), act, st), wino=22), 'conv2d_wino2d_fwd')
        elif (wino_f and stats is not None and act == L.ACT_NONE and g.KW == 3 and
                lib.dynmm_conv2d_wino_fwd_stats_supported(C.byref(g))):
            # the consumer is a training-mode BatchNorm: its batch statistics come out of this launch (stats: a holder the
            # conv2d() wrapper hangs on the output tensor for batch_norm_act to find)
            ns = lib.dynmm_conv2d_wino_fwd_stats_slots(C.byref(g))
            sums, zeroed = _zero_sums(2 * g.Co * ns, x.device)
            if not zeroed:
                sums.zero_()
            L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_wino_fwd_stats(_p(x), _p(utf), _p(bias), _p(y), _p(sums), ns,
                                                                             C.byref(g), st), wino=True), 'conv2d_wino_fwd_stats')
            stats['sums'] = sums
        elif wino_f:
            L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_wino_fwd(_p(x), _p(utf), _p(bias), None, _p(y), C.byref(g), act, st),
                           wino=True), 'conv2d_wino_fwd')
        elif (stats is not None and act == L.ACT_NONE and x2 is None and
                lib.dynmm_conv2d_stem_fwd_stats_supported(C.byref(g))):
            # a ResNet stem feeding its training-mode BatchNorm: the batch statistics come out of the convolution's epilogue
            sums, zeroed = _zero_sums(2 * g.Co, x.device)
            if not zeroed:
                sums.zero_()
            L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_stem_fwd_stats(_p(x), _p(wp), _p(bias), _p(y), _p(sums),
                                                                             C.byref(g), st)), 'conv2d_stem_fwd_stats')
            stats['sums'] = sums
        else:
            L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_fwd(_p(x), _p(x2), _p(wp), None, _p(bias), None, _p(y),
                                                                  C.byref(g), act, st)), 'conv2d_fwd')
        ctx.geom = g
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.has_x2 = x2 is not None
        ctx.wino_d = 22 if w2d else (43 if wino_d43 else (23 if wino_d else 0))
        ctx.mask_input = mask_input       # x is a ReLU output: apply [x > 0] in the dgrad epilogue
        ctx.defer_mask = defer_mask       # our own ReLU backward is applied by the consumer's dgrad
        ctx.link = link
        ctx.bn_link = bn_link
        ctx.save_for_backward(x, x2, ut2d if w2d else (utd43 if wino_d43 else (utd if wino_d else wpd)),
                              y if (act != L.ACT_NONE and not defer_mask) else None)
        ctx.wshape = tuple(weight.shape)
        # w_owner: the nn.Parameter that `weight` is a reshaped view of (Linear / Conv1d weights used as 1x1 convs): its
        # `.grad` has the same memory layout, so the in-place gradient protocol can write straight into it
        ctx.w_param, ctx.b_param = (w_owner if w_owner is not None else weight), bias
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib()
        st = _stream()
        x, x2, wpd, y = ctx.saved_tensors
        g = ctx.geom
        gy = _chk(gy, 'grad')
        dbias = None
        act = L.ACT_NONE if ctx.defer_mask else ctx.act     # deferred: gy arrives already masked
        dbias_ret = None
        # The bias gradient (sum of the masked gy over pixels) rides along with the weight-gradient kernel,
        # which stages every gy tile anyway; only when the weights take no gradient does it need its own pass.
        bias_in_wgrad = ctx.has_bias and ctx.needs_input_grad[2]
        # queued for a grouped launch (in-place gradient protocol only: the gradients go straight into .grad views)
        defer = (ctx.needs_input_grad[2] and DIRECT_GRAD and WGRAD_GROUP > 1 and x2 is None and
                 _direct_ok(ctx.w_param) and (not ctx.has_bias or _direct_ok(ctx.b_param)) and
                 bool(lib.dynmm_conv2d_wgrad_groupable(C.byref(g))))
        if ctx.has_bias and not (defer and bias_in_wgrad):
            dbias, dbias_ret = _grad_dst(ctx.b_param)
        if act != L.ACT_NONE or (ctx.has_bias and not bias_in_wgrad):
            ge = torch.empty_like(gy) if act != L.ACT_NONE else None
            wsb = None
            if not bias_in_wgrad and dbias is not None:
                nb = lib.dynmm_act_bwd_bias_workspace_bytes(g.N, g.Co)
                wsb = torch.empty(max(nb // 4, 1), device=gy.device, dtype=torch.float32)
            L.check(lib.dynmm_act_bwd_bias(_p(gy), _p(y), _p(ge), None if bias_in_wgrad else _p(dbias), _p(wsb),
                                           g.N, g.Co, g.Ho * g.Wo, act, st), 'act_bwd_bias')
            if ge is not None:
                gy = ge
        dx = dx2 = None
        if wpd is not None:
            dx = torch.empty_like(x)
            dx2 = torch.empty_like(x2) if x2 is not None else None
            mask = x if ctx.mask_input else None
            accum = None
            if ctx.link is not None and ctx.link.dres is not None:
                accum, ctx.link.dres = ctx.link.dres, None
            extra = (mask is not None) + (accum is not None)
            if ctx.wino_d:
                # 16-byte loads of gy, 8-byte accesses to the epilogue operands: tensors off that grid (views handed in by the
                # caller) are copied into fresh allocations first
                gyw = gy if gy.data_ptr() % 16 == 0 else gy.clone()
                mask, accum = (t if (t is None or t.data_ptr() % 16 == 0) else t.clone() for t in (mask, accum))
                bl = ctx.bn_link
                # ((ctx.link is None) == (accum is None): when this block has an identity branch its gradient must have ARRIVED here
                #  through the link — had it taken the autograd route instead (link.dres never set), the launch below would mask
                #  the convolution's part only while telling the BatchNorm backward the sum is masked: ADVICE r5)
                if (bl is not None and bl.x is not None and bl.bits is not None and ctx.wino_d == 23 and mask is None and
                        (ctx.link is None) == (accum is None) and
                        bl.x.data_ptr() % 8 == 0 and (g.H * g.W) % 4 == 0 and
                        lib.dynmm_conv2d_wino_dgrad_bnred_supported(C.byref(g))):
                    # x = relu(BN(c) + identity), this convolution and the identity branch behind `accum` its only consumers: the
                    # complete gradient of that output is formed here — mask it with the forward's decisions and leave the
                    # BatchNorm's backward reductions with the link
                    sums, zeroed = _zero_sums(2 * g.Ci * lib.dynmm_conv2d_wino_dgrad_bnred_slots(C.byref(g)), dx.device)
                    if not zeroed:
                        sums.zero_()
                    L.check(_timed('dgrad', g, lambda: lib.dynmm_conv2d_wino_dgrad_bnred2(
                        _p(gyw), _p(wpd), _p(accum), _p(bl.x), _p(bl.bits), _p(bl.mean), _p(bl.invstd), _p(sums), _p(dx),
                        C.byref(g), st), extra=2, wino=ctx.wino_d), 'conv2d_wino_dgrad_bnred2')
                    bl.sums, bl.premasked = sums, True
                elif (bl is not None and bl.x is not None and bl.bits is None and ctx.wino_d == 23 and mask is None and
                        accum is None and
                        bl.x.data_ptr() % 8 == 0 and lib.dynmm_conv2d_wino_dgrad_bnred_supported(C.byref(g))):
                    # x = relu(BN(c)) of a training-mode BatchNorm: mask by [BN(c) > 0] here and leave that BatchNorm's backward
                    # reductions with the link (its backward, which runs next, skips its reduction pass)
                    sums, zeroed = _zero_sums(2 * g.Ci * lib.dynmm_conv2d_wino_dgrad_bnred_slots(C.byref(g)), dx.device)
                    if not zeroed:
                        sums.zero_()
                    L.check(_timed('dgrad', g, lambda: lib.dynmm_conv2d_wino_dgrad_bnred(
                        _p(gyw), _p(wpd), _p(bl.x), _p(bl.mean), _p(bl.invstd), _p(bl.gamma), _p(bl.beta), _p(sums), _p(dx),
                        C.byref(g), st), extra=1, wino=ctx.wino_d), 'conv2d_wino_dgrad_bnred')
                    bl.sums = sums
                else:
                    fn = {22: lib.dynmm_conv2d_wino2d_dgrad, 43: lib.dynmm_conv2d_wino43_dgrad}.get(ctx.wino_d, lib.dynmm_conv2d_wino_dgrad)
                    L.check(_timed('dgrad', g, lambda: fn(_p(gyw), _p(wpd), _p(mask), _p(accum), _p(dx), C.byref(g), st),
                                   extra=extra, wino=ctx.wino_d), 'conv2d_wino_dgrad')
            else:
                L.check(_timed('dgrad', g, lambda: lib.dynmm_conv2d_dgrad(_p(gy), _p(wpd), _p(mask), _p(accum), _p(dx),
                                                                          _p(dx2), C.byref(g), st), extra=extra), 'conv2d_dgrad')
        dw_ret = None
        ws_stream = None
        if DIRECT_GRAD and WGRAD_GROUP > 1:
            _age_wgrad_queues()
        if defer:
            _queue_wgrad(g, x, gy, ctx.w_param, ctx.b_param if bias_in_wgrad else None)
        elif ctx.needs_input_grad[2]:
            dw, dw_ret = _grad_dst(ctx.w_param)
            nbytes = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g))
            if ASYNC_WGRAD and dw_ret is None and PROFILE is None:
                ws_stream = _wgrad_stream()
                _CENSUS.add(ws_stream.cuda_stream)
                ws_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(ws_stream):
                    ws = torch.empty(max(nbytes // 4, 1), device=gy.device, dtype=torch.float32)
                    L.check(lib.dynmm_conv2d_wgrad(_p(x), _p(x2), _p(gy), _p(dw), _p(dbias) if bias_in_wgrad else None,
                                                   _p(ws), nbytes, C.byref(g), ws_stream.cuda_stream), 'conv2d_wgrad')
                _INFLIGHT.append((x, x2, gy))
            else:
                ws = torch.empty(max(nbytes // 4, 1), device=gy.device, dtype=torch.float32)
                L.check(_timed('wgrad', g, lambda: lib.dynmm_conv2d_wgrad(_p(x), _p(x2), _p(gy), _p(dw),
                                                                          _p(dbias) if bias_in_wgrad else None, _p(ws), nbytes,
                                                                          C.byref(g), st)), 'conv2d_wgrad')
        _grads_enqueued(torch.cuda.current_stream(), ws_stream)
        if dw_ret is not None and tuple(dw_ret.shape) != ctx.wshape:
            dw_ret = dw_ret.reshape(ctx.wshape)
        return dx, dx2, dw_ret, dbias_ret, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, act=None, x2=None, mask_input=False, defer_mask=False,
           link=None, w_owner=None, bn_stats=False, bn_link=None):
    """act(conv2d(cat([x, x2], 1), weight) + bias).  Differentiable.

    Backward-fusion hints (set by block code that knows the dataflow; results are unchanged):
      defer_mask : this op's ReLU backward is applied by its (single) consumer — pair with
      mask_input : x is the output of a `defer_mask` op: the dgrad epilogue applies [x > 0];
      link       : GradLink whose residual-branch gradient is added in the dgrad epilogue.
    bn_stats: the output goes straight into a training-mode batch_norm_act: where the forward kernel can, it leaves the BatchNorm's
    batch statistics with the output (`y._bn_sums`) and batch_norm_act skips its statistics pass.
    bn_link: x is the output of batch_norm_act(..., 'relu', bwd_link=bn_link) and this convolution is its only consumer — or, for a
    BatchNorm with an identity branch, its only consumer besides the identity branch behind `link` (BNLink, both forms)."""
    if not torch.is_grad_enabled() and isinstance(weight, torch.nn.Parameter) and w_owner is None:
        # inference: the packed weight is cached on the parameter (conv2d_fused_eval) instead of re-laid-out per call
        # (the factorised blocks' conv -> ReLU pairs were 83 pack launches per forward of config P)
        return conv2d_fused_eval(x, weight, bias, None, act, None, stride, padding, x2)
    holder = {} if (bn_stats and CONV_BN_STATS and x2 is None) else None
    y = _Conv2d.apply(x, x2, weight, bias, _pair(stride), _pair(padding), ACT[act], bool(mask_input),
                      bool(defer_mask), link, w_owner, holder, bn_link if BN_BWD_FUSE else None)
    if holder:
        y._bn_sums = holder['sums']
    if ACT_TRACE is not None and ACT[act] == L.ACT_RELU:
        ACT_TRACE.append(y.detach())
    return y


class _FanOut(Function):
    """n aliases of x whose gradients are summed by ONE kernel of ours (left to right, deterministic) instead of
    autograd's n-1 pairwise accumulation passes."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [_chk(g, 'grad') for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty_like(gs[0])
        while len(gs) > 1:
            take, gs = gs[:4], gs[4:]
            dst = out if not gs else torch.empty_like(out)
            L.check(_lib().dynmm_add_n(_ptr_array(take), len(take), _p(dst), C.c_size_t(dst.numel()), _stream()), 'add_n')
            gs = [dst] + gs
        return out, None


_NO_FANOUT = False       # (tests / A-B: let autograd accumulate pairwise)


def fan_out(x, n):
    """x for n consumers: returns n aliases (no copy).  Outside autograd it is the identity."""
    if n <= 1 or _NO_FANOUT or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return _FanOut.apply(x, n)


_MUTATION_GEN = [0]    # bumped by every HIP kernel of ours that rewrites parameters / buffers through raw pointers


def note_mutation():
    """The kernels write running statistics, step counters and (fused SGD) parameters in place without going
    through torch's version counters; anything cached from those tensors is keyed on this generation too."""
    _MUTATION_GEN[0] += 1


# A stream capture normally re-folds / re-packs its inference weights INSIDE the graph (a replay then always reads the live
# weights).  engine.InferStep keys its graphs on the version stamp of every parameter and buffer and drops them when it moves,
# so its captures may read the cached operands instead (True only inside InferStep._capture): ~200 launches less per replay.
CAPTURE_EVAL_CACHE = False


def conv2d_fused_eval(x, weight, conv_bias, bn, act=None, residual=None, stride=1, padding=0, x2=None):
    """Inference-only: conv + folded eval-mode BatchNorm + residual + activation in ONE kernel.
    `bn` is None (plain bias) or an nn.BatchNorm2d holding running statistics."""
    lib = _lib()
    st = _stream()
    x, x2, weight = _chk(x, 'x'), _chk(x2, 'x2'), _chk(weight, 'weight')
    residual = _chk(residual, 'residual')
    g = _geom(x, x2, weight, _pair(stride), _pair(padding))
    dev = x.device
    # Inference weights do not change between calls: the packed weight tile and the folded BN scale/shift are
    # cached ON the layer's weight tensor object (so the cache dies with the layer), stamped with the storage
    # address + in-place version counter of every tensor they derive from (load_state_dict / optimizer steps bump
    # the versions) and the mutation generation above: a steady-state forward launches only the conv.
    srcs = [weight, conv_bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
    ok_ptrs = x.data_ptr() % 16 == 0 and (residual is None or residual.data_ptr() % 8 == 0)
    wino2 = _wino2d(g, False, x2, infer=True) and ok_ptrs
    wino = not wino2 and _wino(g, False, x2, infer=True) and ok_ptrs
    slot = '_dynmm_eval_cache_wino2d' if wino2 else ('_dynmm_eval_cache_wino' if wino else '_dynmm_eval_cache')
    stamp = (_MUTATION_GEN[0],) + tuple((t.data_ptr(), t._version) for t in srcs if t is not None) + \
        ((float(bn.eps),) if bn is not None else ())
    hit = getattr(weight, slot, None)
    if hit is not None and hit[0] == stamp and (CAPTURE_EVAL_CACHE or not torch.cuda.is_current_stream_capturing()):
        wp, scale, shift = hit[1]
    else:
        scale = shift = None
        if bn is not None:
            scale = torch.empty(g.Co, device=dev, dtype=torch.float32)
            shift = torch.empty(g.Co, device=dev, dtype=torch.float32)
            L.check(lib.dynmm_bn_fold(_p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                      _p(conv_bias), _p(scale), _p(shift), g.Co, bn.eps, st), 'bn_fold')
        else:
            shift = _chk(conv_bias, 'bias')
        if wino2:
            wp = torch.empty(lib.dynmm_wino2d_packed_floats(g.Co, g.Ci), device=dev, dtype=torch.float32)
            L.check(lib.dynmm_wino2d_pack(_p(weight), _p(wp), _p(scale), g.Co, g.Ci, 0, st), 'wino2d_pack')
            scale = None
        elif wino:
            # filter transforms of scale[co] * w: the folded BatchNorm factor rides in the operand, the kernel adds the shift
            wp = torch.empty(lib.dynmm_wino_packed_floats(g.Co, g.Ci, g.KH, g.KW), device=dev, dtype=torch.float32)
            L.check(lib.dynmm_wino_pack(_p(weight), _p(wp), _p(scale), g.Co, g.Ci, g.KH, g.KW, 0, st), 'wino_pack')
            scale = None
        else:
            wp = torch.empty(lib.dynmm_packed_weight_floats(g.Co, g.Ci, g.KH, g.KW, 0), device=dev, dtype=torch.float32)
            L.check(lib.dynmm_pack_weight(_p(weight), _p(wp), None, g.Co, g.Ci, g.KH, g.KW, st), 'pack_weight')
        if not torch.cuda.is_current_stream_capturing():
            setattr(weight, slot, (stamp, (wp, scale, shift)))
    y = torch.empty((g.N, g.Co, g.Ho, g.Wo), device=dev, dtype=torch.float32)
    if wino2:
        L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_wino2d_fwd(_p(x), _p(wp), _p(shift), _p(residual), _p(y), None, 0, C.byref(g),
                                                                     ACT[act], st), extra=int(residual is not None), wino=22),
                'conv2d_wino2d_fwd')
    elif wino:
        L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_wino_fwd(_p(x), _p(wp), _p(shift), _p(residual), _p(y), C.byref(g),
                                                                   ACT[act], st), extra=int(residual is not None), wino=True),
                'conv2d_wino_fwd')
    else:
        L.check(_timed('fwd', g, lambda: lib.dynmm_conv2d_fwd(_p(x), _p(x2), _p(wp), _p(scale), _p(shift), _p(residual),
                                                              _p(y), C.byref(g), ACT[act], st),
                       extra=int(residual is not None)), 'conv2d_fwd')
    return y


# ------------------------------------------------------------------------------------------------
# batch norm (+ residual + activation)
# ------------------------------------------------------------------------------------------------
# Zeroed fp64 accumulators for the per-channel BatchNorm reductions come from an arena that is cleared by ONE
# memset per step (begin_step(), called by the models at the start of a training forward) instead of one
# memset launch per BatchNorm call (~200 per step).  Slices are handed out by a bump pointer and never reused
# within a step; when the arena runs out (or begin_step() is never called) a call falls back to its own memset.
_ARENA = {'buf': None, 'off': 0, 'elems': 1 << 19}


def begin_step():
    """Start of a training step on the current stream: re-arm the zero arena.  Safe whenever no BatchNorm
    kernel of an earlier step is still in flight on another stream (the models join their side streams)."""
    a = _ARENA
    if a['buf'] is not None and a['off'] > 0:
        a['buf'][:a['off']].zero_()
    a['off'] = 0


def _zero_sums(n, device):
    """(tensor of n zero doubles, is_zero flag)."""
    a = _ARENA
    if a['buf'] is None or a['buf'].device != device:
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(n, device=device, dtype=torch.float64), 0
        a['buf'] = torch.zeros(a['elems'], device=device, dtype=torch.float64)
        a['off'] = 0
    if a['off'] + n > a['elems']:
        return torch.empty(n, device=device, dtype=torch.float64), 0
    t = a['buf'][a['off']:a['off'] + n]
    a['off'] += n
    return t, 1


class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, link, nbt, pre_sums=None,
                bwd_link=None):
        lib = _lib()
        st = _stream()
        x, residual = _chk(x, 'x'), _chk(residual, 'residual')
        _same_shape(x, residual, 'batch_norm residual')
        N, Cc, H, W = x.shape
        HW = H * W
        dev = x.device
        sums = None
        if training:
            note_mutation()          # running statistics / step counter are rewritten in place below
        if training and N * HW <= 1:
            raise ValueError(f'Expected more than 1 value per channel when training, got input size {tuple(x.shape)}')
        if training and pre_sums is not None:
            sums = pre_sums                      # left by the producing convolution's epilogue (conv2d(bn_stats=True))
        elif training:
            sums, zeroed = _zero_sums(2 * Cc, dev)
            L.check(lib.dynmm_bn_stats(_p(x), _p(sums), N, Cc, HW, zeroed, st), 'bn_stats')
        mean = torch.empty(Cc, device=dev, dtype=torch.float32)
        invstd = torch.empty(Cc, device=dev, dtype=torch.float32)
        y = torch.empty_like(x)
        # (training = the number of slabs the sums arrive in: 1, or the slots of the producing convolution's epilogue)
        nsl = (sums.numel() // (2 * Cc)) if training else 0
        # ReLU after a residual add, gradient recorded: the backward cannot re-derive the mask from x — the normalise pass leaves
        # the decisions as one bit per element and the two backward passes read those instead of y (csrc/norm.hip)
        bits = None
        if (BN_RELU_BITS and act == L.ACT_RELU and residual is not None and HW % 4 == 0 and
                any(ctx.needs_input_grad[i] for i in (0, 1, 2, 5)) and
                all(t.data_ptr() % 16 == 0 for t in (x, residual, y))):
            bits = torch.empty(lib.dynmm_bn_relu_bits_words(N, Cc, HW), device=dev, dtype=torch.int64)
        L.check(lib.dynmm_bn_apply(_p(x), _p(sums), _p(gamma), _p(beta), _p(running_mean),
                                   _p(running_var), _p(mean), _p(invstd), _p(residual), _p(y), _p(nbt),
                                   N, Cc, HW, eps, momentum, nsl, act, _p(bits), st), 'bn_apply')
        ctx.act = act
        ctx.training = training
        ctx.link = link
        ctx.bwd_link = None
        if bwd_link is not None and training and act == L.ACT_RELU and (residual is None or bits is not None):
            bwd_link.x, bwd_link.mean, bwd_link.invstd, bwd_link.gamma, bwd_link.beta = x, mean, invstd, gamma, beta
            bwd_link.bits = bits if residual is not None else None
            ctx.bwd_link = bwd_link
        ctx.has_res = residual is not None
        # ReLU without residual: the backward re-derives the mask from x (bit-identical to this forward's
        # fma) instead of reading y — one tensor read less in bn_bwd_reduce and in bn_bwd_apply
        need_y = act != L.ACT_NONE and not (act == L.ACT_RELU and residual is None) and bits is None
        ctx.save_for_backward(x, y if need_y else None, gamma, mean, invstd, beta, bits)
        ctx.g_param, ctx.b_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib()
        st = _stream()
        x, y, gamma, mean, invstd, beta, bits = ctx.saved_tensors
        gy = _chk(gy, 'grad')
        if bits is not None and gy.data_ptr() % 16 != 0:
            gy = gy.clone()                      # (a gradient view off the 16-byte grid: the bit path reads 16 bytes per lane)
        N, Cc, H, W = x.shape
        HW = H * W
        dev = x.device
        bl = ctx.bwd_link
        act = ctx.act
        if bl is not None and bl.sums is not None:
            sums, bl.sums = bl.sums, None          # left by the consumer convolution's input-gradient launch (BNLink)
            if bl.premasked:                       # ... which also applied the ReLU decisions to gy (BNRED == 2)
                act, bits, y = L.ACT_NONE, None, None
        else:
            sums, zeroed = _zero_sums(2 * Cc, dev)
            L.check(lib.dynmm_bn_bwd_reduce(_p(gy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums),
                                            N, Cc, HW, ctx.act, zeroed, _p(bits), st), 'bn_bwd_reduce')
        if bl is not None:
            bl.x = bl.bits = None
        dx = torch.empty_like(x)
        need_res = ctx.has_res and ctx.needs_input_grad[5]
        dres = torch.empty_like(x) if (need_res and act != L.ACT_NONE) else None
        dgamma, dgamma_ret = _grad_dst(ctx.g_param)
        dbeta, dbeta_ret = _grad_dst(ctx.b_param)
        L.check(lib.dynmm_bn_bwd_apply(_p(gy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums),
                                       _p(dx), _p(dres), _p(dgamma), _p(dbeta), N, Cc, HW,
                                       (sums.numel() // (2 * Cc)) if ctx.training else 0, act, _p(bits), st), 'bn_bwd_apply')
        if need_res and dres is None:
            dres = gy            # no activation: the residual branch receives the gradient unchanged
        if ctx.link is not None and dres is not None:
            ctx.link.dres, dres = dres, None      # absorbed by the first conv's dgrad epilogue
        _grads_enqueued()
        return dx, dgamma_ret, dbeta_ret, None, None, dres, None, None, None, None, None, None, None, None


def batch_norm_act(x, bn, act=None, residual=None, training=None, link=None, bwd_link=None):
    """act(BatchNorm2d(x) + residual) using the parameters/buffers of the nn.BatchNorm2d `bn`.
    `link`: GradLink that carries the residual's gradient to the op that consumes the same tensor."""
    training = bn.training if training is None else training
    nbt = bn.num_batches_tracked if training else None       # incremented inside the normalise kernel
    if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
        raise L.DynmmHipError('BatchNorm num_batches_tracked must be an int64 tensor on the HIP device')
    pre = getattr(x, '_bn_sums', None) if training else None
    if pre is not None and (pre.numel() % (2 * x.shape[1]) != 0 or pre.numel() == 0):
        pre = None
    y = _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual,
                            bool(training), float(bn.momentum), float(bn.eps), ACT[act], link, nbt, pre, bwd_link)
    if ACT_TRACE is not None and ACT[act] == L.ACT_RELU:
        ACT_TRACE.append(y.detach())
    return y


# ------------------------------------------------------------------------------------------------
# pooling / resampling
# ------------------------------------------------------------------------------------------------
class _MaxPool3x3s2(Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib()
        x = _chk(x, 'x')
        N, Cc, H, W = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty((N, Cc, Ho, Wo), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Cc, Ho, Wo), device=x.device, dtype=torch.int8)
        L.check(lib.dynmm_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), N, Cc, H, W, Ho, Wo, _stream()), 'maxpool_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (N, Cc, H, W, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib()
        (idx,) = ctx.saved_tensors
        N, Cc, H, W, Ho, Wo = ctx.shape
        gy = _chk(gy, 'grad')
        dx = torch.empty((N, Cc, H, W), device=gy.device, dtype=torch.float32)
        L.check(lib.dynmm_maxpool3x3s2_bwd(_p(gy), _p(idx), _p(dx), N, Cc, H, W, Ho, Wo, _stream()), 'maxpool_bwd')
        return dx


def max_pool_3x3_s2(x):
    return _MaxPool3x3s2.apply(x)


class _AdaptiveAvgPool(Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        lib = _lib()
        x = _chk(x, 'x')
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, oh, ow), device=x.device, dtype=torch.float32)
        if oh == 1 and ow == 1:   # global pool: one workgroup per plane
            L.check(lib.dynmm_gap2_fwd(_p(x), None, _p(y), None, N * Cc, H * W, _stream()), 'gap_fwd')
        else:
            L.check(lib.dynmm_adaptive_avgpool_fwd(_p(x), _p(y), N * Cc, H, W, oh, ow, _stream()), 'aap_fwd')
        ctx.shape = (N, Cc, H, W, oh, ow)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib()
        N, Cc, H, W, oh, ow = ctx.shape
        gy = _chk(gy, 'grad')
        dx = torch.empty((N, Cc, H, W), device=gy.device, dtype=torch.float32)
        L.check(lib.dynmm_adaptive_avgpool_bwd(_p(gy), _p(dx), N * Cc, H, W, oh, ow, _stream()), 'aap_bwd')
        return dx, None, None


def adaptive_avg_pool(x, out_hw):
    oh, ow = _pair(out_hw)
    return _AdaptiveAvgPool.apply(x, oh, ow)


class _NearestConcat(Function):
    """cat([x, nearest(y_1, size(x)), nearest(y_2, size(x)), ...], dim=1)."""

    @staticmethod
    def forward(ctx, x, *ys):
        lib = _lib()
        st = _stream()
        x = _chk(x, 'x')
        ys = [_chk(y, 'y') for y in ys]
        N, C0, H, W = x.shape
        Ctot = C0 + sum(y.shape[1] for y in ys)
        out = torch.empty((N, Ctot, H, W), device=x.device, dtype=torch.float32)
        off = 0
        ctx.parts = []
        for t in [x] + ys:
            _, Cc, h, w = t.shape
            L.check(lib.dynmm_nearest_into_fwd(_p(t), _p(out), N, Cc, h, w, Ctot, off, H, W, st), 'nearest_into_fwd')
            ctx.parts.append((Cc, h, w, off))
            off += Cc
        ctx.dims = (N, Ctot, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        st = _stream()
        g = _chk(g, 'grad')
        N, Ctot, H, W = ctx.dims
        outs = []
        for i, (Cc, h, w, off) in enumerate(ctx.parts):
            if not ctx.needs_input_grad[i]:
                outs.append(None)
                continue
            d = torch.empty((N, Cc, h, w), device=g.device, dtype=torch.float32)
            L.check(lib.dynmm_nearest_into_bwd(_p(g), _p(d), N, Cc, h, w, Ctot, off, H, W, st), 'nearest_into_bwd')
            outs.append(d)
        return tuple(outs)


def nearest_concat(x, *ys):
    return _NearestConcat.apply(x, *ys)


class _Upsample2xDw(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, skip):
        lib = _lib()
        x, weight, bias, skip = _chk(x, 'x'), _chk(weight, 'weight'), _chk(bias, 'bias'), _chk(skip, 'skip')
        N, Cc, H, W = x.shape
        if skip is not None and tuple(skip.shape) != (N, Cc, 2 * H, 2 * W):
            raise L.DynmmHipError(f'upsample skip connection {tuple(skip.shape)} does not match the upsampled '
                                  f'feature map {(N, Cc, 2 * H, 2 * W)} (input H, W must be multiples of 32)')
        y = torch.empty((N, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        L.check(lib.dynmm_upsample2x_dw3x3_fwd(_p(x), _p(weight), _p(bias), _p(skip), _p(y), N, Cc, H, W,
                                               _stream()), 'upsample_fwd')
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.has_skip = skip is not None
        ctx.w_param, ctx.b_param = weight, bias
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        x, weight = ctx.saved_tensors
        g = _chk(g, 'grad')
        N, Cc, H, W = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = dw_ret = db = db_ret = None
        if ctx.needs_input_grad[1]:
            dw, dw_ret = _grad_dst(ctx.w_param)
            if ctx.has_bias:
                db, db_ret = _grad_dst(ctx.b_param)
        ws = None
        if dw is not None:
            nb = lib.dynmm_upsample2x_dw3x3_bwd_workspace_bytes(N, Cc)
            ws = torch.empty(max(nb // 4, 1), device=g.device, dtype=torch.float32)
        L.check(lib.dynmm_upsample2x_dw3x3_bwd(_p(g), _p(x), _p(weight), _p(dx), _p(dw), _p(db), _p(ws), N, Cc, H, W,
                                               _stream()), 'upsample_bwd')
        _grads_enqueued()
        return dx, dw_ret, db_ret, (g if ctx.has_skip else None)


def upsample2x_dw3x3(x, weight, bias, skip=None):
    """Learned 2x upsample (nearest + depthwise 3x3 + bias) fused with the decoder's skip add."""
    return _Upsample2xDw.apply(x, weight, bias, skip)


# ------------------------------------------------------------------------------------------------
# SE fusion + gated blend
# ------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class _SEFuseBlend(Function):
    """out = wc*rgb + (1-wc)*(SE_rgb(rgb) + SE_depth(depth))   ('add' mode: SE = identity).
    wc = wcum[:, col] (per-sample scalar) or 0 when wcum is None.

    Compacted form (K16): `depth` holds only the first n_act samples of the (branch-sorted) batch; the fusion is
    evaluated on that prefix and out[n_act:] = rgb[n_act:] (samples that skip this stage).  `inplace` (inference
    only) writes the prefix into rgb's own storage, so skipped samples cost no memory traffic at all."""

    @staticmethod
    def forward(ctx, rgb, depth, wcum, col, use_se, inplace, *params):
        lib = _lib()
        st = _stream()
        rgb, depth = _chk(rgb, 'rgb'), _chk(depth, 'depth')
        N, Cc, H, W = rgb.shape
        n_act = depth.shape[0]
        if n_act > N or tuple(depth.shape[1:]) != tuple(rgb.shape[1:]) or n_act <= 0:
            raise L.DynmmHipError(f'rgb/depth fusion: shape mismatch {tuple(rgb.shape)} vs {tuple(depth.shape)}')
        HW = H * W
        dev = rgb.device
        f32 = dict(device=dev, dtype=torch.float32)
        wc_ptr, wc_stride = None, 0
        if wcum is not None:
            wcum = _chk(wcum, 'wcum')
            wc_stride = wcum.shape[1]
            wc_ptr = wcum.data_ptr() + 4 * col
        sr = sd = hr = hd = gr = gd = None
        parr = None
        if use_se:
            params = [_chk(p, 'se param') for p in params]
            parr = _ptr_array(params)
            sr, sd = torch.empty((n_act, Cc), **f32), torch.empty((n_act, Cc), **f32)
            L.check(lib.dynmm_gap2_fwd(_p(rgb), _p(depth), _p(sr), _p(sd), n_act * Cc, HW, st), 'gap2')
            hr, hd = torch.empty((n_act, Cc // 16), **f32), torch.empty((n_act, Cc // 16), **f32)
            gr, gd = torch.empty((n_act, Cc), **f32), torch.empty((n_act, Cc), **f32)
        a, b = torch.empty((n_act, Cc), **f32), torch.empty((n_act, Cc), **f32)
        L.check(lib.dynmm_se_coeff_fwd(_p(sr), _p(sd), parr, wc_ptr, wc_stride, _p(a), _p(b),
                                       _p(hr), _p(hd), _p(gr), _p(gd), n_act, Cc, int(use_se), st), 'se_coeff_fwd')
        if inplace and n_act < N:
            out = rgb                                   # prefix overwritten below, tail untouched
            ctx.mark_dirty(rgb)
        else:
            out = torch.empty_like(rgb)
            _copy_rows(rgb, out, n_act, N - n_act)
        L.check(lib.dynmm_axpby_fwd(_p(rgb), _p(depth), _p(a), _p(b), _p(out), n_act * Cc, HW, st), 'axpby_fwd')
        ctx.use_se = use_se
        ctx.col = col
        ctx.n_params = len(params)
        ctx.param_objs = list(params)
        ctx.save_for_backward(rgb, depth, wcum, a, b, sr, sd, hr, hd, gr, gd, *params)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        st = _stream()
        rgb, depth, wcum, a, b, sr, sd, hr, hd, gr, gd = ctx.saved_tensors[:11]
        params = list(ctx.saved_tensors[11:])
        g = _chk(g, 'grad')
        N, Cc, H, W = rgb.shape
        n_act = depth.shape[0]
        HW = H * W
        f32 = dict(device=g.device, dtype=torch.float32)
        da, db = torch.empty((n_act, Cc), **f32), torch.empty((n_act, Cc), **f32)
        L.check(lib.dynmm_axpby_bwd_reduce(_p(g), _p(rgb), _p(depth), _p(da), _p(db), n_act * Cc, HW, st), 'axpby_bwd_reduce')
        dparams = dparams_ret = [None] * ctx.n_params
        dsr = dsd = None
        parr = dparr = None
        if ctx.use_se:
            pairs = [_grad_dst(po) for po in ctx.param_objs]     # straight into the flat .grad views when possible
            dparams, dparams_ret = [d for d, _ in pairs], [r for _, r in pairs]
            parr, dparr = _ptr_array(params), _ptr_array(dparams)
            dsr, dsd = torch.empty((n_act, Cc), **f32), torch.empty((n_act, Cc), **f32)
        dwcum = None
        wc_ptr, wc_stride, dwc_ptr = None, 0, None
        if wcum is not None:
            wc_stride = wcum.shape[1]
            wc_ptr = wcum.data_ptr() + 4 * ctx.col
            if ctx.needs_input_grad[2]:
                dwcum = torch.zeros_like(wcum)           # samples past n_act: no gate gradient from this stage
                dwc_ptr = dwcum.data_ptr() + 4 * ctx.col
        ws = torch.empty(lib.dynmm_se_coeff_bwd_workspace_bytes(n_act, Cc) // 4, **f32) if ctx.use_se else None
        L.check(lib.dynmm_se_coeff_bwd(_p(da), _p(db), _p(sr), _p(sd), parr, wc_ptr, wc_stride,
                                       _p(hr), _p(hd), _p(gr), _p(gd), dparr, _p(dsr), _p(dsd),
                                       dwc_ptr, wc_stride, _p(ws), n_act, Cc, int(ctx.use_se), st), 'se_coeff_bwd')
        drgb, ddepth = torch.empty_like(rgb), torch.empty_like(depth)
        L.check(lib.dynmm_axpby_bwd_apply(_p(g), _p(a), _p(b), _p(dsr), _p(dsd), 1.0 / HW,
                                          _p(drgb), _p(ddepth), n_act * Cc, HW, st), 'axpby_bwd_apply')
        _copy_rows(g, drgb, n_act, N - n_act)            # skipped samples: out = rgb
        _grads_enqueued()
        return (drgb, ddepth, dwcum, None, None, None, *dparams_ret)


class _SEFusePool(Function):
    """(max_pool(SE_rgb(rgb) + SE_depth(depth)), max_pool(depth)) — the stem fusion and both 3x3/s2 max-pools of
    …globalgate.py:258-261 without ever writing the full-resolution fused map (csrc/pointwise.hip: axpby_pool_*)."""

    @staticmethod
    def forward(ctx, rgb, depth, use_se, *params):
        lib = _lib()
        st = _stream()
        rgb, depth = _chk(rgb, 'rgb'), _chk(depth, 'depth')
        _same_shape(rgb, depth, 'stem fusion')
        N, Cc, H, W = rgb.shape
        HW = H * W
        f32 = dict(device=rgb.device, dtype=torch.float32)
        sr = sd = hr = hd = gr = gd = None
        parr = None
        if use_se:
            params = [_chk(p, 'se param') for p in params]
            parr = _ptr_array(params)
            sr, sd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
            L.check(lib.dynmm_gap2_fwd(_p(rgb), _p(depth), _p(sr), _p(sd), N * Cc, HW, st), 'gap2')
            hr, hd = torch.empty((N, Cc // 16), **f32), torch.empty((N, Cc // 16), **f32)
            gr, gd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        a, b = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        L.check(lib.dynmm_se_coeff_fwd(_p(sr), _p(sd), parr, None, 0, _p(a), _p(b), _p(hr), _p(hd), _p(gr), _p(gd),
                                       N, Cc, int(use_se), st), 'se_coeff_fwd')
        Ho, Wo = H // 2, W // 2
        yo, yd = torch.empty((N, Cc, Ho, Wo), **f32), torch.empty((N, Cc, Ho, Wo), **f32)
        io = torch.empty((N, Cc, Ho, Wo), device=rgb.device, dtype=torch.int8)
        idd = torch.empty_like(io)
        L.check(lib.dynmm_axpby_pool_fwd(_p(rgb), _p(depth), _p(a), _p(b), _p(yo), io.data_ptr(), _p(yd), idd.data_ptr(),
                                         None, Cc, N * Cc, H, W, st), 'axpby_pool_fwd')
        ctx.use_se = use_se
        ctx.n_params = len(params)
        ctx.param_objs = list(params)
        ctx.save_for_backward(rgb, depth, a, b, sr, sd, hr, hd, gr, gd, io, idd, *params)
        return yo, yd

    @staticmethod
    def backward(ctx, g_o, g_d):
        lib = _lib()
        st = _stream()
        rgb, depth, a, b, sr, sd, hr, hd, gr, gd, io, idd = ctx.saved_tensors[:12]
        params = list(ctx.saved_tensors[12:])
        N, Cc, H, W = rgb.shape
        HW = H * W
        f32 = dict(device=rgb.device, dtype=torch.float32)
        g_o = torch.zeros_like(io, dtype=torch.float32) if g_o is None else _chk(g_o, 'grad')
        g_d = torch.zeros_like(io, dtype=torch.float32) if g_d is None else _chk(g_d, 'grad')
        da, db = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        L.check(lib.dynmm_axpby_pool_bwd_reduce(_p(g_o), io.data_ptr(), _p(rgb), _p(depth), _p(da), _p(db), None, Cc,
                                                N * Cc, H, W, st), 'axpby_pool_bwd_reduce')
        dparams = dparams_ret = [None] * ctx.n_params
        dsr = dsd = None
        parr = dparr = None
        if ctx.use_se:
            pairs = [_grad_dst(po) for po in ctx.param_objs]
            dparams, dparams_ret = [d for d, _ in pairs], [r for _, r in pairs]
            parr, dparr = _ptr_array(params), _ptr_array(dparams)
            dsr, dsd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        ws = torch.empty(lib.dynmm_se_coeff_bwd_workspace_bytes(N, Cc) // 4, **f32) if ctx.use_se else None
        L.check(lib.dynmm_se_coeff_bwd(_p(da), _p(db), _p(sr), _p(sd), parr, None, 0, _p(hr), _p(hd), _p(gr), _p(gd),
                                       dparr, _p(dsr), _p(dsd), None, 0, _p(ws), N, Cc, int(ctx.use_se), st), 'se_coeff_bwd')
        drgb, ddepth = torch.empty_like(rgb), torch.empty_like(depth)
        L.check(lib.dynmm_axpby_pool_bwd_apply(_p(g_o), io.data_ptr(), _p(g_d), idd.data_ptr(), _p(a), _p(b), _p(dsr), _p(dsd),
                                               1.0 / HW, _p(drgb), _p(ddepth), N * Cc, H, W, st), 'axpby_pool_bwd_apply')
        _grads_enqueued()
        return (drgb, ddepth, None, *dparams_ret)


class _StemBNFusePool(Function):
    """Training form of the whole stem tail (resnet.py:229-231 BN + ReLU of both stems, …globalgate.py:258-261):
        (max_pool(SE_rgb(y_r) + SE_depth(y_d)), max_pool(y_d)),   y = relu(batch_norm(x))
    from the two stem CONV outputs x_r, x_d.  The normalised tensors y (2 x 629 MB at batch 32) are never written:
    batch statistics -> dynmm_bn_finalize (mean / invstd / running statistics / scale, shift), then the squeeze and
    the fused blend + pooling kernels apply relu(fma(x, scale, shift)) on load.  The backward first forms the
    gradients of y (axpby_pool_bwd_*), then runs the ordinary BatchNorm backward kernels, which re-derive the ReLU
    mask from x themselves."""

    @staticmethod
    def forward(ctx, xr, xd, gam_r, bet_r, rm_r, rv_r, nbt_r, gam_d, bet_d, rm_d, rv_d, nbt_d, mom_r, eps_r, mom_d,
                eps_d, use_se, pre_r, pre_d, slots, *params):
        lib = _lib()
        st = _stream()
        xr, xd = _chk(xr, 'x_rgb'), _chk(xd, 'x_depth')
        _same_shape(xr, xd, 'stem fusion')
        N, Cc, H, W = xr.shape
        HW = H * W
        dev = xr.device
        f32 = dict(device=dev, dtype=torch.float32)
        if N * HW <= 1:
            raise ValueError(f'Expected more than 1 value per channel when training, got input size {tuple(xr.shape)}')
        note_mutation()
        tr = torch.empty((4, Cc), **f32)                 # scale_r, shift_r, scale_d, shift_d
        stats = torch.empty((4, Cc), **f32)              # mean_r, invstd_r, mean_d, invstd_d
        for k, (x, gam, bet, rm, rv, nbt, mom, eps, pre) in enumerate(((xr, gam_r, bet_r, rm_r, rv_r, nbt_r, mom_r, eps_r, pre_r),
                                                                       (xd, gam_d, bet_d, rm_d, rv_d, nbt_d, mom_d, eps_d, pre_d))):
            if pre is not None:                  # left by the stem convolution's epilogue (dynmm_conv2d_stem_fwd_stats)
                sums = pre
            else:
                sums, zeroed = _zero_sums(2 * Cc, dev)
                L.check(lib.dynmm_bn_stats(_p(x), _p(sums), N, Cc, HW, zeroed, st), 'bn_stats')
            L.check(lib.dynmm_bn_finalize(_p(sums), _p(gam), _p(bet), _p(rm), _p(rv), _p(stats[2 * k]), _p(stats[2 * k + 1]),
                                          _p(nbt), _p(tr[2 * k]), _p(tr[2 * k + 1]), N, Cc, HW, eps, mom, st), 'bn_finalize')
        sr = sd = hr = hd = gr = gd = None
        parr = None
        if use_se:
            params = [_chk(p, 'se param') for p in params]
            parr = _ptr_array(params)
            sr, sd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
            L.check(lib.dynmm_gap2_bnrelu_fwd(_p(xr), _p(xd), _p(tr), Cc, _p(sr), _p(sd), N * Cc, HW, st), 'gap2_bnrelu')
            hr, hd = torch.empty((N, Cc // 16), **f32), torch.empty((N, Cc // 16), **f32)
            gr, gd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        a, b = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        L.check(lib.dynmm_se_coeff_fwd(_p(sr), _p(sd), parr, None, 0, _p(a), _p(b), _p(hr), _p(hd), _p(gr), _p(gd),
                                       N, Cc, int(use_se), st), 'se_coeff_fwd')
        Ho, Wo = H // 2, W // 2
        yo, yd = torch.empty((N, Cc, Ho, Wo), **f32), torch.empty((N, Cc, Ho, Wo), **f32)
        io = torch.empty((N, Cc, Ho, Wo), device=dev, dtype=torch.int8)
        idd = torch.empty_like(io)
        L.check(lib.dynmm_axpby_pool_fwd(_p(xr), _p(xd), _p(a), _p(b), _p(yo), io.data_ptr(), _p(yd), idd.data_ptr(),
                                         _p(tr), Cc, N * Cc, H, W, st), 'axpby_pool_fwd')
        ctx.use_se = use_se
        ctx.n_params = len(params)
        ctx.param_objs = list(params)
        ctx.bn_params = (gam_r, bet_r, gam_d, bet_d)
        ctx.slots = slots
        ctx.save_for_backward(xr, xd, tr, stats, a, b, sr, sd, hr, hd, gr, gd, io, idd, gam_r, bet_r, gam_d, bet_d, *params)
        return yo, yd

    @staticmethod
    def backward(ctx, g_o, g_d):
        lib = _lib()
        st = _stream()
        xr, xd, tr, stats, a, b, sr, sd, hr, hd, gr, gd, io, idd, gam_r, bet_r, gam_d, bet_d = ctx.saved_tensors[:18]
        params = list(ctx.saved_tensors[18:])
        N, Cc, H, W = xr.shape
        HW = H * W
        dev = xr.device
        f32 = dict(device=dev, dtype=torch.float32)
        g_o = torch.zeros_like(io, dtype=torch.float32) if g_o is None else _chk(g_o, 'grad')
        g_d = torch.zeros_like(io, dtype=torch.float32) if g_d is None else _chk(g_d, 'grad')
        da, db = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        L.check(lib.dynmm_axpby_pool_bwd_reduce(_p(g_o), io.data_ptr(), _p(xr), _p(xd), _p(da), _p(db), _p(tr), Cc,
                                                N * Cc, H, W, st), 'axpby_pool_bwd_reduce')
        dparams = dparams_ret = [None] * ctx.n_params
        dsr = dsd = None
        parr = dparr = None
        if ctx.use_se:
            pairs = [_grad_dst(po) for po in ctx.param_objs]
            dparams, dparams_ret = [d for d, _ in pairs], [r for _, r in pairs]
            parr, dparr = _ptr_array(params), _ptr_array(dparams)
            dsr, dsd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        ws = torch.empty(lib.dynmm_se_coeff_bwd_workspace_bytes(N, Cc) // 4, **f32) if ctx.use_se else None
        L.check(lib.dynmm_se_coeff_bwd(_p(da), _p(db), _p(sr), _p(sd), parr, None, 0, _p(hr), _p(hd), _p(gr), _p(gd),
                                       dparr, _p(dsr), _p(dsd), None, 0, _p(ws), N, Cc, int(ctx.use_se), st), 'se_coeff_bwd')
        _grads_enqueued()        # the SE parameters' gradients are complete HERE, on this stream: reported now, not by whichever
                                 # op next calls _grads_enqueued (the deferred branch below returns without one: ADVICE r5)
        # BatchNorm backward of both stems with their incoming gradient derived on the fly from the pooled gradients
        # (dynmm_stem_bn_bwd_*): gy_rgb = a * d(fuse) + dsr/HW ; gy_depth = b * d(fuse) + dsd/HW + d(pooled depth)
        def bn_chain(k):
            x, coef, off, gdp, idp, gam, bet, gp, bp = (
                (xr, a, dsr, None, None, gam_r, bet_r, ctx.bn_params[0], ctx.bn_params[1]),
                (xd, b, dsd, g_d, idd, gam_d, bet_d, ctx.bn_params[2], ctx.bn_params[3]))[k]
            stq = _stream()
            mean, invstd = stats[2 * k], stats[2 * k + 1]
            sums, zeroed = _zero_sums(2 * Cc, dev)
            head = (_p(g_o), io.data_ptr(), _p(gdp), None if idp is None else idp.data_ptr(), _p(coef), _p(off), 1.0 / HW,
                    _p(x), _p(mean), _p(invstd), _p(gam), _p(bet))
            L.check(lib.dynmm_stem_bn_bwd_reduce(*head, _p(sums), N, Cc, H, W, zeroed, stq), 'stem_bn_bwd_reduce')
            dgamma, dgamma_ret = _grad_dst(gp)
            dbeta, dbeta_ret = _grad_dst(bp)
            dx = torch.empty_like(x)
            L.check(lib.dynmm_stem_bn_bwd_apply(*head, _p(sums), _p(dx), _p(dgamma), _p(dbeta), N, Cc, H, W, stq),
                    'stem_bn_bwd_apply')
            _grads_enqueued()
            return dx, dgamma_ret, dbeta_ret
        if ctx.slots is not None:
            # Each stem's BatchNorm backward is left to that stem's _StemBNDeferred node (stem_bn_defer), which autograd runs
            # immediately before the stem convolution's own backward: the first stem's weight gradient is then already on its
            # stream while the second stem's two BatchNorm passes (1.6 GB of traffic) run.  What travels to those nodes as
            # "gradient" is a zero-stride placeholder of the right shape.
            for k in (0, 1):
                ctx.slots[k]['bwd'] = (lambda k=k: bn_chain(k))
            ph = torch.empty(1, **f32)
            return (ph.expand(xr.shape), ph.expand(xd.shape), *([None] * 18), *dparams_ret)
        (dxr, dgr, dbr), (dxd, dgd, dbd) = bn_chain(0), bn_chain(1)
        return (dxr, dxd, dgr, dbr, None, None, None, dgd, dbd, None, None, None, None, None, None, None, None, None, None, None,
                *dparams_ret)


class _StemBNDeferred(Function):
    """Identity on a stem convolution's output, placed right after that convolution (stem_bn_defer): in the backward it runs the
    stem's BatchNorm backward that _StemBNFusePool.backward prepared and left in `slot`.  Autograd executes ready nodes latest
    created first, so the order of the forward — conv A, defer A, conv B, defer B, fuse — makes the backward
    fuse -> BN backward B -> weight gradient B (asynchronous, matrix cores) -> BN backward A (HBM) -> weight gradient A:
    the 1.0 ms of BatchNorm passes at the very end of the step, where nothing else is left to run, overlap a weight gradient
    instead of preceding both."""

    @staticmethod
    def forward(ctx, x, gam, bet, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        fn = ctx.slot.pop('bwd', None)
        if fn is None:
            # nothing was left here: the consumer ran the BatchNorm backward itself (only ONE of the two stems was deferred, or
            # the tensor went elsewhere) and `g` is the real gradient
            return g, None, None, None
        dx, dgam, dbet = fn()
        return dx, dgam, dbet, None


def stem_bn_defer(x, bn):
    """see _StemBNDeferred; call it on a stem convolution's output right after that convolution, before the other stem's"""
    if not (torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad)):
        return x
    slot = {}
    y = _StemBNDeferred.apply(x, bn.weight, bn.bias, slot)
    y._stem_slot = slot
    for attr in ('_bn_sums',):
        if hasattr(x, attr):
            setattr(y, attr, getattr(x, attr))
    return y


def stem_bn_fuse_pool(x_rgb, bn_rgb, x_depth, bn_depth, se_params=None):
    """see _StemBNFusePool; bn_* are the stems' nn.BatchNorm2d modules (training mode)."""
    use_se = se_params is not None
    for bn in (bn_rgb, bn_depth):
        nbt = bn.num_batches_tracked
        if nbt.dtype != torch.int64 or not nbt.is_cuda:
            raise L.DynmmHipError('BatchNorm num_batches_tracked must be an int64 tensor on the HIP device')
    pre = []
    for x in (x_rgb, x_depth):          # conv2d(..., bn_stats=True) hangs the statistics on its output (one slab of [2][C] doubles)
        p = getattr(x, '_bn_sums', None)
        pre.append(p if (p is not None and p.numel() == 2 * x.shape[1]) else None)
    slots = (getattr(x_rgb, '_stem_slot', None), getattr(x_depth, '_stem_slot', None))
    slots = slots if (slots[0] is not None and slots[1] is not None) else None       # (both stems deferred, or neither)
    return _StemBNFusePool.apply(x_rgb, x_depth, bn_rgb.weight, bn_rgb.bias, bn_rgb.running_mean, bn_rgb.running_var,
                                 bn_rgb.num_batches_tracked, bn_depth.weight, bn_depth.bias, bn_depth.running_mean,
                                 bn_depth.running_var, bn_depth.num_batches_tracked, float(bn_rgb.momentum),
                                 float(bn_rgb.eps), float(bn_depth.momentum), float(bn_depth.eps), use_se, pre[0], pre[1], slots,
                                 *(tuple(se_params) if use_se else ()))


_FUSED_STEM_POOL = True        # (module attributes: tests compare the fused stem with the unfused ops)
_FUSED_STEM_BN = True


def stem_bn_fuse_supported(h, w, bn_a, bn_b):
    """h, w: spatial size of the stem conv outputs"""
    return _FUSED_STEM_BN and _FUSED_STEM_POOL and bool(_lib().dynmm_axpby_pool_supported(int(h), int(w))) and \
        bn_a.training and bn_b.training and torch.is_grad_enabled() and bn_a.momentum is not None and \
        bn_b.momentum is not None


def se_fuse_pool_supported(x):
    """even H, W % 8 == 0 (the kernels' 16-byte row accesses); otherwise callers compose the unfused ops"""
    return _FUSED_STEM_POOL and x.dim() == 4 and bool(_lib().dynmm_axpby_pool_supported(int(x.shape[2]), int(x.shape[3])))


def se_fuse_pool(rgb, depth, se_params=None):
    """(max_pool_3x3_s2(se_fuse_blend(rgb, depth, se_params)), max_pool_3x3_s2(depth)) in one forward pass."""
    use_se = se_params is not None
    return _SEFusePool.apply(rgb, depth, use_se, *(tuple(se_params) if use_se else ()))


def se_fuse_blend(rgb, depth, se_params=None, wcum=None, col=0, inplace=False):
    """se_params: None ('add' fusion) or the 8 tensors (W1r,b1r,W2r,b2r,W1d,b1d,W2d,b2d).
    depth may hold only a PREFIX of rgb's batch (compaction): the remaining samples pass rgb through."""
    use_se = se_params is not None
    params = tuple(se_params) if use_se else ()
    return _SEFuseBlend.apply(rgb, depth, wcum, col, use_se, bool(inplace) and not torch.is_grad_enabled(), *params)


# ------------------------------------------------------------------------------------------------
# SkipESANet: per-stage Gumbel gate + 2-way blend
# ------------------------------------------------------------------------------------------------
_PHILOX_OFFSET = [0]          # advanced once per gate evaluation that draws its own noise


def manual_seed(seed):
    """Seed of the Philox stream the Gumbel gates draw from (device-side RNG; rgb_depth_fusion.py:50,56
    draws from torch's global generator — parity with the reference is distributional)."""
    global _PHILOX_SEED
    _PHILOX_SEED = int(seed) & 0xFFFFFFFFFFFFFFFF
    _PHILOX_OFFSET[0] = 0


_PHILOX_SEED = 0x5EEDD1CE


class _ReweighFuse(Function):
    """out   = rgb | rgb+depth | w0*rgb + w1*(rgb+depth)          (blend_mode 0 | 1 | 2, w = wblend[N,2])
    wnext = SqueezeAndExciteReweigh(rgb, depth; temp, hard, prev)  (only when `gate`), [N,2]
    aux   = [N,6] saved gate internals {w, ysoft0, ysoft1, y1, E0, E1} (not differentiable)."""

    @staticmethod
    def forward(ctx, rgb, depth, wblend, prev, noise, blend_mode, gate, temp, hard, *params):
        lib = _lib()
        st = _stream()
        rgb, depth = _chk(rgb, 'rgb'), _chk(depth, 'depth')
        _same_shape(rgb, depth, 'rgb/depth fusion')
        N, Cc, H, W = rgb.shape
        HW = H * W
        f32 = dict(device=rgb.device, dtype=torch.float32)
        wblend, prev, noise = _chk(wblend, 'wblend'), _chk(prev, 'prev'), _chk(noise, 'noise')
        if blend_mode == 2 and (wblend is None or wblend.numel() != 2 * N):
            raise L.DynmmHipError('blend_mode 2 needs wblend[N,2]')
        sr = sd = h = gg = aux = wnext = parr = None
        seed = offset = 0
        if gate:
            params = [_chk(p_, 'gate param') for p_ in params]
            parr = _ptr_array(params)
            sr, sd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
            L.check(lib.dynmm_gap2_fwd(_p(rgb), _p(depth), _p(sr), _p(sd), N * Cc, HW, st), 'gap2')
            h, gg = torch.empty((N, 2 * Cc // 16), **f32), torch.empty((N, 2 * Cc), **f32)
            aux, wnext = torch.empty((N, 6), **f32), torch.empty((N, 2), **f32)
            if noise is None:
                seed, offset = _PHILOX_SEED, _PHILOX_OFFSET[0]
                _PHILOX_OFFSET[0] += 1
        a, b = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
        L.check(lib.dynmm_reweigh_fwd(_p(sr), _p(sd), parr, _p(wblend), int(blend_mode), _p(prev), 1, _p(noise),
                                      seed, offset, float(temp), int(bool(hard)), _p(a), _p(b), _p(wnext),
                                      _p(h), _p(gg), _p(aux), N, Cc, st), 'reweigh_fwd')
        out = torch.empty_like(rgb)
        L.check(lib.dynmm_axpby_fwd(_p(rgb), _p(depth), _p(a), _p(b), _p(out), N * Cc, HW, st), 'axpby_fwd')
        ctx.cfg = (int(blend_mode), bool(gate), float(temp), len(params))
        ctx.param_objs = list(params)
        ctx.save_for_backward(rgb, depth, wblend, prev, a, b, sr, sd, h, gg, aux, *params)
        if gate:
            ctx.mark_non_differentiable(aux)
        return out, wnext, aux

    @staticmethod
    def backward(ctx, g, d_wnext, _d_aux):
        lib = _lib()
        st = _stream()
        rgb, depth, wblend, prev, a, b, sr, sd, h, gg, aux = ctx.saved_tensors[:11]
        params = list(ctx.saved_tensors[11:])
        blend_mode, gate, temp, n_params = ctx.cfg
        N, Cc, H, W = rgb.shape
        HW = H * W
        f32 = dict(device=rgb.device, dtype=torch.float32)
        if g is None:
            g = torch.zeros_like(rgb)
        g = _chk(g, 'grad')
        need_wb = blend_mode == 2 and ctx.needs_input_grad[2]
        da = db = d_wblend = None
        if need_wb:
            da, db = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
            L.check(lib.dynmm_axpby_bwd_reduce(_p(g), _p(rgb), _p(depth), _p(da), _p(db), N * Cc, HW, st),
                    'axpby_bwd_reduce')
            d_wblend = torch.empty((N, 2), **f32)
        dparams = dparams_ret = [None] * n_params
        dsr = dsd = d_prev = parr = dparr = None
        gate_bwd = gate and d_wnext is not None
        if gate_bwd:
            d_wnext = _chk(d_wnext, 'd_wnext')
            pairs = [_grad_dst(po) for po in ctx.param_objs]
            dparams, dparams_ret = [d for d, _ in pairs], [r for _, r in pairs]
            parr, dparr = _ptr_array(params), _ptr_array(dparams)
            dsr, dsd = torch.empty((N, Cc), **f32), torch.empty((N, Cc), **f32)
            if prev is not None and ctx.needs_input_grad[3]:
                d_prev = torch.empty((N,), **f32)
        if need_wb or gate_bwd:
            ws = torch.empty(lib.dynmm_reweigh_bwd_workspace_bytes(N, Cc) // 4, **f32) if gate_bwd else None
            L.check(lib.dynmm_reweigh_bwd(_p(d_wnext) if gate_bwd else None, _p(da), _p(db), _p(sr), _p(sd), parr,
                                          _p(prev), 1, _p(h), _p(gg), _p(aux), dparr, _p(dsr), _p(dsd),
                                          _p(d_wblend), _p(d_prev), _p(ws), temp, N, Cc, st), 'reweigh_bwd')
        drgb, ddepth = torch.empty_like(rgb), torch.empty_like(depth)
        L.check(lib.dynmm_axpby_bwd_apply(_p(g), _p(a), _p(b), _p(dsr), _p(dsd), 1.0 / HW,
                                          _p(drgb), _p(ddepth), N * Cc, HW, st), 'axpby_bwd_apply')
        _grads_enqueued()
        return (drgb, ddepth, d_wblend, d_prev, None, None, None, None, None, *dparams_ret)


def reweigh_fuse(rgb, depth, wblend=None, blend_mode=1, gate_params=None, temp=1.0, hard=False, prev=None,
                 noise=None):
    """One SkipESANet fusion point (model_skip_mod.py:235-311): the stage blend and, when `gate_params`
    (W1,b1,W2,b2 of SqueezeAndExcitationWeight.fc) is given, the gate evaluated on the same two maps.
    Returns (fused, wnext[N,2] | None, aux | None).  `noise` = Exp(1) samples [N,2] (tests); default: the
    kernel draws them with Philox (ops.manual_seed)."""
    gate = gate_params is not None
    params = tuple(gate_params) if gate else ()
    if prev is not None and not prev.is_contiguous():
        prev = prev.contiguous()
    return _ReweighFuse.apply(rgb, depth, wblend, prev, noise, blend_mode, gate, temp, hard, *params)


# ------------------------------------------------------------------------------------------------
# gate head
# ------------------------------------------------------------------------------------------------
class _GateHead(Function):
    @staticmethod
    def forward(ctx, pooled, fc, flop_table, temp, hard, force):
        lib = _lib()
        pooled, fc = _chk(pooled, 'pooled'), _chk(fc, 'fc')
        N = pooled.shape[0]
        J = pooled.numel() // N
        f32 = dict(device=pooled.device, dtype=torch.float32)
        weight, wcum, soft = torch.empty((N, 5), **f32), torch.empty((N, 4), **f32), torch.empty((N, 5), **f32)
        loss = torch.empty((), **f32)
        L.check(lib.dynmm_gate_head_fwd(_p(pooled), _p(fc), _p(weight), _p(wcum), _p(soft), _p(loss),
                                        _p(flop_table), None if force is None else force.data_ptr(), N, J,
                                        float(temp), int(hard), 0, _stream()), 'gate_head_fwd')
        ctx.temp = float(temp)
        ctx.fc_param = fc
        ctx.save_for_backward(pooled, fc, soft, flop_table)
        return weight, wcum, loss

    @staticmethod
    def backward(ctx, d_weight, d_wcum, d_loss):
        lib = _lib()
        pooled, fc, soft, flop_table = ctx.saved_tensors
        N = pooled.shape[0]
        J = pooled.numel() // N
        d_weight, d_wcum, d_loss = _chk(d_weight), _chk(d_wcum), _chk(d_loss)
        d_pooled = torch.empty_like(pooled)
        d_fc, d_fc_ret = _grad_dst(ctx.fc_param)
        L.check(lib.dynmm_gate_head_bwd(_p(d_weight), _p(d_wcum), _p(d_loss), _p(pooled), _p(fc), _p(soft),
                                        _p(flop_table), _p(d_pooled), _p(d_fc), N, J, ctx.temp, _stream()),
                'gate_head_bwd')
        _grads_enqueued()
        return d_pooled, d_fc_ret, None, None, None, None


def gate_head(pooled, fc_weight, flop_table, temp, hard, force_branch=None):
    """(weight[N,5], wcum[N,4], flop_loss) from the pooled gate features.  `force_branch` (device int32 [N],
    hard gates only): fixed branch per sample instead of the arg-max (benchmark / test knob)."""
    if force_branch is not None and (force_branch.dtype != torch.int32 or not force_branch.is_cuda):
        raise L.DynmmHipError('force_branch must be a device int32 tensor')
    return _GateHead.apply(pooled, fc_weight, flop_table, temp, hard, force_branch)


def gate_from_weight(weight, flop_table):
    """baseline / ini_stage: weight is given (constant one-hots); returns (weight, wcum, flop_loss)."""
    lib = _lib()
    weight = _chk(weight, 'weight')
    N = weight.shape[0]
    f32 = dict(device=weight.device, dtype=torch.float32)
    wcum, soft, loss = torch.empty((N, 4), **f32), torch.empty((N, 5), **f32), torch.empty((), **f32)
    L.check(lib.dynmm_gate_head_fwd(None, None, _p(weight), _p(wcum), _p(soft), _p(loss), _p(flop_table), None,
                                    N, 0, 1.0, 0, 1, _stream()), 'gate_head_fwd(mode 1)')
    return weight, wcum, loss


# ------------------------------------------------------------------------------------------------
# gate-decision compaction (K16)
# ------------------------------------------------------------------------------------------------
def gate_decide(weight):
    """Compaction plan from one-hot gate weights [N,5], computed on the device:
    (branch[N], order[N], inv[N], counts[4]) int32 — see dynmm_gate_decide."""
    lib = _lib()
    weight = _chk(weight.detach(), 'weight')
    N = weight.shape[0]
    i32 = dict(device=weight.device, dtype=torch.int32)
    branch, order, inv, counts = (torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(N, **i32),
                                  torch.empty(4, **i32))
    L.check(lib.dynmm_gate_decide(_p(weight), branch.data_ptr(), order.data_ptr(), inv.data_ptr(), counts.data_ptr(),
                                  N, _stream()), 'gate_decide')
    return branch, order, inv, counts


def batch_gather(x, index):
    """x[index] along the batch axis; `index` is a device int32 tensor.  Not differentiable (see batch_permute)."""
    lib = _lib()
    x = _chk(x, 'x')
    n_out = index.numel()
    out = torch.empty((n_out,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    row = x[0].numel()
    L.check(lib.dynmm_batch_gather(_p(x), index.data_ptr(), _p(out), n_out, C.c_size_t(row), _stream()), 'batch_gather')
    return out


class _BatchPermute(Function):
    @staticmethod
    def forward(ctx, x, index, inverse):
        ctx.save_for_backward(index, inverse)
        return batch_gather(x, index)

    @staticmethod
    def backward(ctx, g):
        index, inverse = ctx.saved_tensors
        return batch_gather(_chk(g, 'grad'), inverse), None, None


def batch_permute(x, index, inverse):
    """x[index] for a PERMUTATION `index` of the batch (device int32) with inverse `inverse`; differentiable:
    the gradient is gathered back with the inverse permutation."""
    return _BatchPermute.apply(x, index, inverse)


def batch_merge(base, sub, mapping):
    """out[n] = sub[mapping[n]] if mapping[n] >= 0 else base[n]; `mapping` is a device int32 [N]."""
    lib = _lib()
    base, sub = _chk(base, 'base'), _chk(sub, 'sub')
    out = torch.empty_like(base)
    row = base[0].numel()
    L.check(lib.dynmm_batch_merge(_p(base), _p(sub), mapping.data_ptr(), _p(out), base.shape[0], C.c_size_t(row),
                                  _stream()), 'batch_merge')
    return out


def _copy_rows(src, dst, row0, n_rows):
    """dst[row0:row0+n_rows] = src[row0:row0+n_rows] (whole samples), on the current stream."""
    if n_rows <= 0:
        return
    row = src[0].numel()
    off = row0 * row * 4
    L.check(_lib().dynmm_batch_gather(src.data_ptr() + off, None, dst.data_ptr() + off, n_rows, C.c_size_t(row),
                                      _stream()), 'copy_rows')


# ------------------------------------------------------------------------------------------------
# weighted 2-D cross entropy (caller of the path)
# ------------------------------------------------------------------------------------------------
class _CrossEntropy2d(Function):
    @staticmethod
    def forward(ctx, x, target_u8, class_weight):
        lib = _lib()
        x, class_weight = _chk(x, 'logits'), _chk(class_weight, 'class_weight')
        N, Cc, H, W = x.shape
        acc = torch.empty(2, device=x.device, dtype=torch.float64)
        L.check(lib.dynmm_ce2d_fwd(_p(x), _p(target_u8), _p(class_weight), _p(acc), N, Cc, H * W, 0, _stream()), 'ce2d_fwd')
        ctx.save_for_backward(x, target_u8, class_weight, acc)
        return (acc[0] / acc[1]).float()

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        x, target_u8, class_weight, acc = ctx.saved_tensors
        N, Cc, H, W = x.shape
        gscale = (g.double() / acc[1]).float().reshape(1).contiguous()
        dx = torch.empty_like(x)
        L.check(lib.dynmm_ce2d_bwd(_p(x), _p(target_u8), _p(class_weight), _p(gscale), _p(dx), N, Cc, H * W,
                                   _stream()), 'ce2d_bwd')
        return dx, None, None


def cross_entropy_2d(logits, target, class_weight):
    """sum_px w[t]*CE / sum_px w[t] with target 0 = void (FusionDynMM/src/utils.py:34-50)."""
    t = target if target.dtype == torch.uint8 else target.to(torch.uint8)
    if not t.is_contiguous():
        t = t.contiguous()
    return _CrossEntropy2d.apply(logits, t, class_weight)


class DeferredLogits:
    """Scale-0 output of a training step whose last up-sampling is left to the loss: `x` [N,C,H,W] is the input of
    the decoder's final learned 2x up-sampling (model.py:404-410) and `conv` its depthwise 3x3.  Handed to
    multi_scale_loss_backward, which runs the fused up-sampling + CE kernels (csrc/tail.hip) and never writes the
    [N,C,2H,2W] logits; materialize() computes them with the ordinary kernel for callers that want to look."""

    def __init__(self, x, conv):
        self.x, self.conv = x, conv

    @property
    def shape(self):
        N, Cc, H, W = self.x.shape
        return torch.Size((N, Cc, 2 * H, 2 * W))

    @property
    def requires_grad(self):
        return self.x.requires_grad or self.conv.weight.requires_grad

    @property
    def device(self):
        return self.x.device

    def materialize(self):
        with torch.no_grad():
            return upsample2x_dw3x3(self.x.detach(), self.conv.weight.detach(), self.conv.bias.detach())

    detach = materialize          # what a caller inspecting the step's outputs asks for


def _param_grad_commit(param, t, ret):
    """plain-autograd fallback of the in-place gradient protocol for gradients produced outside a Function."""
    if ret is not None and param.requires_grad:
        param.grad = ret if param.grad is None else param.grad + ret


def multi_scale_loss_backward(outs, targets, class_weight, flop_loss=None, ratio=0.0, budget=0.0, total_out=None):
    """train.py:313-323 for the HIP path, without a single PyTorch arithmetic kernel: the weighted CE of every
    scale (fp64 accumulators), total = sum_s CE_s + ratio * max(0, flop_loss - budget), and the backward pass of
    the whole step — the gradients of the logits come straight from dynmm_ce2d_bwd, seeded on the device, and are
    handed to autograd as the incoming gradients of the model outputs.
    Returns {'losses': [S], 'loss_flop': (), 'total': [1]} (detached device tensors).  `total_out`: a 1-element fp32
    device tensor that receives the total instead of a fresh one (data parallel: the slot that travels with the last
    gradient bucket, dp.GradBucketReducer.loss_slot)."""
    lib = _lib()
    st = _stream()
    outs = [o if isinstance(o, DeferredLogits) else _chk(o, 'logits') for o in outs]
    cw = _chk(class_weight, 'class_weight')
    S = len(outs)
    dev = outs[0].device
    acc = torch.zeros(2 * S, device=dev, dtype=torch.float64)
    tg = []
    tail = None
    for s_, (o, t) in enumerate(zip(outs, targets)):
        t = t if t.dtype == torch.uint8 else t.to(torch.uint8)
        t = t if t.is_contiguous() else t.contiguous()
        tg.append(t)
        N, Cc, H, W = o.shape
        if tuple(t.shape[-2:]) != (H, W) or t.numel() != N * H * W:
            raise L.DynmmHipError(f'target {tuple(t.shape)} does not match logits {tuple(o.shape)}')
        if isinstance(o, DeferredLogits):
            xin = _chk(o.x.detach(), 'tail input')
            wt, bs = _chk(o.conv.weight.detach(), 'w'), _chk(o.conv.bias.detach(), 'b')
            lse = torch.empty((N, H, W), device=dev, dtype=torch.float32)
            L.check(lib.dynmm_up2ce_fwd(_p(xin), _p(wt), _p(bs), t.data_ptr(), _p(cw), _p(lse), acc.data_ptr() + 16 * s_,
                                        N, Cc, H // 2, W // 2, 1, st), 'up2ce_fwd')
            tail = (s_, o, xin, wt, bs, lse)
            continue
        L.check(lib.dynmm_ce2d_fwd(_p(o), _p(t), _p(cw), acc.data_ptr() + 16 * s_, N, Cc, H * W, 1, st), 'ce2d_fwd')
    f32 = dict(device=dev, dtype=torch.float32)
    losses, total, gscale = torch.empty(S, **f32), torch.empty(1, **f32), torch.empty(S, **f32)
    if total_out is not None:
        if total_out.numel() != 1 or total_out.dtype != torch.float32 or total_out.device != dev:
            raise L.DynmmHipError('total_out must be a 1-element fp32 tensor on the logits\' device')
        total = total_out
    use_flop = flop_loss is not None and flop_loss.requires_grad and ratio > 0
    d_flop = torch.empty((), **f32) if use_flop else None
    lf = flop_loss.detach() if flop_loss is not None else None
    L.check(lib.dynmm_loss_head(acc.data_ptr(), S, _p(lf) if ratio > 0 else None, float(ratio), float(budget),
                                _p(losses), _p(total), _p(gscale), _p(d_flop), st), 'loss_head')
    roots, grads = [], []
    for s_, (o, t) in enumerate(zip(outs, tg)):
        if not o.requires_grad:
            continue
        N, Cc, H, W = o.shape
        if isinstance(o, DeferredLogits):
            _, _, xin, wt, bs, lse = tail
            dxin = torch.empty_like(xin)
            dw, dw_ret = _grad_dst(o.conv.weight)
            db, db_ret = _grad_dst(o.conv.bias)
            nb = lib.dynmm_up2ce_bwd_workspace_bytes(N, Cc, H // 2, W // 2)
            ws = torch.empty(max(nb // 4, 1), device=dev, dtype=torch.float32)
            L.check(lib.dynmm_up2ce_bwd(_p(xin), _p(wt), _p(bs), t.data_ptr(), _p(cw), _p(lse), gscale.data_ptr() + 4 * s_,
                                        _p(dxin), _p(dw), _p(db), _p(ws), N, Cc, H // 2, W // 2, st), 'up2ce_bwd')
            _grads_enqueued()
            _param_grad_commit(o.conv.weight, dw, dw_ret)
            _param_grad_commit(o.conv.bias, db, db_ret)
            if o.x.requires_grad:
                roots.append(o.x)
                grads.append(dxin)
            continue
        dx = torch.empty_like(o)
        L.check(lib.dynmm_ce2d_bwd(_p(o), _p(t), _p(cw), gscale.data_ptr() + 4 * s_, _p(dx), N, Cc, H * W, st), 'ce2d_bwd')
        roots.append(o)
        grads.append(dx)
    if use_flop:
        roots.append(flop_loss)
        grads.append(d_flop)
    if roots:
        torch.autograd.backward(roots, grads)
    return {'losses': losses, 'loss_flop': lf if lf is not None else torch.zeros((), **f32), 'total': total}


def validation_loss_accumulate(logits, target, class_weight, acc4):
    """acc4 (float64 [4], device) += (sum w[t]*CE, sum w[t], sum CE, #non-void pixels) of this batch — the running sums of
    CrossEntropyLoss2dForValidData / ...Unweighted.add_loss_of_batch (src/utils.py:65-69, 89-94), one pass over the logits."""
    lib = _lib()
    logits, class_weight = _chk(logits, 'logits'), _chk(class_weight, 'class_weight')
    t = target if target.dtype == torch.uint8 else target.to(torch.uint8)
    t = t if t.is_contiguous() else t.contiguous()
    N, Cc, H, W = logits.shape
    if tuple(t.shape) != (N, H, W):
        raise L.DynmmHipError(f'validation loss: label {tuple(t.shape)} does not match logits {tuple(logits.shape)}')
    if acc4.dtype != torch.float64 or acc4.numel() != 4 or not acc4.is_contiguous():
        raise L.DynmmHipError('acc4 must be a contiguous float64 [4] tensor')
    L.check(lib.dynmm_ce2d_valid(_p(logits), t.data_ptr(), _p(class_weight), acc4.data_ptr(), N, Cc, H * W, _stream()),
            'ce2d_valid')
    return acc4


def eval_confusion(logits, label, cm):
    """cm (int64 [C,C], device) += confusion counts of argmax(bilinear_resize(logits, label size)) vs label-1
    over non-void pixels — eval.py:117-141 fused into one kernel (no resized logits, no arg-max map)."""
    lib = _lib()
    logits = _chk(logits, 'logits')
    lab = label if label.dtype == torch.uint8 else label.to(torch.uint8)
    lab = lab if lab.is_contiguous() else lab.contiguous()
    N, Cc, H, W = logits.shape
    Ho, Wo = lab.shape[-2:]
    if cm.dtype != torch.int64 or not cm.is_contiguous() or cm.numel() != Cc * Cc:
        raise L.DynmmHipError('cm must be a contiguous int64 [C,C] tensor')
    L.check(lib.dynmm_eval_confusion(_p(logits), lab.data_ptr(), cm.data_ptr(), N, Cc, H, W, Ho, Wo, _stream()),
            'eval_confusion')
    return cm
