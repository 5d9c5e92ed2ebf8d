"""Tensor-level front-end of the HIP library: torch tensors in, torch tensors out, raw pointers and
the current HIP stream across the C ABI (include/glb.h).  PyTorch only provides device memory and
streams here.  Nothing in this module computes on the CPU; constructing `HipEngine` without a GPU
raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import GLB_EHIP, GlbError
from ._lib import (F32, BF16, F16, MASK_NONE, MASK_BITS, MASK_F32, MASK_PREPARED, RNG_NONE, RNG_PHILOX, RNG_NOISE,
                   STEP_HW_EXP, KvPlanArgs, MtRowsArgs, StepArgs, TrieArgs, TriePlan, TrieRowsArgs, MT19937, MT_POLY_WORDS, check)

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class HostRng:
    """torch-CPU-compatible MT19937 stream for GLB_RNG_NOISE (parity mode).

    `exponential(n)` returns the float32 values `torch.empty(n).exponential_(1, generator=g)` yields
    for a generator seeded with `seed` (README.md:87 / base.py:125-141 draw through it)."""

    def __init__(self, seed):
        self._lib = _lib.load()
        self.state = MT19937()
        self._lib.glb_mt19937_seed(C.byref(self.state), C.c_uint64(seed))

    def exponential(self, n, out=None):
        if out is None:
            out = torch.empty(n, dtype=torch.float32, pin_memory=torch.cuda.is_available())
        assert out.dtype == torch.float32 and out.is_contiguous() and out.device.type == "cpu"
        check(self._lib.glb_mt19937_exponential_f32(C.byref(self.state), _ptr(out), n))
        return out


def _low_priority_stream(device):
    """A stream of the LOWEST priority HIP has (torch offers normal and high only): work queued on it fills what the
    caller's stream leaves idle instead of competing with it.  Falls back to an ordinary torch stream."""
    try:
        hip = C.CDLL("libamdhip64.so")
        least, greatest = C.c_int(0), C.c_int(0)
        if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) == 0 and least.value > 0:
            h = C.c_void_p()
            with torch.cuda.device(device):
                if hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), C.c_int(least.value)) == 0 and h.value:  # (1: hipStreamNonBlocking)
                    return torch.cuda.ExternalStream(h.value, device=device)
    except OSError:
        pass
    return torch.cuda.Stream(device=device)


class DeviceRng:
    """torch's CPU generator for the parity draw, ON THE DEVICE (glb_mt19937_exponential_rows): the MT19937 stream of
    `torch.Generator().manual_seed(seed)` entered at every particle's row at once - no serial host loop, no noise tensor
    over PCIe.  `rows(...)` returns float32 [n_out, V] Exp(1) rows exactly as `torch.empty(V).exponential_(1, generator)`
    would yield them one after the other, and moves the stream on by the rows consumed.

    The stream's position is a 624-word window on the device; the jump polynomials for stride 2 V are computed on the host
    once per vocabulary (about 0.1 s) and cached on the engine.

    `ahead=True` (what `DeviceSIS` uses): the rows of the NEXT call depend on nothing but where the stream stands after this
    one, so `prefetch()` - called once the step that consumed the rows has been queued - generates them in stream order on a
    side stream while the caller's stream runs the next forward; the next `rows(...)` then only copies the rows its
    `row_slot` names into place and moves the position (the same rows, bit for bit: the same kernels made them)."""

    SMALL = 32

    def __init__(self, engine, seed, vocab, ahead=False):
        self.eng, self.seed, self.vocab, self.ahead = engine, int(seed), int(vocab), bool(ahead)
        self._side = None   # the side stream
        self._pre = None    # (max_draw, rows buffer, done event): what prefetch() has queued
        self._ws = None     # this stream's own windows (the engine's scratch is shared by every caller)
        self._last = None   # max_draw of the last rows() call
        self.reset()

    def reset(self):
        if self._pre is not None:  # (let the side stream finish with the window before it is replaced)
            self._pre[2].synchronize()
            self._pre = None
        w = np.zeros(624, np.uint32)
        check(self.eng.lib.glb_mt19937_window(C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF), C.c_void_p(w.ctypes.data)))
        self.window = torch.from_numpy(w.view(np.int32)).to(self.eng.device)
        self.rows_drawn = 0  # host-side tally when the counts are known here (diagnostic)

    def _args(self, max_draw, n_big, polys):
        eng = self.eng
        need = eng.lib.glb_mt19937_rows_workspace(max_draw, self.SMALL)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=eng.device)
        a = MtRowsArgs()
        a.struct_size = C.sizeof(MtRowsArgs)
        a.window = self.window.data_ptr()
        a.polys = polys.data_ptr()
        a.n_small, a.n_big = self.SMALL, n_big
        a.vocab, a.max_draw_rows = self.vocab, max_draw
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        return a

    def rows(self, n_out, row_slot=None, n_draw=None, max_draw=None, out=None):
        """n_out rows; row_slot: int32 [n_out] device - the stream row every output row takes (negative: a row of ones),
        None: identity; n_draw: int32 device scalar tensor - rows the stream moves on by (None: max_draw); max_draw: host
        upper bound of it (default n_out)."""
        eng, V = self.eng, self.vocab
        max_draw = int(n_out if max_draw is None else max_draw)
        polys, n_big = eng.mt_polys(V, max_draw + 1)
        if out is None:
            out = torch.empty((n_out, V), dtype=torch.float32, device=eng.device)
        eng._check_dev(row_slot, n_draw, out)
        if row_slot is not None and (row_slot.dtype != torch.int32 or row_slot.numel() != n_out):
            raise ValueError("row_slot must be int32 [n_out]")
        if n_draw is not None and n_draw.dtype != torch.int32:
            raise TypeError("n_draw must be an int32 device scalar")
        pre, self._pre = self._pre, None
        if pre is not None and pre[0] != max_draw:  # (made for another shape: wait it out - it reads the window - and drop it)
            pre[2].synchronize()
            pre = None
        a = self._args(max_draw, n_big, polys)
        a.window_out = self.window.data_ptr()
        a.n_draw = None if n_draw is None else n_draw.data_ptr()
        a.n_out_rows = n_out
        a.row_slot = None if row_slot is None else row_slot.data_ptr()
        a.out, a.out_ld = out.data_ptr(), out.stride(0) if n_out > 1 else max(V, out.stride(0))
        if pre is not None:  # the rows stand in pre[1], the windows in the workspace: copy and move on
            torch.cuda.current_stream(eng.device).wait_event(pre[2])
            a.reuse_windows = 1
            a.rows_from, a.rows_from_ld = pre[1].data_ptr(), pre[1].stride(0)
        check(eng.lib.glb_mt19937_exponential_rows(C.byref(a), eng._stream()))
        if n_draw is None:
            self.rows_drawn += max_draw
        self._last = max_draw
        return out

    def prefetch(self):
        """Queue the generation of the next call's rows (as many as the last call could draw, in stream order) on the side
        stream; it starts when everything queued on the current stream so far is done - call it after the step that reads
        the last rows() has been queued.  No-op unless `ahead`."""
        if not self.ahead or self._last is None or self._pre is not None:
            return
        eng, V, max_draw = self.eng, self.vocab, self._last
        if self._side is None:
            self._side = _low_priority_stream(eng.device)
            self._rows_buf = None
        if self._rows_buf is None or self._rows_buf.shape[0] < max_draw:
            self._rows_buf = torch.empty((max_draw, V), dtype=torch.float32, device=eng.device)
        polys, n_big = eng.mt_polys(V, max_draw + 1)
        a = self._args(max_draw, n_big, polys)
        a.n_out_rows = max_draw
        a.out, a.out_ld = self._rows_buf.data_ptr(), self._rows_buf.stride(0)
        self._side.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(self._side):
            check(eng.lib.glb_mt19937_exponential_rows(C.byref(a), eng._stream()))
            done = torch.cuda.Event()
            done.record(self._side)
        self._pre = (max_draw, self._rows_buf, done)


class PreparedMasks:
    """Bit masks in the kernels' own layout (glb_mask_prepare): built once for masks that do not change."""

    def __init__(self, blob, n_masks, vocab, dtype):
        self.blob, self.n_masks, self.vocab, self.dtype = blob, n_masks, vocab, dtype


class StepPlan:
    """A fused step whose argument block is already filled (HipEngine.step_plan)."""

    __slots__ = ("eng", "args", "keep", "out", "_ref")

    def __init__(self, eng, args, keep, out):
        self.eng, self.args, self.keep, self.out = eng, args, keep, out
        self._ref = C.byref(args)

    def run(self, offset=None, seed=None):
        a = self.args
        if offset is not None:
            a.offset = offset
        if seed is not None:
            a.seed = seed
        if self.eng._step_ws is not self.keep[-1]:
            raise RuntimeError("the engine's scratch buffer was reallocated since this plan was made: make a new plan")
        check(self.eng.lib.glb_logprob_mask_sample(self._ref, self.eng._stream()))
        return self.out

    def run_timed(self, events, offset=None, seed=None):
        """run() whose launch(es) carry `events` = (start, stop) - two recorded-once torch.cuda.Event(enable_timing=True)
        - as their own start / stop stamps (glb_logprob_mask_sample_timed): start.elapsed_time(stop) is the duration of
        the step's launches on the device."""
        a = self.args
        if offset is not None:
            a.offset = offset
        if seed is not None:
            a.seed = seed
        if self.eng._step_ws is not self.keep[-1]:
            raise RuntimeError("the engine's scratch buffer was reallocated since this plan was made: make a new plan")
        check(self.eng.lib.glb_logprob_mask_sample_timed(self._ref, self.eng._stream(), C.c_void_p(events[0].cuda_event),
                                                         C.c_void_p(events[1].cuda_event)))
        return self.out


class _OnDevice:
    """The library's entry points, each run with the engine's device current.  HIP launches go to the calling thread's
    current device; the engine's tensors and stream belong to `self.device`, so a caller that has another device
    current (one process driving several GPUs) must not be able to send a launch to the wrong one."""

    def __init__(self, lib, index):
        self._lib, self._index = lib, index

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        index = self._index

        def call(*args):
            if torch.cuda.current_device() == index:
                return fn(*args)
            with torch.cuda.device(index):
                return fn(*args)

        setattr(self, name, call)  # resolved once per entry point
        return call


class HipEngine:
    """All device work of the hot path for one GPU."""

    CONTRACTS = ("poly", "hw", "auto")

    def __init__(self, device=None, contract="auto"):
        """contract: the arithmetic of the fused step's terms (include/glb.h, GLB_STEP_HW_EXP) - "poly":
        the polynomial exponential, restated bit for bit by the oracle; "hw": the hardware's v_exp_f32 - within one ulp of
        2^y, deterministic and independent of launch geometry on gfx950, checked against the oracle by tolerance and
        against torch's ids in parity mode (tests/test_step_hw_gpu.py); "auto" (the default): "hw" for 16-bit logits (a
        sixth less time on rows bound by instruction issue), "poly" for float32 rows (bound by memory: 4 % to gain, and
        their results stay the oracle's bit for bit).  `step(contract=...)` overrides it per call."""
        if contract not in self.CONTRACTS:
            raise ValueError(f"contract must be one of {self.CONTRACTS}, got {contract!r}")
        self.contract = contract
        lib = _lib.load()
        if not torch.cuda.is_available() or lib.glb_device_count() <= 0:
            raise RuntimeError(
                "HipEngine needs a HIP device (MI355X / gfx950); none is visible and there is no CPU fallback"
            )
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError(f"HipEngine device must be a HIP device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _OnDevice(lib, self.device.index)
        self._ws = None
        self._step_ws = None
        self._trie_ws = None
        self._mt_polys = {}
        self._ptr_tables = {}

    def __del__(self):
        try:
            if self._step_ws is not None:
                self.lib.glb_workspace_release(_ptr(self._step_ws))
        except Exception:  # interpreter shutdown
            pass

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _i32(self, n):
        return torch.empty(n, dtype=torch.int32, device=self.device)

    def _f32(self, n):
        return torch.empty(n, dtype=torch.float32, device=self.device)

    def _check_dev(self, *ts):
        for t in ts:
            if t is not None and (t.device != self.device or not t.is_contiguous()):
                if t.device != self.device:
                    raise ValueError(f"tensor on {t.device}, engine on {self.device}")
                raise ValueError("tensor must be contiguous")

    # ------------------------------------------------------------------------------------------
    def _scratch(self, need):
        """The step's scratch buffer, zeroed and registered with the library (glb_workspace_init): calls that find their
        workspace registered run as ONE launch."""
        if self._step_ws is None or self._step_ws.numel() < need:
            if self._step_ws is not None:
                self.lib.glb_workspace_release(_ptr(self._step_ws))
            self._step_ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
            check(self.lib.glb_workspace_init(_ptr(self._step_ws), self._step_ws.numel(), self._stream()))
        return self._step_ws

    def step(self, logits, vocab=None, row_of=None, mask_kind=MASK_NONE, mask=None, mask_id=None,
             rng_mode=RNG_NONE, noise=None, seed=0, offset=0, particle_base=0, logit_scale=1.0,
             want_lse=True, out=None, row_mask_id=None, out_margin=None, _plan=False, timing_events=None,
             contract=None):
        """Fused particle step (glb_logprob_mask_sample).  Returns (logZ, lse, token) device tensors.

        logits: [n_rows, ld] (last dim contiguous; rows may be strided), vocab <= ld.
        mask: int32 bit rows / float rows, or a `PreparedMasks` (prepare_masks).  `row_mask_id` gives the mask per
        logits row (the mask is a function of the context): shared rows are then reduced once.
        A launch whose waves gave up waiting shows as token -2 / NaN and in `error_word()`: `raise_if_failed`.
        contract: "poly" / "hw" / "auto" (None: the engine's) - see __init__.
        """
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise ValueError("logits must be 2-D with unit inner stride")
        if logits.dtype not in _DT:
            raise TypeError(f"unsupported logits dtype {logits.dtype}")
        n_rows, width = logits.shape
        ld = logits.stride(0) if n_rows > 1 else max(width, logits.stride(0))
        V = width if vocab is None else vocab
        n = n_rows if row_of is None else row_of.numel()
        self._check_dev(row_of, mask_id, row_mask_id, noise)
        if out is None:
            logZ, lse, tok = self._f32(n), (self._f32(n) if want_lse else None), None
            if rng_mode != RNG_NONE:
                tok = self._i32(n)
        else:
            logZ, lse, tok = out
        a = StepArgs()
        a.struct_size = C.sizeof(StepArgs)
        a.logits = logits.data_ptr()
        a.dtype = _DT[logits.dtype]
        a.n_rows, a.vocab, a.ld = n_rows, V, ld
        a.logit_scale = logit_scale
        a.n_particles = n
        a.row_of = None if row_of is None else row_of.data_ptr()
        n_masks = 0
        if isinstance(mask, PreparedMasks):
            if mask.vocab != V or (mask.dtype == F32) != (_DT[logits.dtype] == F32):
                raise ValueError("prepared masks were built for another vocabulary / element width")
            a.mask_kind = MASK_PREPARED
            a.mask = mask.blob.data_ptr()
            a.mask_ld = 0
            a.n_masks = mask.n_masks
        else:
            a.mask_kind = mask_kind
            if mask_kind != MASK_NONE:
                self._check_dev(mask)
                if mask.dim() != 2 or mask.stride(1) != 1:
                    raise ValueError("mask must be 2-D")
                want = torch.float32 if mask_kind == MASK_F32 else torch.int32
                if mask.dtype != want:
                    raise TypeError(f"mask dtype {mask.dtype}, expected {want}")
                a.mask = mask.data_ptr()
                a.mask_ld = mask.stride(0) if mask.shape[0] > 1 else mask.shape[1]
                a.n_masks = mask.shape[0]
                n_masks = mask.shape[0] if mask_kind == MASK_BITS else 0
        if a.mask_kind != MASK_NONE:
            a.mask_id = None if mask_id is None else mask_id.data_ptr()
            a.row_mask_id = None if row_mask_id is None else row_mask_id.data_ptr()
        a.rng_mode = rng_mode
        if rng_mode == RNG_NOISE:
            if noise.dim() != 2 or noise.dtype != torch.float32:
                raise ValueError("noise must be float32 [n_particles, >=vocab]")
            a.noise = noise.data_ptr()
            # one row for everybody (batch_sample seeds every sequence alike): pitch 0
            a.noise_ld = noise.stride(0) if noise.shape[0] > 1 else (0 if n > 1 else noise.shape[1])
        a.seed, a.offset, a.particle_base = seed, offset, particle_base
        a.out_logZ = None if logZ is None else logZ.data_ptr()
        a.out_lse = None if lse is None else lse.data_ptr()
        a.out_token = None if tok is None else tok.data_ptr()
        a.out_margin = None if out_margin is None else out_margin.data_ptr()  # parity mode: tie margin of every draw
        contract = self.contract if contract is None else contract
        if contract not in self.CONTRACTS:
            raise ValueError(f"contract must be one of {self.CONTRACTS}, got {contract!r}")
        a.flags = STEP_HW_EXP if contract == "hw" or (contract == "auto" and a.dtype != F32) else 0
        ws = self._scratch(self.lib.glb_step_workspace_bytes(n, n_rows, V, n_masks))
        a.workspace = ws.data_ptr()
        a.workspace_bytes = ws.numel()
        if _plan:  # step_plan(): the filled argument block, the tensors it points into, the outputs
            return StepPlan(self, a, (logits, row_of, mask, mask_id, row_mask_id, noise, out_margin, ws),
                            (logZ, lse, tok))
        if timing_events is not None:  # (start, stop) from timing_events(): the launches' own start / stop stamps
            check(self.lib.glb_logprob_mask_sample_timed(C.byref(a), self._stream(), C.c_void_p(timing_events[0].cuda_event),
                                                         C.c_void_p(timing_events[1].cuda_event)))
        else:
            check(self.lib.glb_logprob_mask_sample(C.byref(a), self._stream()))
        return logZ, lse, tok

    # ---- failed one-launch calls (include/glb.h: glb_workspace_check) ---------------------------------------------
    def error_word(self):
        """int32 [1] device view of the scratch buffer's error word: nonzero after a one-launch call whose waves gave
        up waiting for their records (token -2 / NaN outputs).  Hosts that copy results back let it ride along and hand
        the value to `raise_if_failed`."""
        ws = self._step_ws
        if ws is None:
            return torch.zeros(1, dtype=torch.int32, device=self.device)
        return ws[ws.numel() - 64: ws.numel() - 60].view(torch.int32)

    def raise_if_failed(self, err_count=None, tokens=None, lse=None, what="glb_logprob_mask_sample"):
        """Raise when a one-launch call did not complete: `err_count` (the error word's value, already on the host),
        host `tokens` holding -2, or host `lse` / logZ values holding NaN.  Clears the error word."""
        bad = bool(err_count)
        if tokens is not None and not bad:
            bad = bool((np.asarray(tokens) == -2).any())
        if lse is not None and not bad:
            bad = bool(np.isnan(np.asarray(lse, dtype=np.float32)).any())
        if bad:
            if self._step_ws is not None:
                self.error_word().zero_()
            raise GlbError(GLB_EHIP, f"{what}: waves of the launch gave up waiting for their row's records (token -2 / "
                                     "NaN): the call did not complete and its results must not be used")

    def check(self):
        """Synchronising check of the scratch buffer's error word (glb_workspace_check): raises after a failed call."""
        if self._step_ws is not None:
            check(self.lib.glb_workspace_check(_ptr(self._step_ws), self._stream()))

    def set_spin_limit(self, microseconds):
        """Watchdog of the waits inside the one-launch calls (0: the default, 2 s; None: GLB_SPIN_NONE - a wave gives up
        at the first poll that finds a record missing, which forces the failure path)."""
        check(self.lib.glb_set_spin_limit(0xFFFFFFFFFFFFFFFF if microseconds is None else int(microseconds)))

    def timing_events(self, n):
        """n (start, stop) pairs of HIP events for `step(timing_events=...)` / `StepPlan.run_timed`: created and
        recorded once here (torch makes the hipEvent_t on first record), so that using them later puts nothing but the
        launch itself on the stream."""
        with torch.cuda.device(self.device):
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in pairs:
                a.record()
                b.record()
            torch.cuda.current_stream(self.device).synchronize()
        return pairs

    def step_plan(self, logits, **kw):
        """`step()` with the host work done once: validates and fills the argument block, returns a `StepPlan` whose
        `run(offset=..., seed=...)` only launches (a few microseconds of host time instead of a few tens - for loops
        that call the same configuration again and again, e.g. `bench.py --workload kernel`).  The tensors must keep
        their storage; the scratch buffer is the engine's, so a plan is valid until another, larger step reallocates
        it (run() checks)."""
        return self.step(logits, _plan=True, **kw)

    def noise_rng(self, seed, vocab, ahead=False):
        """The parity draw's noise source: torch's CPU generator seeded with `seed`, rows of `vocab` exponentials, on the
        device (DeviceRng; ahead: the next call's rows generated on a side stream once `prefetch()` is called)."""
        return DeviceRng(self, seed, vocab, ahead=ahead)

    def mt_polys(self, vocab, n_windows):
        """Device table of the MT19937 jump polynomials for rows of `vocab` exponentials (stride 2 * vocab words), enough
        for `n_windows` row windows per call (glb_mt19937_jump_polys; computed on the host once and cached).  Returns
        (int64 tensor [(SMALL + n_big), 312], n_big)."""
        small = DeviceRng.SMALL
        need_big = max(1, -(-int(n_windows) // small))
        ent = self._mt_polys.get(vocab)
        if ent is None or ent[1] < need_big:
            n_big = max(need_big, 2 * ent[1] if ent else 0)
            host = np.zeros((small + n_big, MT_POLY_WORDS), np.uint64)
            check(self.lib.glb_mt19937_jump_polys(2 * int(vocab), small, n_big, C.c_void_p(host.ctypes.data)))
            ent = self._mt_polys[vocab] = (torch.from_numpy(host.view(np.int64)).to(self.device), n_big)
        return ent

    def prepare_masks(self, bits, vocab, logits_dtype=torch.float32):
        """int32 bit rows [K, >= ceil(V/32)] -> `PreparedMasks` for logits of `logits_dtype` (glb_mask_prepare)."""
        if bits.dim() == 1:
            bits = bits[None]
        self._check_dev(bits)
        if bits.dtype != torch.int32:
            raise TypeError("bit masks must be int32 words")
        K = bits.shape[0]
        need = self.lib.glb_mask_prepared_bytes(K, vocab)
        blob = torch.empty(need, dtype=torch.uint8, device=self.device)
        dt = _DT[logits_dtype]
        check(self.lib.glb_mask_prepare(_ptr(bits), K, vocab, bits.stride(0) if K > 1 else bits.shape[1], dt,
                                        _ptr(blob), need, self._stream()))
        return PreparedMasks(blob, K, vocab, dt)

    def update_prepared_masks(self, prepared, bits, rows):
        """Re-prepare only the masks `rows` (int32 device tensor) of `prepared` (a `PreparedMasks` made from `bits` before)
        from their current bit rows (glb_mask_prepare_rows): masks that change a few at a time cost what changed."""
        if bits.dim() == 1:
            bits = bits[None]
        self._check_dev(bits, rows)
        if bits.dtype != torch.int32 or rows.dtype != torch.int32:
            raise TypeError("bit masks and row indices must be int32")
        if bits.shape[0] != prepared.n_masks:
            raise ValueError("the bit rows are not the ones these masks were prepared from")
        K = prepared.n_masks
        check(self.lib.glb_mask_prepare_rows(_ptr(bits), K, prepared.vocab, bits.stride(0) if K > 1 else bits.shape[1], prepared.dtype,
                                             _ptr(rows), rows.numel(), _ptr(prepared.blob), prepared.blob.numel(), self._stream()))
        return prepared

    def log_softmax_rows(self, logits, vocab=None, logit_scale=1.0, out=None, want_lse=False, workspace=None,
                         out_dtype=torch.float32):
        """out[r] = logits[r] - logsumexp(logits[r]) (glb_log_softmax_rows).  `out_dtype`: float32 (default) or the
        logits' own dtype - what the reference returns (cache.py:96 keeps the model's dtype): the float32 result rounded
        to nearest even.  `workspace`: a uint8 device tensor to lend instead of the engine's own scratch (one the
        library has not initialised takes the form that does not tag its records)."""
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise ValueError("logits must be 2-D with unit inner stride")
        n_rows, width = logits.shape
        V = width if vocab is None else vocab
        ld = logits.stride(0) if n_rows > 1 else max(width, logits.stride(0))
        if out is None:
            out = torch.empty((n_rows, V), dtype=out_dtype, device=self.device)
        if out.dtype not in (torch.float32, logits.dtype):
            raise TypeError(f"log-probabilities leave as float32 or as {logits.dtype}, not {out.dtype}")
        lse = self._f32(n_rows) if want_lse else None
        out_ld = out.stride(0) if n_rows > 1 else max(V, out.stride(0))
        need = self.lib.glb_log_softmax_workspace_bytes(n_rows, V)
        ws = self._scratch(need) if workspace is None else workspace
        if ws.numel() < need:
            raise ValueError(f"workspace of {ws.numel()} bytes, {need} needed")
        check(self.lib.glb_log_softmax_rows(_ptr(logits), _DT[logits.dtype], n_rows, V, ld, logit_scale, _ptr(out),
                                            _DT[out.dtype], out_ld, _ptr(lse), _ptr(ws), ws.numel(), self._stream()))
        return (out, lse) if want_lse else out

    def row_lse(self, logits, vocab=None, logit_scale=1.0):
        """logsumexp(logits[r] * logit_scale) per row, float32 [n_rows] (glb_log_softmax_rows with no rows out: the rows are
        read once, nothing but n_rows floats is written) - the lse `trie_rows` / `trie_masses` take next to raw logits."""
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise ValueError("logits must be 2-D with unit inner stride")
        n_rows, width = logits.shape
        V = width if vocab is None else vocab
        ld = logits.stride(0) if n_rows > 1 else max(width, logits.stride(0))
        lse = self._f32(n_rows)
        ws = self._scratch(self.lib.glb_log_softmax_workspace_bytes(n_rows, V))
        check(self.lib.glb_log_softmax_rows(_ptr(logits), _DT[logits.dtype], n_rows, V, ld, logit_scale, None, F32, 0,
                                            _ptr(lse), _ptr(ws), ws.numel(), self._stream()))
        return lse

    def mask_to_bits(self, mask):
        """{0,-inf} float log-masks [K, V] -> packed int32 bit rows [K, ceil(V/32)] + non-binary flag."""
        if mask.dim() == 1:
            mask = mask[None]
        mask = mask.to(self.device, torch.float32).contiguous()
        K, V = mask.shape
        W = (V + 31) // 32
        bits = torch.empty((K, W), dtype=torch.int32, device=self.device)
        flag = self._i32(1)
        check(self.lib.glb_mask_f32_to_bits(_ptr(mask), K, V, V, _ptr(bits), W, _ptr(flag), self._stream()))
        return bits, flag

    # ------------------------------------------------------------------------------------------
    def hash_contexts(self, tokens, starts, lengths):
        """The contexts' hashes (int64 tensor holding the 64-bit values) for `group_contexts(hashes=...)`; a context's
        hash extends token by token (`particles_advance(hashes=...)` keeps it up to date)."""
        n = lengths.numel()
        self._check_dev(tokens, starts, lengths)
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        check(self.lib.glb_hash_contexts(_ptr(tokens), _ptr(starts), _ptr(lengths), n, _ptr(out), self._stream()))
        return out

    def group_contexts(self, tokens, starts, lengths, hashes=None):
        """Exact dedup, first-appearance order.  Returns (group_of[n], rep[n], n_groups[1]) on device.  `hashes`: the
        contexts' hashes if the caller keeps them (hash_contexts); the tokens are then read only to confirm duplicates."""
        n = lengths.numel()
        self._check_dev(tokens, starts, lengths, hashes)
        need = self.lib.glb_group_contexts_workspace(n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
        group_of, rep, ng = self._i32(n), self._i32(n), self._i32(1)
        check(self.lib.glb_group_contexts(_ptr(tokens), _ptr(starts), _ptr(lengths), n, _ptr(hashes), _ptr(group_of),
                                          _ptr(rep), _ptr(ng), _ptr(self._ws), self._ws.numel(), self._stream()))
        return group_of, rep, ng

    def match_prefixes(self, tokens, starts, lengths, prefix_tokens, prefix_starts, prefix_lengths):
        n = lengths.numel()
        npre = 0 if prefix_lengths is None else prefix_lengths.numel()
        pref, base = self._i32(n), self._i32(n)
        check(self.lib.glb_match_prefixes(_ptr(tokens), _ptr(starts), _ptr(lengths), n, _ptr(prefix_tokens),
                                          _ptr(prefix_starts), _ptr(prefix_lengths), npre, _ptr(pref), _ptr(base),
                                          self._stream()))
        return pref, base

    def gather_padded(self, tokens, starts, lengths, sel, n_sel, base, pad_id, p_max, l_max):
        dev = self.device
        ids = torch.empty((n_sel, l_max), dtype=torch.int64, device=dev)
        am = torch.empty((n_sel, p_max + l_max), dtype=torch.int64, device=dev)
        pos = torch.empty((n_sel, l_max), dtype=torch.int64, device=dev)
        last = self._i32(n_sel)
        check(self.lib.glb_gather_padded(_ptr(tokens), _ptr(starts), _ptr(lengths), _ptr(sel), n_sel, _ptr(base),
                                         pad_id, p_max, l_max, _ptr(ids), _ptr(am), _ptr(pos), _ptr(last),
                                         self._stream()))
        return ids, am, pos, last

    def gather_kv_padded(self, slab_ptrs, slab_len, prefix_of, heads, head_dim, p_max, dtype):
        """slab_ptrs: int64 device tensor of raw pointers to [heads, len_k, head_dim] slabs."""
        n_rows = prefix_of.numel()
        out = torch.empty((n_rows, heads, p_max, head_dim), dtype=dtype, device=self.device)
        check(self.lib.glb_gather_kv_padded(_ptr(slab_ptrs), _ptr(slab_len), slab_len.numel(), _ptr(prefix_of),
                                            n_rows, heads, head_dim, p_max, out.element_size(), _ptr(out),
                                            self._stream()))
        return out

    def particles_advance(self, contexts, lengths, active, log_weights, logZ, token, eos_id, max_len, hashes=None):
        n, ld = contexts.shape
        check(self.lib.glb_particles_advance(_ptr(contexts), ld, _ptr(lengths), _ptr(active), _ptr(log_weights),
                                             _ptr(logZ), _ptr(token), n, eos_id, max_len, _ptr(hashes), self._stream()))

    def normalize_weights(self, log_weights):
        n = log_weights.numel()
        probs, stats = self._f32(n), self._f32(2)
        check(self.lib.glb_normalize_weights(_ptr(log_weights), n, _ptr(probs), _ptr(stats), self._stream()))
        return probs, stats

    # ---- device-resident particle state ------------------------------------------------------------------
    def kv_append(self, slab, new_rows, pos, rows=None):
        """slab[rows[i] (or i), h, pos[i], :] = new_rows[i, h, 0, :] (glb_kv_append).  slab [R, H, cap, Dh] contiguous;
        new_rows [n, H, 1, Dh] with unit inner stride (usually a transposed view of the projection output); rows: int32
        [n] slab row of every forward row (shared KV rows), None: row i."""
        R, H, cap, Dh = slab.shape
        n = new_rows.shape[0]
        assert new_rows.shape == (n, H, 1, Dh) and new_rows.stride(3) == 1 and slab.is_contiguous()
        assert rows is not None or n == R
        check(self.lib.glb_kv_append(_ptr(slab), _ptr(new_rows), _ptr(pos), _ptr(rows), n, H, cap, Dh, new_rows.stride(0),
                                     new_rows.stride(1), slab.element_size(), self._stream()))

    def slab_attention(self, query, k_new, v_new, k_slab, v_slab, pos, scale):
        """Attention of a one-token forward over slab rows where they lie, the new token's K / V appended on the way
        (glb_slab_attention).  query [R, H, 1, Dh], k_new / v_new [R, H_kv, 1, Dh] (unit inner stride; usually strided
        views of the projection output), slabs [R, H_kv, cap, Dh] contiguous, pos int32 [R].  Returns [R, 1, H, Dh]."""
        R, H, _, Dh = query.shape
        Hkv, cap = k_slab.shape[1], k_slab.shape[2]
        assert query.stride(3) == 1 and k_new.stride(3) == 1 and v_new.stride(3) == 1 and k_slab.is_contiguous() and v_slab.is_contiguous()
        assert k_new.dtype == v_new.dtype == query.dtype == k_slab.dtype
        out = torch.empty((R, 1, H, Dh), dtype=query.dtype, device=self.device)
        check(self.lib.glb_slab_attention(_ptr(query), query.stride(0), query.stride(1), _ptr(k_new), k_new.stride(0),
                                          k_new.stride(1), _ptr(v_new), v_new.stride(0), v_new.stride(1), _ptr(k_slab),
                                          _ptr(v_slab), _ptr(pos), R, H, Hkv,
                                          cap, Dh, float(scale), _DT[query.dtype], _ptr(out), self._stream()))
        return out

    @staticmethod
    def slab_attention_supports(dtype, head_dim):
        return dtype in _DT and head_dim in (16, 32, 64, 128)

    def short_attention(self, query, key, value, mask, scale):
        """Masked attention of a padded batch of short contexts (glb_short_attention).  query [U, H, Lq, Dh], key / value
        [U, H_kv, Lk, Dh] (unit inner stride), mask: bool [U, 1, Lq, Lk] (true = attend) or None (causal).  Returns
        [U, Lq, H, Dh]."""
        U, H, Lq, Dh = query.shape
        Hkv, Lk = key.shape[1], key.shape[2]
        assert query.stride(3) == 1 and key.stride(3) == 1 and value.stride(3) == 1 and key.dtype == value.dtype == query.dtype
        out = torch.empty((U, Lq, H, Dh), dtype=query.dtype, device=self.device)
        i64x3 = C.c_int64 * 3
        m_ptr, m_sr, m_sq = None, 0, 0
        if mask is not None:
            assert mask.dtype == torch.bool and mask.dim() == 4 and mask.shape[-1] == Lk and mask.stride(3) == 1 and mask.shape[1] == 1
            m_ptr, m_sr, m_sq = _ptr(mask), (mask.stride(0) if mask.shape[0] > 1 else 0), (mask.stride(2) if mask.shape[2] > 1 else 0)
        check(self.lib.glb_short_attention(_ptr(query), i64x3(*query.stride()[:3]), _ptr(key), i64x3(*key.stride()[:3]), _ptr(value),
                                           i64x3(*value.stride()[:3]), m_ptr, m_sr, m_sq, U, H, Hkv, Lq, Lk, Dh, float(scale),
                                           _DT[query.dtype], _ptr(out), self._stream()))
        return out

    def kv_gather_rows(self, srcs, dsts, src_row_of, len_of, srcs_stable=True):
        """dsts[t][i, h, p] = srcs[t][src_row_of[i], h, p] for p < len_of[i], for every tensor pair of the two lists
        (all [rows, heads, cap, head_dim], contiguous) in ONE launch through device pointer tables
        (glb_kv_gather_rows).  src_row_of[i] < 0 leaves row i untouched."""
        s0, d0 = srcs[0], dsts[0]
        assert all(t.is_contiguous() for t in srcs) and all(t.is_contiguous() for t in dsts)
        # (srcs_stable=False: the sources are a one-off - the KV a forward has just returned: their table is not kept)
        sp = self._ptr_table(srcs) if srcs_stable else torch.tensor([t.data_ptr() for t in srcs], dtype=torch.int64, device=self.device)
        dp = self._ptr_table(dsts)
        check(self.lib.glb_kv_gather_rows(_ptr(sp), _ptr(dp), len(srcs), d0.shape[0], d0.shape[1], d0.shape[3],
                                          s0.shape[2], d0.shape[2], _ptr(src_row_of), _ptr(len_of), d0.element_size(),
                                          self._stream()))

    def _ptr_table(self, tensors):
        """Device table of the tensors' addresses.  Slab sets are handed over again and again: the table of a set is made
        once (a host list -> device copy from pageable memory is a synchronous copy - one in every step kept the host from
        running ahead of the GPU).  An entry names its tensors by WEAK references: it never keeps a slab set alive (a set is
        gigabytes at the Llama-3-8B shape, and an engine outlives many DeviceSIS / SlabKV objects), and it is only valid
        while every one of them is the very object it was made for - an address the allocator hands out again after its
        owner died does not pass for the old tensor.  Entries whose tensors are gone leave at the next miss."""
        import weakref

        key = tuple(t.data_ptr() for t in tensors)
        ent = self._ptr_tables.get(key)
        if ent is not None and all(r() is t for r, t in zip(ent[1], tensors)):
            return ent[0]
        dead = [k for k, (_, refs) in self._ptr_tables.items() if any(r() is None for r in refs)]
        for k in dead:
            del self._ptr_tables[k]
        if len(self._ptr_tables) >= 64:
            self._ptr_tables.pop(next(iter(self._ptr_tables)))
        table = torch.tensor(key, dtype=torch.int64, device=self.device)
        self._ptr_tables[key] = (table, [weakref.ref(t) for t in tensors])
        return table

    def gather_rows_i32(self, src, row_of, out=None):
        """out[i] = src[row_of[i]] for an int32 matrix (glb_gather_rows_i32)."""
        n, width = row_of.numel(), src.shape[1]
        if out is None:
            out = torch.empty((n, width), dtype=torch.int32, device=self.device)
        check(self.lib.glb_gather_rows_i32(_ptr(src), src.stride(0), _ptr(row_of), n, width, _ptr(out), out.stride(0),
                                           self._stream()))
        return out

    # ---- KV rows shared between contexts: the block table on the device ------------------------------------------
    def match_rows(self, tokens, starts, lengths, rep, n_groups, row_tok, row_len, row_hash):
        """For every dedup group the table row that holds exactly its context, else the row that holds its first L - 1
        tokens, else -1 (glb_match_rows).  Returns (old_row int32 [n], group_hash int64 [n]); entries past the group
        count are unspecified."""
        n = lengths.numel()
        R, cap = row_tok.shape
        self._check_dev(tokens, starts, lengths, rep, n_groups, row_tok, row_len, row_hash)
        old = self._i32(n)
        gh = torch.empty(n, dtype=torch.int64, device=self.device)
        check(self.lib.glb_match_rows(_ptr(tokens), _ptr(starts), _ptr(lengths), _ptr(rep), _ptr(n_groups), n, _ptr(row_tok),
                                      _ptr(row_len), _ptr(row_hash), R, cap, _ptr(old), _ptr(gh), self._stream()))
        return old, gh

    def kv_plan(self, group_of, rep, n_groups, old_row, lengths, n_rows, cap, by_context=False, stamps=None, call_no=0,
                table=None):
        """The block table of one step (glb_kv_plan).  `old_row`: the row every group's prefix sits in (match_rows'
        output), or with `by_context` every CONTEXT's row (the previous plan's `row_of_context`).  `stamps`: int64
        [n_rows], free rows are handed out longest unused first (and stamped `call_no`).  `table`: (row_tok, row_len,
        row_hash, group_hash, tokens, starts) - the rows' contents, rewritten for every group that holds a row.
        Returns a dict of int32 device tensors (include/glb.h names without the out_ prefix); `head` (8 words) is all
        the host has to read."""
        n = group_of.numel()
        self._check_dev(group_of, rep, n_groups, old_row, lengths, stamps)
        buf = torch.empty(8 * n + 4 * n_rows + 8, dtype=torch.int32, device=self.device)
        names_n = ("group_row", "logits_row", "rows_a", "ctx_a", "pos_a", "ctx_b", "rows_b", "row_of_context")
        names_r = ("copy_src", "copy_len", "ctx_of_row", "pos_of_row")
        out, o = {}, 0
        for k in names_n:
            out[k] = buf[o:o + n]
            o += n
        for k in names_r:
            out[k] = buf[o:o + n_rows]
            o += n_rows
        out["head"] = buf[o:o + 8]
        need = self.lib.glb_kv_plan_workspace(n, n_rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
        a = KvPlanArgs()
        a.struct_size = C.sizeof(KvPlanArgs)
        a.n, a.n_rows, a.cap = n, n_rows, cap
        a.group_of, a.rep, a.n_groups = group_of.data_ptr(), rep.data_ptr(), n_groups.data_ptr()
        a.old_row, a.old_row_by_context = old_row.data_ptr(), 1 if by_context else 0
        a.lengths = lengths.data_ptr()
        a.row_stamps = None if stamps is None else stamps.data_ptr()
        a.call_no = int(call_no)
        if table is not None:
            row_tok, row_len, row_hash, group_hash, tokens, starts = table
            self._check_dev(row_tok, row_len, row_hash, group_hash, tokens, starts)
            a.row_tok, a.row_len, a.row_hash = row_tok.data_ptr(), row_len.data_ptr(), row_hash.data_ptr()
            a.group_hash, a.tokens, a.starts = group_hash.data_ptr(), tokens.data_ptr(), starts.data_ptr()
        for k in names_n + names_r + ("head",):
            setattr(a, "out_" + k, out[k].data_ptr())
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        check(self.lib.glb_kv_plan(C.byref(a), self._stream()))
        return out

    def resample_systematic(self, log_weights, seed, offset):
        """Ancestors of a systematic resampling step over `log_weights` (the gathered population), int32 [n]
        (glb_resample_systematic); also returns logsumexp of the weights as a 1-element tensor."""
        n = log_weights.numel()
        need = self.lib.glb_resample_workspace(n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
        anc, stats = self._i32(n), self._f32(1)
        check(self.lib.glb_resample_systematic(_ptr(log_weights), n, seed, offset, _ptr(anc), _ptr(stats),
                                               _ptr(self._ws), self._ws.numel(), self._stream()))
        return anc, stats

    # ---- token -> byte trie masses ---------------------------------------------------------------------------
    def trie_masses(self, ws, flat, op=0, from_logprobs=False, lse=None, logit_scale=1.0, nodes=None, layout="rows"):
        """Token -> byte trie masses (glb_trie_masses).  ws: [B, >=V] rows of any supported element type - weights, or
        with from_logprobs log-probabilities, or with from_logprobs AND lse (float32 [B], the fused step's out_lse) raw
        LOGITS: the leaf weight is exp(x * logit_scale - lse[r]), no log-prob matrix in between.  Result:
          layout "rows"   float32 [B, n_nodes]                       (every node, row-major)
          nodes given     float32 [B, len(nodes)]                    (only the nodes asked for: int32 device tensor)
          layout "nodes"  float32 [n_nodes, pitch] view of the engine's scratch, pitch = B rounded up to 64: value of
                          node n for row r at [n, r]; valid until the next trie call (nothing is transposed back)."""
        if ws.dim() != 2 or ws.stride(1) != 1 or ws.dtype not in _DT:
            raise ValueError("weights must be [B, V] float32 / bfloat16 / float16 with unit inner stride")
        B = ws.shape[0]
        V, n_nodes = flat["vocab"], flat["n_nodes"]
        self._check_dev(lse, nodes)
        a = TrieArgs()
        a.struct_size = C.sizeof(TrieArgs)
        a.weights, a.dtype = ws.data_ptr(), _DT[ws.dtype]
        a.ld = ws.stride(0) if B > 1 else max(V, ws.stride(0))
        a.n_rows, a.vocab = B, V
        a.lse = None if lse is None else lse.data_ptr()
        a.logit_scale, a.from_logprobs, a.op = logit_scale, 1 if from_logprobs else 0, op
        a.n_nodes, a.n_levels = n_nodes, flat["n_levels"]
        ls = flat["level_start_host"]  # contiguous int32 NumPy array: it sizes the per-level launches
        a.leaf_node, a.level_start_host = flat["leaf_node"].data_ptr(), ls.ctypes.data
        a.level_nodes, a.child_ptr, a.child_idx = (flat[k].data_ptr() for k in ("level_nodes", "child_ptr", "child_idx"))
        need = self.lib.glb_trie_workspace_ex(B, n_nodes)
        if self._trie_ws is None or self._trie_ws.numel() < need:
            self._trie_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        a.workspace, a.workspace_bytes = self._trie_ws.data_ptr(), self._trie_ws.numel()
        out = None
        if nodes is not None:
            if nodes.dtype != torch.int32:
                raise TypeError("nodes must be int32")
            out = torch.empty((B, nodes.numel()), dtype=torch.float32, device=self.device)
            a.sel_nodes, a.n_sel, a.out_sel, a.out_sel_ld = nodes.data_ptr(), nodes.numel(), out.data_ptr(), out.stride(0)
        elif layout == "nodes":
            a.keep_node_major = 1
        elif layout == "rows":
            out = torch.empty((B, n_nodes), dtype=torch.float32, device=self.device)
            a.out, a.out_ld = out.data_ptr(), out.stride(0)
        else:
            raise ValueError(f"unknown layout {layout!r}")
        check(self.lib.glb_trie_masses(C.byref(a), self._stream()))
        if out is not None:
            return out
        pitch = (B + 63) // 64 * 64
        return self._trie_ws[: n_nodes * pitch * 4].view(torch.float32).view(n_nodes, pitch)

    def trie_rows(self, ws, plan, op=0, from_logprobs=False, lse=None, logit_scale=1.0, nodes=None, layout="rows", out=None,
                  out_slots=None):
        """Token -> byte trie masses with one row of a part of the trie resident in LDS (glb_trie_rows): the weights are
        read once, the result written once, row-major.  plan: `TokenByteTrie.plan_device_arrays()`.  ws as in
        `trie_masses`.  Result: layout "rows" float32 [B, n_nodes]; nodes given (int32 device tensor) [B, len(nodes)] - or,
        `nodes` int32 [B, K], every row's OWN nodes (negative: none, 0 out): only the parts of the trie that hold a row's
        nodes are read and reduced for that row;
        layout "slots" [B, n_slots] in the plan's slot numbering (plan["slot_of"]: node -> slot).  out_slots: float32
        [B, n_slots] to be filled with the slots IN ADDITION to the requested output (one call, the C entry point's several
        outputs at once)."""
        if ws.dim() != 2 or ws.stride(1) != 1 or ws.dtype not in _DT:
            raise ValueError("weights must be [B, V] float32 / bfloat16 / float16 with unit inner stride")
        B = ws.shape[0]
        V = plan["vocab"]
        self._check_dev(lse, nodes, out)
        pl = plan.get("_c")
        if pl is None:
            pl = TriePlan()
            pl.struct_size = C.sizeof(TriePlan)
            for k in ("n_parts", "n_top", "n_cut", "n_slots", "max_local", "top_base", "lds_bytes", "n_nodes"):
                setattr(pl, k, int(plan[k]))
            for k in ("desc", "idepth", "leaf_src", "leaf_local", "run_tab", "top_local", "slot_of", "cptr16", "inode16", "pn_local16"):
                setattr(pl, k, plan[k].data_ptr())
            if plan.get("sweep"):  # (the kernel that reads a row front to back: TokenByteTrie.plan(sweep=True))
                pl.tok_local16, pl.inode64 = plan["tok_local16"].data_ptr(), plan["inode64"].data_ptr()
                pl.lds_top_bytes, pl.vocab = int(plan["lds_top_bytes"]), int(plan["vocab"])
            plan["_c"] = pl
        a = TrieRowsArgs()
        a.struct_size = C.sizeof(TrieRowsArgs)
        a.weights, a.dtype = ws.data_ptr(), _DT[ws.dtype]
        a.ld = ws.stride(0) if B > 1 else max(V, ws.stride(0))
        a.n_rows, a.vocab = B, V
        a.lse = None if lse is None else lse.data_ptr()
        a.logit_scale, a.from_logprobs, a.op = logit_scale, 1 if from_logprobs else 0, op
        need = self.lib.glb_trie_rows_workspace(B, C.byref(pl))
        if need:
            if self._trie_ws is None or self._trie_ws.numel() < need:
                self._trie_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            a.workspace, a.workspace_bytes = self._trie_ws.data_ptr(), self._trie_ws.numel()
        per_row = nodes is not None and nodes.dim() == 2
        if nodes is not None:
            if nodes.dtype != torch.int32:
                raise TypeError("nodes must be int32")
            if per_row and (nodes.shape[0] != B or nodes.stride(1) != 1):
                raise ValueError(f"per-row nodes must be int32 [{B}, K] with unit inner stride")
            width = nodes.shape[1] if per_row else nodes.numel()
        elif layout == "rows":
            width = plan["n_nodes"]
        elif layout == "slots":
            width = plan["n_slots"]
        else:
            raise ValueError(f"unknown layout {layout!r}")
        if out is None:
            out = torch.empty((B, width), dtype=torch.float32, device=self.device)
        elif out.shape != (B, width) or out.dtype != torch.float32 or out.stride(1) != 1:
            raise ValueError(f"out must be float32 [{B}, {width}] with unit inner stride")
        if width == 0 or B == 0:
            return out
        if nodes is not None:
            a.sel_nodes, a.n_sel, a.out_sel, a.out_sel_ld = nodes.data_ptr(), width, out.data_ptr(), out.stride(0)
            a.sel_row_stride = max(nodes.stride(0), width) if per_row else 0
        elif layout == "rows":
            a.out_nodes, a.out_nodes_ld = out.data_ptr(), out.stride(0)
        else:
            a.out_slots, a.out_slots_ld = out.data_ptr(), out.stride(0)
        if out_slots is not None:
            if out_slots.shape != (B, plan["n_slots"]) or out_slots.dtype != torch.float32 or out_slots.stride(1) != 1:
                raise ValueError(f"out_slots must be float32 [{B}, {plan['n_slots']}] with unit inner stride")
            self._check_dev(out_slots)
            a.out_slots, a.out_slots_ld = out_slots.data_ptr(), out_slots.stride(0)
        check(self.lib.glb_trie_rows(C.byref(a), C.byref(pl), self._stream()))
        return out

    def trie_reduce(self, ws, flat, op=0, from_logprobs=False, out=None):
        """out[r, node] = sum / max of the weights of the tokens below `node` (glb_trie_reduce).  ws: float32 [B, >=V]
        device rows; flat: the trie's arrays (trie.TokenByteTrie.device_arrays: device tensors + the host level table)."""
        if ws.dim() != 2 or ws.stride(1) != 1 or ws.dtype != torch.float32:
            raise ValueError("weights must be float32 [B, V] with unit inner stride")
        B = ws.shape[0]
        V, n_nodes = flat["vocab"], flat["n_nodes"]
        if out is None:
            out = torch.empty((B, n_nodes), dtype=torch.float32, device=self.device)
        ls = flat["level_start_host"]  # contiguous int32 NumPy array: it sizes the per-level launches
        need = self.lib.glb_trie_workspace(B, n_nodes)  # node-major scratch for batches of 32 rows and more
        if need and (self._trie_ws is None or self._trie_ws.numel() < need):
            self._trie_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        check(self.lib.glb_trie_reduce(_ptr(ws), ws.stride(0) if B > 1 else max(V, ws.stride(0)), B, V, n_nodes,
                                       flat["n_levels"], _ptr(flat["leaf_node"]), C.c_void_p(ls.ctypes.data),
                                       _ptr(flat["level_nodes"]), _ptr(flat["child_ptr"]), _ptr(flat["child_idx"]), op,
                                       1 if from_logprobs else 0, _ptr(out), out.stride(0),
                                       _ptr(self._trie_ws) if need else None, need, self._stream()))
        return out
