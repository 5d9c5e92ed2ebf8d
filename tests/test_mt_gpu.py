"""torch's CPU generator on the device (glb_mt19937_exponential_rows, csrc/glb_mt.hip) against the serial host stream the
reference runs (glb_mt19937_exponential_f32 == the oracle == torch, tests/test_oracle.py, tests/test_mt_cpu.py): bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

import genlm_backend_amd  # noqa: F401
from genlm_backend_amd import _lib

pytestmark = pytest.mark.gpu


def _host_stream(seed, n):
    lib = _lib.load()
    st = _lib.MT19937()
    lib.glb_mt19937_seed(C.byref(st), seed)
    out = np.empty(n, np.float32)
    assert lib.glb_mt19937_exponential_f32(C.byref(st), out.ctypes.data_as(C.c_void_p), n) == 0
    return out


def test_device_rows_are_torchs_stream_directly(engine):
    """The chain's last link closed ON the GPU box: rows made by the device generator, flattened, against the golden
    torch itself produced (tests/golden/torch_kernel_ops.npz, mt::seed99_first4096 =
    torch.empty(4096).exponential_(1, generator=manual_seed(99)), oracle/make_goldens.py) - no host stream of this
    library in between - at three row widths (the stream is one sequence however it is cut into rows), and against torch
    run here on the host for a seed and a position no fixture holds."""
    import os

    from genlm_backend_amd.engine import DeviceRng

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_kernel_ops.npz"))["mt::seed99_first4096"]
    for V, rows in ((4096, 1), (64, 64), (311, 13)):
        got = DeviceRng(engine, 99, V).rows(rows)
        torch.cuda.synchronize()
        flat = got.cpu().numpy().reshape(-1)
        assert np.array_equal(flat.view(np.uint32), gold[:flat.size].view(np.uint32)), (V, rows)
    g = torch.Generator()
    g.manual_seed(31337)
    want = torch.empty(40 * 50257).exponential_(1, generator=g).numpy().reshape(40, 50257)
    rng = DeviceRng(engine, 31337, 50257)
    first, second = rng.rows(33), rng.rows(7)  # (33 rows: both jump launches; the second call starts where the first ended)
    torch.cuda.synchronize()
    assert np.array_equal(first.cpu().numpy().view(np.uint32), want[:33].view(np.uint32))
    assert np.array_equal(second.cpu().numpy().view(np.uint32), want[33:].view(np.uint32))


@pytest.mark.parametrize("V,rows", [(50257, 70), (4099, 33), (1, 40), (128256, 9), (311, 1)])
def test_rows_equal_the_serial_stream(engine, V, rows):
    """Several calls in a row: the stream moves on by what each call consumed; rows beyond one polynomial level (rows > 32)
    take both jump launches."""
    from genlm_backend_amd.engine import DeviceRng

    rng = DeviceRng(engine, 1234, V)
    want = _host_stream(1234, 3 * rows * V).reshape(3 * rows, V)
    for call in range(3):
        got = rng.rows(rows)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want[call * rows:(call + 1) * rows].view(np.uint32)), call


@pytest.mark.parametrize("V,total,mine", [(311, 200, 40), (50257, 130, 33), (4099, 64, 1)])
def test_a_shard_reads_only_its_rows_of_the_global_stream(engine, V, total, mine):
    """A sharded population's parity draws (DeviceSIS._parity_noise_sharded): `total` stream rows are dealt over all ranks,
    this rank's `mine` output rows take scattered ones of them and the stream moves on by a count that lives on the device.
    The jump launches make only the windows somebody here reads (mt_need_kernel's flags: the rows named, and the row the
    stream stands at afterwards); the rows and the next call's start are the serial stream's."""
    from genlm_backend_amd.engine import DeviceRng

    rng = DeviceRng(engine, 4242, V)
    rs = np.random.default_rng(V)
    n_draw = [total - 7, total]
    want = _host_stream(4242, (n_draw[0] + n_draw[1]) * V).reshape(-1, V)
    base = 0
    for call in range(2):
        slot = rs.choice(n_draw[call], mine, replace=False).astype(np.int32)
        if mine > 3:
            slot[1] = -1  # a particle of this rank that draws nothing
        got = rng.rows(mine, row_slot=torch.from_numpy(slot).to(engine.device),
                       n_draw=torch.tensor(n_draw[call], dtype=torch.int32, device=engine.device), max_draw=total)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        for i, sl in enumerate(slot):
            exp = np.ones(V, np.float32) if sl < 0 else want[base + sl]
            assert np.array_equal(got[i].view(np.uint32), exp.view(np.uint32)), (call, i)
        base += n_draw[call]


def test_slots_counts_on_the_device_and_rows_of_ones(engine):
    """Output rows take stream rows in any order (row_slot), particles that draw nothing get ones, and the stream moves on
    by a count that lives on the device - the resolution order of hf.py:285-288 with inactive particles skipped."""
    from genlm_backend_amd.engine import DeviceRng

    V, N = 50257, 48
    rng = DeviceRng(engine, 7, V)
    rs = np.random.default_rng(3)
    want = _host_stream(7, (3 * N + 1) * V).reshape(3 * N + 1, V)
    used = 0
    for call in range(3):
        act = rs.random(N) < 0.7
        n_act = int(act.sum())
        perm = rs.permutation(n_act)
        slot = np.full(N, -1, np.int32)
        slot[np.nonzero(act)[0]] = perm
        if call == 2:  # nothing is consumed when nobody draws
            slot[:] = -1
            n_act = 0
        got = rng.rows(N, row_slot=torch.from_numpy(slot).to(engine.device),
                       n_draw=torch.tensor(n_act, dtype=torch.int32, device=engine.device), max_draw=N)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        for i in range(N):
            if slot[i] < 0:
                assert (got[i] == 1.0).all()
            else:
                assert np.array_equal(got[i].view(np.uint32), want[used + slot[i]].view(np.uint32)), (call, i)
        used += n_act
    # one shared row per step (base.py:148-179: every sequence of batch_sample is seeded alike): DeviceSampler's use
    one = rng.rows(1).cpu().numpy()
    assert np.array_equal(one[0].view(np.uint32), want[used].view(np.uint32))


def test_headline_size_and_a_later_position(engine):
    """1024 particles x 50257 (BASELINE config 2), the second step of a run: 206 MB of noise that used to be 0.8 s of host
    time + a PCIe copy."""
    from genlm_backend_amd.engine import DeviceRng

    V, N = 50257, 1024
    rng = DeviceRng(engine, 1234, V)
    rng.rows(N)
    got = rng.rows(N)
    torch.cuda.synchronize()
    want = _host_stream(1234, 2 * N * V)[N * V:].reshape(N, V)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_rows_generated_ahead_on_a_side_stream_are_the_same_rows(engine):
    """DeviceRng(ahead=True): prefetch() generates the next call's rows in stream order on a side stream, rows() then only
    copies what row_slot names and moves the position (glb_mt_rows_args: reuse_windows, rows_from).  Same floats as the
    serial stream whatever the steps draw: all, some, none; a call of another shape drops what was made ahead; reset()."""
    from genlm_backend_amd.engine import DeviceRng

    V, N = 4099, 70
    rs = np.random.default_rng(5)
    want = _host_stream(99, (6 * N + 8) * V).reshape(6 * N + 8, V)
    for trial in range(2):
        rng = DeviceRng(engine, 99, V, ahead=True) if trial == 0 else rng
        if trial == 1:
            rng.reset()  # (with rows made ahead still pending)
        used = 0
        for call in range(5):
            act = rs.random(N) < (1.0, 0.6, 0.0, 0.9, 0.5)[call]
            n_act = int(act.sum())
            slot = np.full(N, -1, np.int32)
            slot[np.nonzero(act)[0]] = rs.permutation(n_act)
            got = rng.rows(N, row_slot=torch.from_numpy(slot).to(engine.device),
                           n_draw=torch.tensor(n_act, dtype=torch.int32, device=engine.device), max_draw=N)
            rng.prefetch()
            assert rng._pre is not None  # (something is on its way on the side stream)
            busy = torch.randn((512, 512), device=engine.device) @ torch.randn((512, 512), device=engine.device)  # (the caller's stream goes on)
            got = got.cpu().numpy()
            for i in range(N):
                if slot[i] < 0:
                    assert (got[i] == 1.0).all()
                else:
                    assert np.array_equal(got[i].view(np.uint32), want[used + slot[i]].view(np.uint32)), (trial, call, i)
            used += n_act
        # another shape: what was made ahead (70 rows) does not fit and is dropped; the stream stands where it stood
        one = rng.rows(8)
        rng.prefetch()
        assert np.array_equal(one.cpu().numpy().view(np.uint32), want[used:used + 8].view(np.uint32))
        again = rng.rows(8)  # (served from the rows made ahead)
        assert np.array_equal(again.cpu().numpy().view(np.uint32), want[used + 8:used + 16].view(np.uint32))
        del busy


def test_argument_errors(engine):
    from genlm_backend_amd.engine import DeviceRng

    rng = DeviceRng(engine, 1, 100)
    with pytest.raises(ValueError):
        rng.rows(4, row_slot=torch.zeros(3, dtype=torch.int32, device=engine.device))
    a = _lib.MtRowsArgs()
    a.struct_size = 4
    assert engine.lib.glb_mt19937_exponential_rows(C.byref(a), None) == _lib.GLB_EINVAL
