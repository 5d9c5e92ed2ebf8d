"""Which of the GEMM library's own solutions PyTorch runs for a shape (torch.cuda.tunable, "TunableOp").

The transformer forward of this backend is PyTorch's; its time is GEMMs (82 % of a re-encoding GPT-2 step, two thirds of a
KV-row step), and for every (transposes, m, n, k, leading dimensions) rocBLAS / hipBLASLt hold dozens of kernels of which
the default heuristic picks one.  TunableOp times them once per shape and records the winner; a recorded file is tied to
the build of PyTorch, HIP, rocBLAS, hipBLASLt and the GPU it was made on (the file's "Validator" lines - PyTorch refuses a
file whose validators differ and runs the default solutions).

    use_recorded()      run the shapes recorded in tuned/<arch>.csv with their recorded solution; every other shape, and
                        every shape when the file does not fit this installation, by the library's default.  No tuning,
                        no extra time.  Measured on the shapes of `bench.py` (1024 particles, GPT-2-small, fp32): 19.2
                        -> 17.8 ms a re-encoding step, 3.37 -> 3.22 ms a KV-row step.
    record(path)        tune every new shape this process meets (seconds a shape, once) and write `path` at exit:
                        how tuned/<arch>.csv was made (tools/record_gemm_tuning.sh) and how to add an application's shapes.

Process-wide switches of PyTorch, so the backend touches them only when asked: `AsyncAmdLM(gemms="recorded")` /
`load_model_by_name(name, llm_opts={"gemms": "recorded"})` takes the recorded solutions (`acquire()`), `AsyncAmdLM.close()`
gives them back (`release()`: the last holder switches TunableOp off again) - the option `bench.py --gemms recorded` goes
through, so that a bench line's `value` is one a user of the backend reproduces with that option; the default
(gemms="library") leaves PyTorch alone.  The solutions are the library's kernels at the tensors' own precision; what
changes is the order of a GEMM's additions (the last bits of the logits), as between any two library versions."""
import os

import torch

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned")


def recorded_file(device=None):
    """tuned/<arch>.csv for the device's architecture (gfx950: MI355X), or None."""
    if not torch.cuda.is_available():
        return None
    arch = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).gcnArchName.split(":")[0]
    path = os.path.join(_DIR, f"{arch}.csv")
    return path if os.path.exists(path) else None


def use_recorded(path=None, device=None):
    """Run recorded shapes with their recorded solutions.  Returns the number of shapes taken over (0: no file for this
    architecture, or one made by another build of the libraries - the library's defaults stay in charge)."""
    import torch.cuda.tunable as tunable

    path = path or recorded_file(device)
    if path is None:
        return 0
    n = 0
    try:
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.record_untuned_enable(False)
        if tunable.read_file(path):
            n = len(tunable.get_results())
    except Exception:  # (a PyTorch whose TunableOp differs: the defaults stay in charge)
        n = 0
    if n == 0:
        try:
            tunable.enable(False)
        except Exception:
            pass
    return n


_holders = 0
_held_shapes = 0


def acquire(device=None):
    """use_recorded() for one more holder (AsyncAmdLM(gemms="recorded")); returns the number of shapes taken over."""
    global _holders, _held_shapes
    if _holders == 0:
        _held_shapes = use_recorded(device=device)
    _holders += 1
    return _held_shapes


def release():
    """One holder less; the last one switches TunableOp off again (PyTorch's state as this module found it)."""
    global _holders, _held_shapes
    if _holders == 0:
        return
    _holders -= 1
    if _holders == 0:
        _held_shapes = 0
        try:
            off()
        except Exception:
            pass


def record(path):
    """Tune every shape this process meets that `path` does not hold yet; PyTorch writes `path` when the process ends."""
    import torch.cuda.tunable as tunable

    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(path, insert_device_ordinal=False)
    if os.path.exists(path):
        tunable.read_file(path)


def off():
    import torch.cuda.tunable as tunable

    tunable.enable(False)
