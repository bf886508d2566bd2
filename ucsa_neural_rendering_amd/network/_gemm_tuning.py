"""rocBLAS / hipBLASLt solution choices for DeepLab's 1x1 convolutions.

In channels-last every 1x1 convolution (forward, dX, dW) is one GEMM over all
pixels of the batch; which library kernel runs it is the library's heuristic,
and for these tall-skinny shapes ([9600 or 38400] x [64..2048]) the heuristic
is up to 2x off the best solution (e.g. 9600 x 512 -> 256, fp32: 54 us default,
25 us tuned).  ``gemm_tuning/tunableop_gfx950.csv`` is PyTorch TunableOp's
result file for those shapes, tuned on an MI355X (tools/gemm_tune.sh) at the
benchmark's batch of 8 images of 240x320, fp32.  ``ensure()`` switches
TunableOp on in look-up-only mode (tuning disabled: nothing is timed or written at run time):
shapes that are not in the table run the default solution.  R-101 fp32 step:
41.5 -> 39.5 ms.  TunableOp costs host time per GEMM call (it builds and hashes a
signature string): the launch-bound bf16-autocast step got 2-4 ms SLOWER with
it (24.9 -> 27-29 ms), so ``use()`` -- called by ``DeepLabV3.forward`` --
switches it off while autocast is on.

Left alone when the user controls TunableOp through ``PYTORCH_TUNABLEOP_*``.
The file carries validator lines (PyTorch / ROCm / hipBLASLt / rocBLAS versions,
gfx arch): on another stack TunableOp ignores the entries.
"""
import os

TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))),
                     "gemm_tuning", "tunableop_gfx950.csv")
_done = False


def ensure():
    """Idempotent; returns the table path when it was installed."""
    global _done, _state
    if _done:
        return TABLE
    if any(k.startswith("PYTORCH_TUNABLEOP_") for k in os.environ):
        return None
    import torch
    if not (torch.cuda.is_available() and os.path.isfile(TABLE)):
        return None
    tun = torch.cuda.tunable
    tun.set_filename(TABLE, insert_device_ordinal=False)
    tun.tuning_enable(False)
    tun.enable(True)
    _done, _state = True, True
    return TABLE


_state = None


def table_matches():
    """{"matched": bool, "why": str}: do the table's validator lines (PyTorch,
    HIP, hipBLASLt, rocBLAS versions, gfx arch) equal the running stack's?
    TunableOp silently ignores every entry of a file whose validators differ;
    this says so out loud (bench.py's ``tuning_tables_matched``).  Needs a GPU."""
    import torch
    if not torch.cuda.is_available():
        return {"matched": False, "why": "no GPU"}
    if not os.path.isfile(TABLE):
        return {"matched": False, "why": "no shipped table"}
    want = {}
    with open(TABLE) as fh:
        for ln in fh:
            r = ln.strip().split(",")
            if r and r[0] == "Validator" and len(r) >= 3:
                want[r[1]] = ",".join(r[2:])
    try:
        have = {k: str(v) for k, v in torch.cuda.tunable.get_validators()}
    except Exception as e:  # noqa: BLE001
        return {"matched": False, "why": f"get_validators failed: {e!r}"}
    bad = [f"{k}: table {want[k]} / running {have.get(k)}" for k in want if have.get(k) != want[k]]
    if bad:
        return {"matched": False, "why": "; ".join(bad)}
    if not _done:
        return {"matched": False, "why": "validators equal, table not installed "
                                         "(PYTORCH_TUNABLEOP_* set, or no DeepLabV3 built yet)"}
    return {"matched": True, "why": "validators equal, look-up only"}


def use(flag):
    """Route GEMMs through the table (fp32) or straight to the library
    (autocast: launch-bound, see above).  No-op when ``ensure()`` did not
    install the table."""
    global _state
    if not _done or flag == _state:
        return
    import torch
    torch.cuda.tunable.enable(bool(flag))
    _state = bool(flag)
