"""MIOpen solver choices for DeepLab's 3x3 / 7x7 convolutions, tuned on an
MI355X and shipped with the package (``miopen_db/``: MIOpen's user find-db and
perf-db text files, produced by tools/miopen_tune.sh).

Without them every fresh machine re-runs MIOpen's solver search on the first
step (each applicable solver once per convolution configuration, the naive
direct kernels included: 15-30 s per DeepLab mode) and ends on the untuned
implicit-GEMM parameters (R-101 fp32 step 44.7 ms); with them the first step
compiles the chosen kernels only (4-5 s) and the step takes 41.6 ms.

MIOpen reads ``MIOPEN_USER_DB_PATH`` when it first opens its databases, i.e.
at the first convolution, so setting it at package import is early enough.  A
value set by the user wins.  MIOpen appends the configurations it has not seen
to the same files, so the directory must be writable: a read-only install gets
a private copy under the temp directory.
"""
import os
import shutil
import tempfile

_PKG_DB = os.path.join(os.path.dirname(os.path.realpath(__file__)), "miopen_db")


def use_shipped_db():
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    path = _PKG_DB
    if not os.path.isdir(path):
        return None
    if not os.access(path, os.W_OK):
        copy = os.path.join(tempfile.gettempdir(),
                            f"ucsa_neural_rendering_amd_miopen_db_{os.getuid()}")
        if not os.path.isdir(copy):
            shutil.copytree(path, copy)
        path = copy
    os.environ["MIOPEN_USER_DB_PATH"] = path
    return path


def default_cudnn_benchmark():
    """What ``torch.backends.cudnn.benchmark`` should be when the experiment
    does not say: the reference sets it (scripts/train_joint.py) so that MIOpen
    searches its solvers exhaustively -- which MIOpen does AGAIN in every new
    process (the naive direct kernels included: ~18 s for a DeepLab step's
    configurations) even when the result is already in its databases.  With
    the shipped databases in use the search has been done: False (PyTorch then
    asks MIOpen for the recorded best solver, tuned parameters included, and
    only searches -- once, non-exhaustively -- for configurations the
    databases do not hold).  Without them: True, the reference's setting."""
    cur = os.environ.get("MIOPEN_USER_DB_PATH", "")
    shipped = {os.path.basename(f) for f in os.listdir(_PKG_DB)} if os.path.isdir(_PKG_DB) else set()
    try:
        have = set(os.listdir(cur)) if cur else set()
    except OSError:
        have = set()
    in_use = bool(shipped) and shipped <= have and (
        os.path.realpath(cur) == _PKG_DB or "ucsa_neural_rendering_amd_miopen_db_" in cur)
    return not in_use
