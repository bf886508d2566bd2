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
value set by the user wins.  MIOpen APPENDS the configurations it has not seen
to the files it is pointed at, so it is never pointed at the package's own
(git-tracked) files: ``use_shipped_db`` works on a per-user copy, created
atomically (``mkdtemp`` + ``rename``: ranks starting together cannot see a
half-copied directory) and keyed by the shipped files' content.

The file names encode what the entries were tuned for: ``gfx950100`` = arch
gfx950 with 0x100 = 256 CUs, ``HIP.3_5_0_<tweak>`` = MIOpen's version.  On any
other device or MIOpen build MIOpen looks for differently named files and
ignores these; ``shipped_db_matches`` says whether that is the case, and
``default_cudnn_benchmark`` then keeps the reference's exhaustive search.
"""
import hashlib
import mmap
import os
import re
import shutil
import tempfile

_PKG_DB = os.path.join(os.path.dirname(os.path.realpath(__file__)), "miopen_db")
_NAME = re.compile(r"^(gfx[0-9a-f]+?)([0-9a-f]{2,3})\.HIP\.(\d+)_(\d+)_(\d+)_(.+?)\.u(?:f)?db\.txt$")
_COPY_TAG = "ucsa_neural_rendering_amd_miopen_db_"


def shipped_files():
    try:
        return sorted(f for f in os.listdir(_PKG_DB) if f.endswith(".txt"))
    except OSError:
        return []


def _content_key():
    h = hashlib.sha256()
    for f in shipped_files():
        h.update(f.encode())
        with open(os.path.join(_PKG_DB, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def _cache_root():
    for base in (os.environ.get("XDG_CACHE_HOME"),
                 os.path.join(os.path.expanduser("~"), ".cache"),
                 tempfile.gettempdir()):
        if not base:
            continue
        try:
            os.makedirs(base, exist_ok=True)
        except OSError:
            continue
        if os.access(base, os.W_OK):
            return base
    return None


def use_shipped_db():
    """Point MIOpen at a private copy of the shipped databases; returns the
    directory (None when there is nothing to ship or nowhere to copy to)."""
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    files = shipped_files()
    if not files:
        return None
    root = _cache_root()
    if root is None:
        return None
    copy = os.path.join(root, f"{_COPY_TAG}{os.getuid()}_{_content_key()}")
    if not os.path.isdir(copy):
        tmp = tempfile.mkdtemp(prefix=".ucsa_miopen_", dir=root)
        try:
            for f in files:
                shutil.copy2(os.path.join(_PKG_DB, f), os.path.join(tmp, f))
            os.rename(tmp, copy)        # atomic; loses the race quietly below
        except OSError:
            shutil.rmtree(tmp, ignore_errors=True)
            if not os.path.isdir(copy):
                return None
    os.environ["MIOPEN_USER_DB_PATH"] = copy
    return copy


def _miopen_tweak_in_library(tweak: str):
    """Is `tweak` (the build id MIOpen puts into its db file names) the one of
    the MIOpen library torch loads?  The string sits in the library's data
    section; None when the library cannot be found."""
    try:
        import torch
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libMIOpen.so")
        with open(lib, "rb") as fh:
            m = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                return m.find(tweak.encode()) >= 0
            finally:
                m.close()
    except (OSError, ValueError, ImportError):
        return None


def shipped_db_matches(device=None):
    """{"matched": bool, "why": str}: do the shipped find-db / perf-db names
    match the running device (arch, CU count) and MIOpen build?  Needs a GPU
    (``matched`` False with the reason otherwise)."""
    files = shipped_files()
    if not files:
        return {"matched": False, "why": "no shipped db"}
    try:
        import torch
        if not torch.cuda.is_available():
            return {"matched": False, "why": "no GPU"}
        props = torch.cuda.get_device_properties(device if device is not None else
                                                 torch.cuda.current_device())
        arch = props.gcnArchName.split(":")[0]
        cus = int(props.multi_processor_count)
        v = int(torch.backends.cudnn.version() or 0)
    except Exception as e:  # noqa: BLE001 - report, never raise at import sites
        return {"matched": False, "why": f"device query failed: {e!r}"}
    have = (v // 1000000, v // 1000 % 1000, v % 1000)
    for f in files:
        m = _NAME.match(f)
        if not m:
            return {"matched": False, "why": f"unparsed db name {f}"}
        f_arch, f_cu, mj, mn, pt, tweak = m.groups()
        if f_arch != arch or int(f_cu, 16) != cus:
            return {"matched": False,
                    "why": f"db is for {f_arch} / {int(f_cu, 16)} CUs, device is {arch} / {cus}"}
        if (int(mj), int(mn), int(pt)) != have:
            return {"matched": False,
                    "why": f"db is for MIOpen {mj}.{mn}.{pt}, running {have[0]}.{have[1]}.{have[2]}"}
        tw = _miopen_tweak_in_library(tweak)
        if tw is False:
            return {"matched": False, "why": f"MIOpen build id {tweak} not in the loaded library"}
    return {"matched": True, "why": f"{arch}, {cus} CUs, MIOpen {have[0]}.{have[1]}.{have[2]}"}


def shipped_db_in_use():
    cur = os.environ.get("MIOPEN_USER_DB_PATH", "")
    shipped = set(shipped_files())
    try:
        have = set(os.listdir(cur)) if cur else set()
    except OSError:
        have = set()
    return bool(shipped) and shipped <= have and _COPY_TAG in os.path.basename(cur.rstrip("/"))


_default = None


def default_cudnn_benchmark():
    """What ``torch.backends.cudnn.benchmark`` should be when the experiment
    does not say: the reference sets it (scripts/train_joint.py) so that MIOpen
    searches its solvers exhaustively -- which MIOpen does AGAIN in every new
    process (the naive direct kernels included: ~18 s for a DeepLab step's
    configurations) even when the result is already in its databases.  False
    only when the shipped databases are the ones MIOpen reads AND they were
    tuned for the running device and MIOpen build (``shipped_db_matches``):
    PyTorch then asks MIOpen for the recorded best solver, tuned parameters
    included, and only searches -- once, non-exhaustively -- for
    configurations the databases do not hold.  Otherwise True, the
    reference's setting."""
    global _default
    if _default is None:
        _default = not (shipped_db_in_use() and shipped_db_matches()["matched"])
    return _default
