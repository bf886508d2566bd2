"""CPU-only: the C-ABI library builds for gfx950, loads, and exports exactly
the symbols include/ucsa_hip.h declares; the ctypes table covers all of them.
No kernel is launched (there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ucsa_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ucsa_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from ucsa_neural_rendering_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_header_symbols_are_exported(built_lib):
    syms = declared_symbols()
    assert "ucsa_render_fwd" in syms and "ucsa_near_far_from_aabb" in syms
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ucsa_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared but not exported: {missing}"
    extra = sorted(exported - set(syms))
    assert not extra, f"exported but not declared in the header: {extra}"


def test_ctypes_table_covers_header(built_lib):
    assert sorted(built_lib.SIGNATURES) == declared_symbols()
    l = built_lib.lib()  # resolves every symbol, raises otherwise
    assert l.ucsa_version() == 100
    assert l.ucsa_error_string(0) == b"ok"
    assert b"argument #3" in l.ucsa_error_string(-1003)


def test_grid_init_host_function_matches_survey_table(built_lib):
    g = built_lib.make_grid(4.0, 16, 19, 16, 2.0**0.6)
    res = [g.level[i].res for i in range(16)]
    assert res == [16, 25, 37, 56, 85, 128, 195, 295, 446, 676, 1024, 1553,
                   2353, 3566, 5405, 8192]
    assert g.total_entries == 6537456
    assert [g.level[i].hashed for i in range(16)] == [0] * 4 + [1] * 12
    assert g.level[5].scale == 127.0 and g.level[15].scale == 8191.0
    # argument validation returns UCSA_ERR_ARG - index, never crashes
    bad = built_lib.Grid()
    rc = built_lib.lib().ucsa_grid_init(ctypes.byref(bad), -1.0, 16, 19, 16, 1.5)
    assert rc == -1001


def test_null_arguments_are_rejected_before_any_launch(built_lib):
    l = built_lib.lib()
    assert l.ucsa_near_far_from_aabb(None, None, None, 8, 0.2, None, None, None) == -1000
    assert l.ucsa_sample_coarse(None, None, None, 8, 16, None, None) == -1000
    assert l.ucsa_mlp_pack(7, None, None, 40, None) == -1000
    assert l.ucsa_resample(None, None, None, 8, 16, 16, 1.0, None, None) == -1000


def test_product_has_no_oracle_import():
    """The shipped package must never route through the CPU oracle."""
    pat = re.compile(r"^\s*(from|import)\s+(oracle|tests)\b", re.M)
    for sub in ("ucsa_neural_rendering_amd", "nr4seg", "tools", "scripts"):
        for dp, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith(".py"):
                    txt = open(os.path.join(dp, f)).read()
                    assert not pat.search(txt), os.path.join(dp, f)


def test_reference_import_paths_resolve():
    """The reference's own import lines work against the alias package
    (scripts/train_joint.py:9-17, joint_train_lightning_net.py:12-27,
    joint_train_data_module.py:8 of the reference)."""
    from nr4seg import ROOT_DIR  # noqa: F401
    from nr4seg.dataset import ScanNetNGPJoint  # noqa: F401
    from nr4seg.dataset.ngp_utils import get_rays, nerf_matrix_to_ngp  # noqa: F401
    from nr4seg.lightning import JointTrainDataModule, JointTrainLightningNet  # noqa: F401
    from nr4seg.nerf.network_tcnn_semantics import SemanticNeRFNetwork  # noqa: F401
    from nr4seg.nerf.raymarching import raymarching
    from nr4seg.nerf.renderer_semantics import SemanticNeRFRenderer  # noqa: F401
    from nr4seg.network import DeepLabV3  # noqa: F401
    from nr4seg.utils import flatten_dict, load_yaml  # noqa: F401
    from nr4seg.utils.metrics import SemanticsMeter  # noqa: F401
    for name in ("near_far_from_aabb", "march_rays_train", "composite_rays_train",
                 "composite_rays_train_semantics", "march_rays", "composite_rays",
                 "composite_rays_semantics", "compact_rays"):
        assert callable(getattr(raymarching, name)), name


def test_ctypes_structures_match_the_header_layout(built_lib, tmp_path):
    """The structs that cross the boundary by pointer (ucsa_grid, ucsa_aug_params,
    ucsa_train_buffers, ucsa_train_packs): size and field offsets of the ctypes
    mirrors against what a C compiler makes of include/ucsa_hip.h."""
    structs = {"ucsa_grid": built_lib.Grid, "ucsa_grid_level": built_lib.GridLevel,
               "ucsa_aug_params": built_lib.AugParams,
               "ucsa_train_buffers": built_lib.TrainBuffers,
               "ucsa_train_packs": built_lib.TrainPacks}
    lines = []
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ucsa_hip.h"\n'
                   "int main(void) {\n" + "\n".join(lines) + "\nreturn 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    c_layout = dict(l.split() for l in out.strip().splitlines())
    for cname, ct in structs.items():
        assert int(c_layout[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(c_layout[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)


def test_every_environment_switch_of_the_library_is_in_its_table_and_documented():
    """the library reads its UCSA_* switches from ONE table, once
    per process (csrc/render.hip kEnvNames / ucsa_getenv; an unknown name asserts).
    Every name the sources look up is in the table, every table entry is looked up,
    and INTEGRATION.md's "Environment variables" section names each of them."""
    import glob
    import re
    src = os.path.join(ROOT, "ucsa_neural_rendering_amd", "csrc")
    used = set()
    for f in glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")):
        text = open(f).read()
        assert not re.search(r"(?<![_A-Za-z])getenv\(", text.replace("ucsa_getenv(", "")) \
            or f.endswith("render.hip"), f"{f}: a plain getenv() outside the table"
        used |= set(re.findall(r'(?:ucsa_getenv|env_u|simple_gather_below)\(\s*"(UCSA_[A-Z0-9_]+)"', text))
    r = open(os.path.join(src, "render.hip")).read()
    table = set(re.findall(r'"(UCSA_[A-Z0-9_]+)"',
                           r[r.index("kEnvNames[] = {"):r.index("constexpr int kEnvCount")]))
    assert used == table, (used - table, table - used)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("### Environment variables"):doc.index("## 7. Entry-point index")]
    missing = [n for n in sorted(table) if n not in sec
               and not any(n.startswith(p.rstrip("*")) for p in re.findall(r"`(UCSA_[A-Z_]+)=", sec) if False)]
    # (INTEGRATION abbreviates the UCSA_ENC_SIMPLE family as `UCSA_ENC_SIMPLE=n` / `_H` / `_RAYS`)
    missing = [n for n in missing if not n.startswith("UCSA_ENC_SIMPLE")]
    assert not missing, missing
