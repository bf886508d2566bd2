"""The tuned MIOpen databases ship with the package and are what MIOpen is
pointed at (ucsa_neural_rendering_amd/_miopen_db.py)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)


def test_package_points_miopen_at_the_shipped_db():
    env = {k: v for k, v in os.environ.items() if k != "MIOPEN_USER_DB_PATH"}
    r = _run("import os, ucsa_neural_rendering_amd as u; "
             "print(os.environ['MIOPEN_USER_DB_PATH']); print(u.MIOPEN_DB_PATH)", env)
    assert r.returncode == 0, r.stderr
    a, b = r.stdout.split()
    # a per-user COPY (MIOpen appends to what it is pointed at: never the
    # package's tracked files), holding exactly the shipped files
    pkg = os.path.join(ROOT, "ucsa_neural_rendering_amd", "miopen_db")
    assert a == b and os.path.realpath(a) != os.path.realpath(pkg)
    assert "ucsa_neural_rendering_amd_miopen_db_" in os.path.basename(a)
    for f in os.listdir(pkg):
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(pkg, f), "rb").read()


def test_copy_is_atomic_under_concurrent_imports(tmp_path):
    """Eight processes importing the package at once into an empty cache: all
    end on the same complete directory, no half-copied state is visible."""
    env = {k: v for k, v in os.environ.items() if k != "MIOPEN_USER_DB_PATH"}
    env["XDG_CACHE_HOME"] = str(tmp_path)
    code = ("import os, ucsa_neural_rendering_amd as u; d = u.MIOPEN_DB_PATH; "
            "print(d, sorted((f, os.path.getsize(os.path.join(d, f))) for f in os.listdir(d)))")
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, text=True) for _ in range(8)]
    outs = {p.communicate(timeout=300)[0] for p in procs}
    assert all(p.returncode == 0 for p in procs) and len(outs) == 1, outs
    assert str(tmp_path) in next(iter(outs))
    assert [d for d in os.listdir(tmp_path) if d.startswith(".ucsa_miopen_")] == []


def test_a_user_setting_wins():
    env = dict(os.environ, MIOPEN_USER_DB_PATH="/tmp/somewhere_else")
    r = _run("import os, ucsa_neural_rendering_amd as u; print(u.MIOPEN_DB_PATH)", env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "/tmp/somewhere_else"


def test_db_holds_the_deeplab_configurations():
    d = os.path.join(ROOT, "ucsa_neural_rendering_amd", "miopen_db")
    fdb = glob.glob(os.path.join(d, "gfx950*.ufdb.txt"))
    assert len(fdb) == 1, "one find-db for gfx950 (256 CUs)"
    keys = [ln.split("=")[0] for ln in open(fdb[0]) if "=" in ln]
    # the dilated 3x3 convolutions of layer3 / layer4 / ASPP at the bench's
    # batch of 8 images 240x320 (output stride 8 -> 30x40 maps), all three
    # directions, fp32 and bf16, channels-last
    for dil in ("2x2", "4x4", "12x12", "24x24", "36x36"):
        for dt in ("FP32", "BF16"):
            for direction in "FBW":
                assert any(f"-30-40-8-{dil}-1x1-{dil}-0-NHWC-NHWC-NHWC-{dt}-{direction}" in k
                           for k in keys), (dil, dt, direction)


def test_gemm_table_ships_and_names_the_1x1_convolution_shapes():
    path = os.path.join(ROOT, "ucsa_neural_rendering_amd", "gemm_tuning", "tunableop_gfx950.csv")
    rows = [ln.strip().split(",") for ln in open(path) if ln.strip()]
    assert any(r[0] == "Validator" and r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950")
               for r in rows)
    keys = {(r[0], r[1]) for r in rows if r[0] != "Validator"}
    # layer3's bottleneck 1x1 convolutions at 8 x 30 x 40 pixels: forward (TN),
    # dX (NN) and dW (NT); fp32 only (TunableOp is switched off under autocast)
    for dt in ("float",):
        assert (f"GemmTunableOp_{dt}_TN", "tn_256_9600_1024_ld_1024_1024_256") in keys
        assert (f"GemmTunableOp_{dt}_NN", "nn_1024_9600_256_ld_1024_256_1024") in keys
        assert (f"GemmTunableOp_{dt}_NT", "nt_1024_256_9600_ld_1024_256_1024") in keys


def test_gemm_table_is_left_alone_when_the_user_controls_tunableop(monkeypatch):
    from ucsa_neural_rendering_amd.network import _gemm_tuning
    monkeypatch.setenv("PYTORCH_TUNABLEOP_ENABLED", "0")
    monkeypatch.setattr(_gemm_tuning, "_done", False)
    assert _gemm_tuning.ensure() is None


def test_cudnn_benchmark_default_needs_a_matching_device():
    """Without a GPU (or on a device / MIOpen build the shipped names were not
    tuned for) the reference's cudnn.benchmark = True stays; the lookup-only
    mode on a matching MI355X is asserted by tests/test_gpu_deeplab_parity.py."""
    code = ("from ucsa_neural_rendering_amd._miopen_db import default_cudnn_benchmark as d, "
            "shipped_db_matches as m; import ucsa_neural_rendering_amd; print(d(), m()['matched'])")
    env = {k: v for k, v in os.environ.items() if k != "MIOPEN_USER_DB_PATH"}
    r = _run(code, env)
    assert r.returncode == 0 and r.stdout.split() == ["True", "False"], r.stdout + r.stderr
    r = _run(code, dict(env, MIOPEN_USER_DB_PATH="/tmp/not_ours"))
    assert r.returncode == 0 and r.stdout.split()[0] == "True", r.stdout + r.stderr


def test_db_name_parse():
    from ucsa_neural_rendering_amd import _miopen_db as m
    for f in m.shipped_files():
        g = m._NAME.match(f)
        assert g and g.group(1) == "gfx950" and int(g.group(2), 16) == 256
        assert g.groups()[2:5] == ("3", "5", "0")
