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
    assert a == b == os.path.join(ROOT, "ucsa_neural_rendering_amd", "miopen_db")


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
