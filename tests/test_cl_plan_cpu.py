"""scripts/cl_deeplab.py's stage plan against the reference's loop
(``/root/reference/scripts/cl_deeplab.py:62-86``, restated here as the expected
values): scene order, stage names, which checkpoint each stage loads and
where ``load_pretrain`` applies.  CPU only."""
import os

from scripts.cl_deeplab import SCENE_ORDER, stage_plan


def test_scene_order_is_the_references():
    assert SCENE_ORDER == [f"scene000{i}_00" for i in range(10)]


def test_stage_plan_matches_reference_loop():
    exp = {"general": {"checkpoint_load": "ckpts/best-epoch=143-step=175536.ckpt"}}
    plan = stage_plan(exp, "run", "experiments")
    assert len(plan) == 10
    for i, st in enumerate(plan):
        assert st["name"] == f"run/stage_{i}"
        assert st["scenes"] == SCENE_ORDER[:i + 1]          # grows by one per stage
        assert st["load_from_checkpoint"] is True and st["resume_from_checkpoint"] is False
        if i == 0:
            assert st["load_pretrain"] is True
            assert st["checkpoint_load"] == "ckpts/best-epoch=143-step=175536.ckpt"
        else:
            assert st["load_pretrain"] is False
            assert st["checkpoint_load"] == os.path.join("experiments", "run",
                                                         f"stage_{i - 1}", "deeplab.ckpt")
    assert len(stage_plan(exp, "run", "experiments", 3)) == 3
