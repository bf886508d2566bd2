"""scripts/cl_deeplab.py's stage plan against the reference's loop
(``/root/reference/scripts/cl_deeplab.py:62-86``, restated here as the expected
values): scene order, stage names, which checkpoint each stage loads and
where ``load_pretrain`` applies.  CPU only."""
import os

import pytest

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


def test_trainer_prefetch_thread_keeps_order_and_surfaces_errors():
    """`trainer: {prefetch: N}` (the role of the reference's DataLoader workers):
    batches come from a background thread in the loader's order, `limit_batches`
    still applies, an exception in the loader is re-raised in the consumer, and a
    consumer that stops early does not leave the thread blocked."""
    import threading
    import time
    import torch
    from ucsa_neural_rendering_amd.lightning.trainer import Trainer
    tr = Trainer(max_epochs=1, device="cpu", prefetch=2)
    data = [{"a": torch.ones(1) * k, "H": torch.tensor([k])} for k in range(7)]
    got = list(tr._batches(data))
    assert [i for i, _ in got] == list(range(7)) and [int(b["a"]) for _, b in got] == list(range(7))
    tr.limit_batches = 3
    assert [i for i, _ in tr._batches(data)] == [0, 1, 2]
    tr.limit_batches = None

    def bad():
        yield data[0]
        raise RuntimeError("boom")

    with pytest.raises(RuntimeError, match="boom"):
        list(tr._batches(bad()))
    g = tr._batches([dict(d) for d in data] * 20)
    next(g)
    g.close()
    time.sleep(0.6)
    assert not [t for t in threading.enumerate() if t.name == "ucsa-prefetch" and t.is_alive()]
