"""CPU: the jsonnet config surface (eavqa_amd.utils.config_system) on this build's configs and - when the reference
checkout is present (build container only) - on every config file of the reference."""
import glob
import os

import pytest

from eavqa_amd.utils import config_system as cs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/configs"


def test_own_configs_evaluate_and_select_classes_by_name():
    cfg = cs.load_config(os.path.join(ROOT, "configs", "vqa2", "clip_cap_gpt2_large.jsonnet"), opts=["train.lr=0.0003", "seed=7"])
    assert cfg.model_config.ModelClass == "ClipCaptionPrefix"
    assert dict(cfg.model_config.model_args) == dict(prefix_length=10, clip_length=10, prefix_size=512, mapping_type="mlp",
                                                     num_layers=8, model_version="gpt2-large")
    assert cfg.train.lr == 0.0003 and cfg.seed == 7 and cfg.train.batch_size == 64
    assert cfg.train.additional.gradient_accumulation_steps == 2          # override won over base_env (mergePatch)
    assert cfg.valid.step_size == 100                                      # inherited from base_env
    assert cfg.data_loader.additional.max_target_length == 10
    from eavqa_amd.trainers import clipcap_executor
    assert hasattr(clipcap_executor, cfg.model_config.ModelClass) and cfg.train.type == "ClipCapExecutor"
    few = cs.load_config(os.path.join(ROOT, "configs", "vqa2", "few_shot_opt_2p7b.jsonnet"), mode="test")
    assert few.data_loader.additional.num_shots == 4 and few.mode == "test"


def test_t0_configs_select_the_vct0_classes_by_name():
    """The T5 / T0 path's configs (the reference's headline model, configs/vqa2/few_shot_vqa_hotpotqa.jsonnet and
    configs/conceptual_captions/conceptual_captions.jsonnet): ModelClass and executor are found by name in the executor module."""
    from eavqa_amd.trainers import vct0_executor
    few = cs.load_config(os.path.join(ROOT, "configs", "vqa2", "few_shot_vqa_t0_3b.jsonnet"), mode="test")
    cc = cs.load_config(os.path.join(ROOT, "configs", "conceptual_captions", "vct0_t0_3b.jsonnet"))
    for cfg in (few, cc):
        assert cfg.model_config.ModelClass == "VCT0Prefix" and hasattr(vct0_executor, cfg.model_config.ModelClass) and hasattr(vct0_executor, cfg.train.type)
        assert dict(cfg.model_config.model_args)["model_version"] == "bigscience/T0_3B"
    assert few.data_loader.additional.num_shots == 2 and cc.train.additional.gradient_accumulation_steps == 4


def test_merge_patch_semantics():
    assert cs.merge_patch({"a": {"b": 1, "c": 2}, "d": 3}, {"a": {"b": None, "e": 5}, "d": [1]}) == {"a": {"c": 2, "e": 5}, "d": [1]}
    assert cs.evaluate_snippet("local x = 2; local y = {a: x, 'b': [1, x,], }; std.mergePatch(y, {b: null, c: 'q' + \"r\"})") == {"a": 2, "c": "qr"}
    with pytest.raises(cs.JsonnetError):
        cs.evaluate_snippet("{a: std.length([1])}")
    with pytest.raises(cs.JsonnetError):
        cs.evaluate_snippet("{a: nope}")


def test_opts_are_literals_not_eval():
    cfg = cs.AttrDict({"a": {"b": {"c": 1}}})
    cs.parse_optional_args(cfg, ["a.b.c=[1,2]", "a.b.d=hello", "a.x=__import__('os').getcwd()"])
    assert cfg.a.b.c == [1, 2] and cfg.a.b.d == "hello" and cfg.a.x == "__import__('os').getcwd()"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_every_reference_config_evaluates():
    files = sorted(glob.glob(os.path.join(REF, "**", "*.jsonnet"), recursive=True))
    assert len(files) == 8
    for f in files:
        d = cs.evaluate_file(f, lenient=True)
        assert isinstance(d, dict) and "model_config" in d and "train" in d, f
    # the vqa2 base file uses locals it never defines (an error in real jsonnet too): strict mode says so
    with pytest.raises(cs.JsonnetError, match="VinVL_features"):
        cs.evaluate_file(os.path.join(REF, "vqa2", "clip_cap.jsonnet"))
    cfg, _ = cs.get_config_from_json(os.path.join(REF, "vqa2", "clip_cap.jsonnet"), lenient=True)
    assert cfg.model_config.ModelClass == "ClipCaptionPrefix" and cfg.train.type == "ClipCapExecutor"
    assert dict(cfg.model_config.model_args) == dict(prefix_length=10, clip_length=10, prefix_size=512, mapping_type="mlp",
                                                     num_layers=8, model_version="gpt2")
    assert cfg.train.additional.gradient_accumulation_steps == 4 and cfg.seed == 2021
    cc, _ = cs.get_config_from_json(os.path.join(REF, "conceptual_captions", "conceptual_captions.jsonnet"))
    assert cc.model_config.ModelClass == "VCT0Prefix" and cc.data_loader.type == "DataLoaderConceptualCaptions"


def test_schedules():
    import torch
    from eavqa_amd.trainers.optim import ConstantScheduleWithWarmup, CosineAnnealing, LinearScheduleWithWarmup

    class Opt:
        param_groups = [dict(lr=1.0, initial_lr=1.0)]
    s = ConstantScheduleWithWarmup(Opt(), 4)
    seq = []
    for _ in range(6):
        seq.append(s.get_last_lr()[0]); s.step()
    assert seq == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0]
    ref = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR
    cos_ref = CosineAnnealingLR(ref, 10, eta_min=1e-5)
    c = CosineAnnealing(Opt(), 10)
    for _ in range(7):
        assert abs(c.get_last_lr()[0] - cos_ref.get_last_lr()[0]) < 1e-6
        ref.step(); cos_ref.step(); c.step()
    lin = LinearScheduleWithWarmup(Opt(), 2, 10)
    vals = []
    for _ in range(11):
        vals.append(round(lin.get_last_lr()[0], 4)); lin.step()
    assert vals == [0.0, 0.5, 1.0, 0.875, 0.75, 0.625, 0.5, 0.375, 0.25, 0.125, 0.0]
