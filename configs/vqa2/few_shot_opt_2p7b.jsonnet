// BASELINE.json configs[3]: few-shot VQA2 generation, 4 in-context shots, CLIP ViT-L/14 + OPT-2.7B.
local base_env = import 'base_env.jsonnet';
local override = {
  "experiment_name": "few_shot_opt_2p7b",
  "model_config": {
    "base_model": "facebook/opt-2.7b",
    "ModelClass": "ClipCaptionPrefix",
    "model_args": {prefix_length: 10, prefix_size: 768, mapping_type: "mlp", model_version: "facebook/opt-2.7b"},
    "vision_encoder": "ViT-L/14",
  },
  "data_loader": {
    "type": "DataLoaderVQA2",
    "additional": {'max_source_length': 1024, 'max_target_length': 10, 'num_shots': 4},
  },
  "test": {"batch_size": 32},
};
std.mergePatch(base_env, override)
