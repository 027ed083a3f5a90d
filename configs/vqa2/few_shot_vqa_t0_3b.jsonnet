// The reference's headline few-shot configuration (configs/vqa2/few_shot_vqa_hotpotqa.jsonnet): VCT0Prefix over bigscience/T0_3B,
// prompts with in-context examples, sentinel tokens <extra_id_i> expanded into the prefix of image i.
local base_env = import 'base_env.jsonnet';
local override = {
  "experiment_name": "few_shot_vqa_t0_3b",
  "model_config": {
    "base_model": "T0_3B",
    "ModelClass": "VCT0Prefix",
    "model_args": {prefix_length: 10, prefix_size: 768, mapping_type: "mlp", num_layers: 8, model_version: "bigscience/T0_3B"},
    "vision_encoder": "ViT-L/14",
  },
  "data_loader": {
    "type": "DataLoaderVQA2",
    "additional": {'max_source_length': 1024, 'max_target_length': 20, 'num_shots': 2, 'no_prefix': false,
                   'pass_examples_through_encoder_one_at_a_time': false, 'ensemble_one_shots': false,
                   'num_permutations_of_in_context_examples': 0},
  },
  "train": {"type": "FewShotVQAExecutor"},
  "test": {"batch_size": 32},
};
std.mergePatch(base_env, override)
