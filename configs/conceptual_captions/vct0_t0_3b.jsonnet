// The reference's Conceptual-Captions mapper training (configs/conceptual_captions/conceptual_captions.jsonnet): VCT0Prefix over
// bigscience/T0_3B, the encoder sees the 10 prefix vectors only, the decoder is teacher-forced on the caption.
local base_env = import 'base_env.jsonnet';
local override = {
  "experiment_name": "vct0_t0_3b",
  "model_config": {
    "base_model": "T0_3B",
    "ModelClass": "VCT0Prefix",
    "model_args": {prefix_length: 10, prefix_size: 768, mapping_type: "mlp", num_layers: 8, model_version: "bigscience/T0_3B"},
  },
  "data_loader": {"type": "DataLoaderConceptualCaptions", "additional": {'max_target_length': 20}},
  "train": {"type": "VCT0Executor", "batch_size": 32, "lr": 1e-4, "additional": {"gradient_accumulation_steps": 4, "warmup_steps": 0}},
};
std.mergePatch(base_env, override)
