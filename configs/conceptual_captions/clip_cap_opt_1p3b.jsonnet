// BASELINE.json configs[2]: CLIP ViT-L/14 -> OPT-1.3B bf16, MLP mapper, prefix 10, data parallel over the 8 GPUs of a node
// (bench.py --workload cfg3).  README recipe of the reference: batch 64 x accumulation 2, lr 1e-4, 10 epochs.
local base_env = import 'base_env.jsonnet';
local override = {
  "experiment_name": "clip_cap_opt_1p3b",
  "model_config": {
    "base_model": "facebook/opt-1.3b",
    "ModelClass": "ClipCaptionPrefix",
    "TokenizerClass": "GPT2Tokenizer",
    "TokenizerModelVersion": "facebook/opt-1.3b",
    "model_args": {prefix_length: 10, clip_length: 10, prefix_size: 768, mapping_type: "mlp", num_layers: 8,
                   model_version: "facebook/opt-1.3b"},
    "vision_encoder": "ViT-L/14",
    "SPECIAL_TOKENS": {"bos_token": "<BOS>", "additional_special_tokens": []},
  },
  "train": {"epochs": 10, "scheduler": "none"},
};
std.mergePatch(base_env, override)
