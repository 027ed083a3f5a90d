// BASELINE.json configs[1] as a Conceptual-Captions run: CLIP ViT-B/32 -> GPT-2-large bf16, MLP mapper, prefix 10
// (bench.py --workload cfg2; configs/vqa2/clip_cap_gpt2_large.jsonnet is the same model wired to the VQA2 loader).
local base_env = import 'base_env.jsonnet';
local override = {
  "experiment_name": "cc_clip_cap_gpt2_large",
  "model_config": {
    "base_model": "gpt2-large",
    "ModelClass": "ClipCaptionPrefix",
    "TokenizerClass": "GPT2Tokenizer",
    "TokenizerModelVersion": "gpt2-large",
    "model_args": {prefix_length: 10, clip_length: 10, prefix_size: 512, mapping_type: "mlp", num_layers: 8, model_version: "gpt2-large"},
    "vision_encoder": "ViT-B/32",
    "SPECIAL_TOKENS": {"bos_token": "<BOS>", "additional_special_tokens": []},
  },
  "train": {"epochs": 10, "scheduler": "none"},
};
std.mergePatch(base_env, override)
