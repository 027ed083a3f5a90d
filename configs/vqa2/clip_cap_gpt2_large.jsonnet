// BASELINE.json configs[1]: CLIP ViT-B/32 -> GPT-2-large, MLP mapper, prefix 10 (the benchmarked workload).
// Same structure as the reference's configs/vqa2/clip_cap.jsonnet: model_config.ModelClass / model_args select the
// model by name (src/trainers/clipcap_exector.py:52-53).
local base_env = import 'base_env.jsonnet';
local train_batch_size = 64;
local lr = 1e-4;
local gradient_accumulation_steps = 2;   // README recipe: batch 64 x accumulation 2

local override = {
  "experiment_name": "clip_cap_gpt2_large",
  "model_config": {
    "base_model": "gpt2-large",
    "ModelClass": "ClipCaptionPrefix",
    "TokenizerClass": "GPT2Tokenizer",
    "TokenizerModelVersion": "gpt2-large",
    "ConfigClass": "GPT2Config",
    "model_args": {
      prefix_length: 10,
      clip_length: 10,
      prefix_size: 512,
      mapping_type: "mlp",
      num_layers: 8,
      model_version: "gpt2-large",
    },
    "vision_encoder": "ViT-B/32",
    "SPECIAL_TOKENS": {"bos_token": "<BOS>", "additional_special_tokens": []},
  },
  "data_loader": {
    "type": "DataLoaderConceptualCaptions",
    "additional": {'max_source_length': 1024, 'max_decoder_source_length': 1024, 'max_target_length': 10},
  },
  "train": {
    "type": "ClipCapExecutor",
    "batch_size": train_batch_size,
    "lr": lr,
    "scheduler": "none",
    "additional": {"gradient_accumulation_steps": gradient_accumulation_steps, "warmup_steps": 0, "gradient_clipping": 0},
  },
};

std.mergePatch(base_env, override)
