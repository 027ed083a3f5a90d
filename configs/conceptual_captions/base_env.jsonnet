// Conceptual-Captions mapper training: the vqa2 base environment with the CC data loader
// (the reference keeps one base_env per task folder: configs/conceptual_captions/base_env.jsonnet).
local base = import '../vqa2/base_env.jsonnet';
std.mergePatch(base, {
  "data_loader": {"type": "DataLoaderConceptualCaptions", "dataset_type": "ConceptualCaptionsDataset",
                  "additional": {'max_source_length': 1024, 'max_decoder_source_length': 1024, 'max_target_length': 10}},
  "train": {"type": "ClipCapExecutor", "batch_size": 64, "lr": 1e-4,
            "additional": {"gradient_accumulation_steps": 2, "warmup_steps": 0, "gradient_clipping": 0}},
})
