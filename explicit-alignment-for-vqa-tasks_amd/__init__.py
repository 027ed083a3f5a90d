"""MI355X-native hot path of rs-anderson/explicit-alignment-for-vqa-tasks (import as ``eavqa_amd``)."""
