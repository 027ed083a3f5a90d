// Internal (not part of include/eavqa.h): the persistent one-kernel decode step behind eavqa_lm_block_forward (decode_layer.hip).
#pragma once
#include <stdint.h>
#include "eavqa.h"

// EAVQA_OK: the whole step (all layers) was enqueued on `stream`; EAVQA_E_SHAPE: shape not covered, take the multi-kernel route.
int eavqa_detail_lm_decode_persistent(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act, float eps, int B,
                                      int row0, int S_max, float* x, const int32_t* key_mask, int64_t ld_mask, void* a, void* ctx, float* x1,
                                      void* f, float* part, float* part2, int ks_qkv, int ks_o, int ks_fc1, int ks_fc2, void* stream);
