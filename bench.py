#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: mapper training samples/s on Conceptual-Captions-shaped
synthetic batches (BASELINE.json metric M1), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one per-GPU batch: CLIP ViT encode -> mapping network ->
frozen causal LM over [prefix | caption] -> shifted CE -> backward into the mapper -> gradient
all-reduce (RCCL, N > 1) -> fused AdamW.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (contract in the task description) that also carries
  roofline     - achieved TFLOP/s of the dominant kernel (the MFMA GEMM) from HIP events recorded on
                 the launch stream around every GEMM launch of an extra, instrumented step that
                 runs right after the timed region (so the timed value is not perturbed), plus the
                 HBM-bound ops of that step (bytes / duration / 8 TB/s) under roofline.hbm_kernels,
  cpu_baseline - the CPU oracle (oracle/ref_cpu.py, kind "port") timed on this host's cores on a
                 bounded sample of the same workload (N = 1, rank 0 only), and
  extra        - (default cfg2 run, N = 1) a list of the other 1-GPU BASELINE configs, each in the same shape:
                 few-shot VQA2 generate (cfg4, questions/s, per-phase roofline), cfg3 bf16 and cfg5 fp8 training
                 (5 warm-up + 10 timed steps each, own roofline), and the reference's own headline model, T0_3B:
                 few-shot generate (t0_3b_fewshot) and Conceptual-Captions mapper training (t0_3b_cc_train).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: RCCL between processes needs the non-legacy mode (already exported on the boxes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(vit="ViT-B/32", lm="gpt2-large", mapping_type="mlp", prefix_length=10, batch=64, text_len=32,
                 desc="CLIP ViT-B/32 -> GPT-2-large (774M), MLP mapper, prefix 10, CC-shaped synthetic batches"),
    "cfg1": dict(vit="ViT-B/32", lm="gpt2", mapping_type="mlp", prefix_length=10, batch=4, text_len=32,
                 desc="CLIP ViT-B/32 -> GPT-2-small, MLP mapper, batch 4 (the reference's CPU-runnable case)"),
    "cfg3": dict(vit="ViT-L/14", lm="facebook/opt-1.3b", mapping_type="mlp", prefix_length=10, batch=64, text_len=32,
                 desc="CLIP ViT-L/14 -> OPT-1.3B, MLP mapper, prefix 10"),
    # BASELINE.json configs[4]: fp8 frozen LM, 32-token prefix.  Mapper: the reference's TransformerMapper (1.17 B parameters at
    # E = 4096, src/models/clipcap.py:265-271) - its MLP formula (:256-262) gives 768 -> 65 536 -> 131 072 = 8.64 B parameters, larger
    # than the LM; `--mapping-type mlp` runs that variant (155 GB of mapper state on one GPU).
    "cfg5": dict(vit="ViT-L/14", lm="facebook/opt-6.7b", mapping_type="transformer", prefix_length=32, clip_length=32, batch=32, text_len=32,
                 desc="CLIP ViT-L/14 -> OPT-6.7B (fp8 e4m3 weights on the block-scaled MFMA with --dtype fp8), transformer mapper (8 layers), prefix 32"),
}

# algorithmic FLOPs per sample: ViT fwd + 3 x mapper + 2 x LM fwd, 2*params*tokens for
# GEMMs + 4*S^2*E per layer for attention
def flops_per_sample(vit_cfg, lm_cfg, L, S, D, mapping_type="mlp", clip_length=10, num_layers=8):
    W, N = vit_cfg.width, vit_cfg.n_patch + 1
    vit = vit_cfg.n_layer * (2 * N * (4 * W * W + 2 * W * vit_cfg.mlp) + 4 * N * N * W) + 2 * vit_cfg.n_patch * 3 * vit_cfg.patch ** 2 * W + 2 * W * vit_cfg.proj
    E, F, V = lm_cfg.n_embd, lm_cfg.ffn, lm_cfg.vocab
    lm = lm_cfg.n_layer * (2 * S * (4 * E * E + 2 * E * F) + 4 * S * S * E) + 2 * S * E * V
    if mapping_type == "transformer":
        seq = clip_length + L
        mapper = 2 * D * clip_length * E + num_layers * (2 * seq * 8 * E * E + 4 * seq * seq * E)
    else:
        H = E * L // 2
        mapper = 2 * (D * H + H * E * L)
    return vit + 3 * mapper + 2 * lm


def build_workload(name, dtype, device, rank, mapping_type=None, weight_format="native"):
    from eavqa_amd.data.synthetic import cc_batch
    from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import KNOWN_CONFIGS, FrozenCausalLM, LMConfig, random_init_state_dict
    from eavqa_amd.trainers.optim import FusedAdamW

    w = dict(WORKLOADS[name])
    if mapping_type:
        w["mapping_type"] = mapping_type
    vcfg = KNOWN_VITS[w["vit"]]
    lcfg = LMConfig.from_hf_dict(KNOWN_CONFIGS[w["lm"]])
    vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, device), dtype, device)
    lm = FrozenCausalLM(lcfg, random_init_state_dict(lcfg, 2021, device), dtype, device, weight_format=weight_format)
    torch.manual_seed(2021)   # mapper init = nn.Linear default under the reference seed
    model = ClipCaptionPrefix(prefix_length=w["prefix_length"], clip_length=w.get("clip_length", w["prefix_length"]), prefix_size=vcfg.proj,
                              mapping_type=w["mapping_type"], lm=lm, dtype=dtype, device=device).train()
    opt = FusedAdamW(model.clip_project.flat, lr=1e-4)
    pad = lcfg.eos_token_id
    batch = cc_batch(w["batch"], lcfg.vocab, pad, image_size=vcfg.image, max_len=w["text_len"], seed=2021 + rank, device="cpu")
    lengths = batch["attention_mask"].sum(1).tolist()      # host-side metadata of the collate (caption lengths,
    label_count = int((batch["labels"] != -100).sum())     # number of scored positions)
    batch = {k: v.to(device) for k, v in batch.items()}
    batch["question_lengths"] = lengths
    batch["label_count"] = label_count
    return w, vcfg, lcfg, vit, model, opt, batch, pad


class Stepper:
    """One training step, software-pipelined across steps:
      * the CLIP encode of batch i+1 runs on a side stream while the LM forward/backward of batch i runs on the main
        stream (the encode depends on nothing trainable; the LM's N = 1280 GEMMs leave ~40 % of the CUs idle);
      * with N > 1 the gradient all-reduce of step i overlaps the same window and AdamW(i) is applied right before the
        mapper forward of step i+1 (the mapper gradient is the last thing backward produces).
    Every step performs one forward/backward and one optimiser update, and one encode - except the very first step of a run,
    which also encodes its own batch to fill the pipeline (untimed: it falls into the warm-up)."""

    def __init__(self, vit, model, opt, batch, pad, sync, overlap_vit=True, pipelined_update=False):
        self.vit, self.model, self.opt, self.batch, self.pad, self.sync = vit, model, opt, batch, pad, sync
        self.pending_update = False
        # AdamW chunk by chunk on its own stream, the mapper's forward waiting layer by layer (FusedAdamW.step(chunks=...)): the update is
        # HBM-bound, the mapper's GEMMs are not - the first layers run while the last ones are still being updated
        mapper = getattr(model, "clip_project", None)
        self.chunks = mapper.update_chunks() if (pipelined_update and hasattr(mapper, "update_chunks")) else None
        self.side = torch.cuda.Stream() if overlap_vit else None
        self.next_emb = None
        self.next_ready = None

    def _apply_update(self):
        if self.pending_update:
            self.sync.finish()
            if self.sync is not self.opt:            # the sharded optimiser has already updated its shard behind the reduce-scatter
                self.opt.step(grad_scale=self.sync.grad_scale, chunks=self.chunks)
            self.opt.zero_grad()
            self.pending_update = False

    def _encode_async(self):
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)                              # starts behind what is queued on main (backward of step i-1)
        with torch.cuda.stream(self.side):
            emb = self.vit.encode_image(self.batch["pixel_values"])
            ev = torch.cuda.Event()
            ev.record(self.side)
        emb.record_stream(main)                                  # consumed on the main stream
        self.next_emb, self.next_ready = emb, ev

    def step(self):
        b = self.batch
        if self.side is None:
            emb = self.vit.encode_image(b["pixel_values"])
        else:
            if self.next_emb is None:
                self._encode_async()                             # pipeline fill (first step only)
            emb, ready = self.next_emb, self.next_ready
            torch.cuda.current_stream().wait_event(ready)
        if self.side is not None:
            # batch i+1's encode starts behind the backward of batch i-1: with N > 1 it fills the otherwise exposed wait
            # for the gradient all-reduce, then runs on beside AdamW and the LM FORWARD of batch i (whose N = 3840 / 5120
            # GEMMs leave CUs idle; beside the backward, whose 128 x 80 tiles fill every CU, the two streams only slow
            # each other down: measured)
            self._encode_async()
        self._apply_update()                                     # all-reduce wait + AdamW of the previous step
        out = self.model(question_tokens=b["input_ids"], prefix=emb, question_mask=b["attention_mask"], labels=b["labels"],
                         pad_token_id=self.pad, question_lengths=b["question_lengths"], label_count=b["label_count"])
        arm = getattr(self.sync, "arm", None)
        if arm is not None:
            arm()                                    # sharded exchange: reduce-scatter per bucket as the mapper's backward completes it
        out.loss.backward()
        self.sync.start()
        self.pending_update = True
        return out.loss

    def flush(self):
        self._apply_update()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def exchange_ms(self):
        """Device time of the last step's gradient exchange (HIP events on the exchange's own stream), None when there is none."""
        fn = getattr(self.sync, "last_exchange_ms", None)
        return fn() if fn is not None else None


FEWSHOT = dict(vit="ViT-L/14", lm="facebook/opt-2.7b", prefix_length=10, batch=32, shots=4, seg_len=20, new_tokens=10,
               desc="few-shot VQA2 generate (BASELINE configs[3]): CLIP ViT-L/14 + OPT-2.7B, 4 in-context shots + query "
                    "(5 images/question), 20 text tokens per segment, prompt 150 positions after prefix insertion, 10 new tokens")


def fewshot_qps(dtype, device, reps=6, encode_ahead=True):
    """Questions/s of the few-shot generate path (metric M2): ViT encode of 5 images per question, MLP mapper,
    sentinel expansion (insert_prefix_into_input), prefill, 10 greedy steps with a KV cache."""
    from eavqa_amd.data.synthetic import fewshot_batch
    from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import KNOWN_CONFIGS, FrozenCausalLM, LMConfig, random_init_state_dict
    f = FEWSHOT
    vcfg = KNOWN_VITS[f["vit"]]
    lcfg = LMConfig.from_hf_dict(KNOWN_CONFIGS[f["lm"]])
    vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, device), dtype, device)
    lm = FrozenCausalLM(lcfg, random_init_state_dict(lcfg, 2021, device), dtype, device)
    torch.manual_seed(2021)
    model = ClipCaptionPrefix(prefix_length=f["prefix_length"], prefix_size=vcfg.proj, mapping_type="mlp", lm=lm, dtype=dtype,
                              device=device).eval()
    sentinel = lcfg.vocab - 1
    b = fewshot_batch(f["batch"], lcfg.vocab, f["shots"], f["seg_len"], sentinel, image_size=vcfg.image, device=device)
    B, n_img = f["batch"], f["shots"] + 1

    def run(marks=None):
        px = b["pixel_values"]
        if marks is not None:
            marks.append(("start", _event()))
        emb = vit.encode_image(px.reshape(B * n_img, *px.shape[2:])).view(B, n_img, -1)
        if marks is not None:
            marks.append(("encode", _event()))
        return model.generate_fewshot(b["input_ids"], emb, b["attention_mask"], num_shots=f["shots"], special_token_id=sentinel,
                                      max_length=f["new_tokens"], pad_token_id=lcfg.pad_token_id, eos_token_id=None, marks=marks)

    def generate(emb):
        return model.generate_fewshot(b["input_ids"], emb.view(B, n_img, -1), b["attention_mask"], num_shots=f["shots"], special_token_id=sentinel,
                                      max_length=f["new_tokens"], pad_token_id=lcfg.pad_token_id, eos_token_id=None)

    run()
    torch.cuda.synchronize()
    out, dt = _timed_batches(vit, b["pixel_values"].reshape(B * n_img, *b["pixel_values"].shape[2:]), generate, reps, encode_ahead)
    assert len(out) == B and len(out[0]) == f["new_tokens"]
    # per-phase roofline of one more, instrumented batch (HIP events on the launch stream; a ~40 ms head start of queued work
    # keeps host latency out of the brackets, as in gemm_roofline)
    blk = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    blk_c = torch.empty(8192, 8192, device=device, dtype=torch.bfloat16)
    for _ in range(40):
        ops_gemm(blk, blk, out=blk_c)
    marks = []
    run(marks)
    torch.cuda.synchronize()
    ev = dict(marks)
    ms = dict(encode=ev["start"].elapsed_time(ev["encode"]), prefill=ev["encode"].elapsed_time(ev["prefill"]),
              decode=ev["prefill"].elapsed_time(ev["decode"]))
    L, S0 = f["prefix_length"], n_img * (1 + f["seg_len"]) + (f["prefix_length"] - 1) * n_img
    E, F, V, NL = lcfg.n_embd, lcfg.ffn, lcfg.vocab, lcfg.n_layer
    W, N = vcfg.width, vcfg.n_patch + 1
    vit_flop = B * n_img * (vcfg.n_layer * (2 * N * (4 * W * W + 2 * W * vcfg.mlp) + 4 * N * N * W) + 2 * vcfg.n_patch * 3 * vcfg.patch ** 2 * W + 2 * W * vcfg.proj)
    H_map = E * L // 2
    prefill_flop = B * (NL * (2 * S0 * (4 * E * E + 2 * E * F) + 4 * S0 * S0 * E) + 2 * E * V) + B * n_img * 2 * (vcfg.proj * H_map + H_map * E * L)
    steps = f["new_tokens"] - 1                                  # the first token comes out of the prefill
    weight_bytes = 2.0 * (NL * (4 * E * E + 2 * E * F) + E * V)
    kv_bytes = sum(2.0 * NL * B * (S0 + t + 1) * E * 2 for t in range(steps)) / max(steps, 1)      # K and V rows read per step, bf16
    roof = [
        dict(phase="vit_encode", bound="mfma", kernel="eavqa_gemm (ViT-L/14 tower) + eavqa_attn_mfma::fwd", ms=round(ms["encode"], 3),
             achieved=round(vit_flop / (ms["encode"] * 1e-3) / 1e12, 1), peak=2500.0, unit="TFLOP/s"),
        dict(phase="mapper_prefill", bound="mfma", kernel="eavqa_lm_block_forward (prefill: eavqa_gemm + attention over 150 positions)",
             ms=round(ms["prefill"], 3), achieved=round(prefill_flop / (ms["prefill"] * 1e-3) / 1e12, 1), peak=2500.0, unit="TFLOP/s"),
        dict(phase="decode", bound="hbm", kernel="eavqa_lm_block_forward (decode: gemm_bf16_splitk + attn_decode), weights + KV cache read once per step",
             ms=round(ms["decode"], 3), ms_per_step=round(ms["decode"] / max(steps, 1), 3), steps=steps,
             achieved=round((weight_bytes + kv_bytes) * steps / (ms["decode"] * 1e-3) / 1e9, 1), peak=8000.0, unit="GB/s",
             bytes_per_step=round(weight_bytes + kv_bytes)),
    ]
    for r in roof:
        r["frac"] = round(r["achieved"] / r["peak"], 4)
    del model, lm, vit
    torch.cuda.empty_cache()
    return {"metric": "fewshot_vqa_questions_per_sec", "value": round(B / dt, 2), "unit": "questions/s", "ms_per_batch": round(dt * 1e3, 2),
            "config": {"workload": "few-shot cfg4: " + f["desc"], "batch": B, "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "kv_cache": True,
                       "batches_timed": reps, "encode_ahead": bool(encode_ahead)},
            "roofline": roof}


T0 = dict(vit="ViT-L/14", lm="bigscience/T0_3B", prefix_length=10, fewshot_batch=32, shots=4, seg_len=20, new_tokens=10, train_batch=64, text_len=32,
          desc="the reference's headline model (SURVEY F2): CLIP ViT-L/14 -> MLP mapper -> T0_3B (T5 v1.1 XL encoder-decoder), VCT0Prefix, prefix 10")


def _timed_batches(vit, px, generate, reps, encode_ahead):
    """``reps`` batches of (image tower -> generate) back to back, wall clock per batch.  With ``encode_ahead`` the tower of batch i + 1 is
    issued on a second stream before batch i is generated (models/clip_vit.py::EncodeAhead; the first batch's tower is inside the timed
    region and overlaps nothing), otherwise the plain sequential loop."""
    from eavqa_amd.models.clip_vit import EncodeAhead
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if encode_ahead:
        ahead = EncodeAhead(vit)
        ticket = ahead.submit(px)
        for i in range(reps):
            emb = ahead.result(ticket)
            if i + 1 < reps:
                ticket = ahead.submit(px)
            out = generate(emb)
    else:
        for _ in range(reps):
            out = generate(vit.encode_image(px))
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps


def _t0_models(dtype, device, train):
    from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
    from eavqa_amd.models.t5 import KNOWN_T5, FrozenT5, T5Config, random_init_t5_state_dict
    from eavqa_amd.models.vct0 import VCT0Prefix
    vcfg = KNOWN_VITS[T0["vit"]]
    vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, device), dtype, device)
    tcfg = T5Config.from_hf_dict(KNOWN_T5[T0["lm"]])
    lm = FrozenT5(tcfg, random_init_t5_state_dict(tcfg, 2021, device), dtype, device)      # synthetic weights, as every leg of this bench
    torch.manual_seed(2021)
    model = VCT0Prefix(prefix_length=T0["prefix_length"], prefix_size=vcfg.proj, mapping_type="mlp", lm=lm, dtype=dtype, device=device)
    return vcfg, tcfg, vit, (model.train() if train else model.eval())


def _t5_stack_flops(c, enc_pos, dec_pos):
    """Forward FLOPs per sample of the T5 stacks: 2 x parameters x positions for the projections + 4 S^2 inner per attention."""
    E, I, F = c.d_model, c.inner, c.d_ff
    ffn = (3 if c.gated else 2) * E * F
    enc = c.n_layer * (2 * enc_pos * (4 * E * I + ffn) + 4 * enc_pos * enc_pos * I)
    dec = c.n_dec_layer * (2 * dec_pos * (6 * E * I + ffn) + 2 * enc_pos * 2 * E * I + 4 * dec_pos * dec_pos * I + 4 * dec_pos * enc_pos * I)
    return enc, dec


def t0_fewshot_qps(dtype, device, reps=6, encode_ahead=True):
    """Few-shot VQA2 generate with the reference's own model (few_shot_vqa_executor.py:195-205 -> VCT0Model.generate, vct0.py:396-491):
    32 questions x (4 shots + query), ViT-L/14 encode of 5 images per question, MLP mapper, sentinel expansion, T5 encoder over the 150
    interleaved positions, cross K / V of 24 decoder layers once, 9 cached greedy decoder steps (max_length 10)."""
    from eavqa_amd.data.synthetic import fewshot_batch
    from eavqa_amd.models.t5 import _StepDriver
    vcfg, c, vit, model = _t0_models(dtype, device, train=False)
    lm = model.lm
    B, shots, new = T0["fewshot_batch"], T0["shots"], T0["new_tokens"]
    n_img = shots + 1
    b = fewshot_batch(B, c.vocab, shots, T0["seg_len"], 32099, image_size=vcfg.image, device=device)
    px = b["pixel_values"].reshape(B * n_img, *b["pixel_values"].shape[2:])

    def run():
        emb = vit.encode_image(px).view(B, n_img, -1)
        return model.generate(prefix=emb, question_tokens=b["input_ids"], question_mask=b["attention_mask"], num_shots=shots, max_length=new)

    out = run()
    torch.cuda.synchronize()
    out, dt = _timed_batches(vit, px, lambda emb: model.generate(prefix=emb.view(B, n_img, -1), question_tokens=b["input_ids"],
                                                                  question_mask=b["attention_mask"], num_shots=shots, max_length=new),
                             reps, encode_ahead)
    assert tuple(out.shape) == (B, new)
    # phases of one more batch, each bracketed by HIP events on the launch stream behind a head start of queued work
    blk = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    blk_c = torch.empty(8192, 8192, device=device, dtype=torch.bfloat16)
    for _ in range(40):
        ops_gemm(blk, blk, out=blk_c)
    e = [_event()]
    emb = vit.encode_image(px).view(B, n_img, -1)
    e.append(_event())
    rows = model._project(emb)
    enc, mask, S = model._encode_interleaved(b["input_ids"], b["attention_mask"], rows, n_img, 32099)
    kv = lm.cross_kv(enc)
    e.append(_event())
    t_max = new
    cache = [(torch.empty((B * t_max, c.inner), device=device, dtype=dtype), torch.empty((B * t_max, c.inner), device=device, dtype=dtype)) for _ in lm.dec]
    driver = _StepDriver(lm, cache, kv, B, t_max)
    rel = lm.rel_table(True, t_max)
    tok = torch.zeros(B, dtype=torch.int64, device=device)
    for t in range(1, t_max):                         # (the cache buffers and the table are warm; these steps are not timed)
        driver.step(lm.embed(tok), mask, t, S, rel)
    e.append(_event())
    for t in range(1, t_max):                         # a greedy step without the token bookkeeping: embed, 24 decoder layers, lm_head
        lg = lm.logits(driver.step(lm.embed(tok), mask, t, S, rel))
    e.append(_event())
    torch.cuda.synchronize()
    ms = dict(encode=e[0].elapsed_time(e[1]), t5_encoder=e[1].elapsed_time(e[2]), decode=e[3].elapsed_time(e[4]))
    W, N = vcfg.width, vcfg.n_patch + 1
    vit_flop = B * n_img * (vcfg.n_layer * (2 * N * (4 * W * W + 2 * W * vcfg.mlp) + 4 * N * N * W) + 2 * vcfg.n_patch * 3 * vcfg.patch ** 2 * W + 2 * W * vcfg.proj)
    E, I, F = c.d_model, c.inner, c.d_ff
    H_map = E * T0["prefix_length"] // 2
    enc_flop = B * (_t5_stack_flops(c, S, 0)[0] + c.n_dec_layer * 2 * S * 2 * E * I) + B * n_img * 2 * (vcfg.proj * H_map + H_map * E * T0["prefix_length"])
    steps = new - 1
    weight_bytes = 2.0 * (c.n_dec_layer * (6 * E * I + (3 if c.gated else 2) * E * F) + E * c.vocab)
    kv_bytes = 2.0 * c.n_dec_layer * B * S * 2 * I                       # cross K / V of every layer, read once per step (bf16)
    roof = [
        dict(phase="vit_encode", bound="mfma", kernel="eavqa_gemm (ViT-L/14 tower) + eavqa_attn_mfma::fwd_resident64", ms=round(ms["encode"], 3),
             achieved=round(vit_flop / (ms["encode"] * 1e-3) / 1e12, 1), peak=2500.0, unit="TFLOP/s"),
        dict(phase="mapper_t5_encoder_cross_kv", bound="mfma", kernel="FrozenT5.encode over 150 positions + cross K / V of 24 layers (eavqa_gemm, eavqa_attn_mfma::fwd with relative bias)",
             ms=round(ms["t5_encoder"], 3), achieved=round(enc_flop / (ms["t5_encoder"] * 1e-3) / 1e12, 1), peak=2500.0, unit="TFLOP/s"),
        dict(phase="decode", bound="hbm", kernel="eavqa_t5_decoder_step (gemm_bf16_splitk + attn_decode + rms / gated consumers) + lm_head, weights + cross K / V read once per step",
             ms=round(ms["decode"], 3), ms_per_step=round(ms["decode"] / steps, 3), steps=steps,
             achieved=round((weight_bytes + kv_bytes) * steps / (ms["decode"] * 1e-3) / 1e9, 1), peak=8000.0, unit="GB/s",
             bytes_per_step=round(weight_bytes + kv_bytes)),
    ]
    for r in roof:
        r["frac"] = round(r["achieved"] / r["peak"], 4)
    del model, lm, vit, driver, cache, kv
    torch.cuda.empty_cache()
    return {"metric": "fewshot_vqa_questions_per_sec", "value": round(B / dt, 2), "unit": "questions/s", "ms_per_batch": round(dt * 1e3, 2),
            "config": {"workload": "t0_3b_fewshot: " + T0["desc"] + "; 4 in-context shots + query (5 images / question), 20 text tokens per segment, "
                                   "150 encoder positions, max_length 10", "batch": B, "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "kv_cache": True,
                       "batches_timed": reps, "encode_ahead": bool(encode_ahead)},
            "roofline": roof}


class _T0TrainModel:
    """Adapter: ``Stepper`` calls the causal-LM model's keyword signature; ``VCT0Model.forward`` takes (prefix, labels) (vct0.py:380-394)."""

    def __init__(self, model):
        self.model, self.clip_project = model, model.clip_project

    def __call__(self, *, prefix, labels, **_):
        return self.model(prefix=prefix, labels=labels)


def t0_cc_train_leg(dtype_name, steps, warmup, device, args):
    """Mapper training on CC-shaped batches through the frozen T0_3B (vct0_exector.py:130-146; the recipe of the reference's
    README.md:199-209: batch 64, prefix 10, MLP mapper): ViT-L/14 encode -> mapper -> T5 encoder over the 10 prefix positions ->
    teacher-forced decoder over the caption -> CE -> dgrad through decoder, cross-attention and encoder -> mapper wgrad -> fused AdamW."""
    from eavqa_amd import ops
    from eavqa_amd.data.synthetic import cc_batch
    from eavqa_amd.trainers.data_parallel import GradSync
    from eavqa_amd.trainers.optim import FusedAdamW
    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    vcfg, c, vit, vct0 = _t0_models(dtype, device, train=True)
    model = _T0TrainModel(vct0)
    opt = FusedAdamW(vct0.clip_project.flat, lr=1e-4)
    B = T0["train_batch"]
    batch = cc_batch(B, c.vocab, c.pad_token_id, image_size=vcfg.image, max_len=T0["text_len"], seed=2021, device="cpu")
    batch = {k: v.to(device) for k, v in batch.items()}
    batch["question_lengths"], batch["label_count"] = None, None
    sync = GradSync(vct0.clip_project.flat.grad, 1, exchange=True)
    stepper = Stepper(vit, model, opt, batch, c.pad_token_id, sync, overlap_vit=not args.no_overlap, pipelined_update=args.pipelined_update)
    for _ in range(warmup):
        stepper.step()
    stepper.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = stepper.step()
    stepper.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_per_step = 1e3 * dt / steps
    roof = step_roofline(vit, model, opt, batch, c.pad_token_id, sync, ops, "t0_3b_cc_train", dtype_name, ms_per_step) if not args.no_roofline else None
    L, T = T0["prefix_length"], batch["labels"].shape[1]
    W, N = vcfg.width, vcfg.n_patch + 1
    vit_f = vcfg.n_layer * (2 * N * (4 * W * W + 2 * W * vcfg.mlp) + 4 * N * N * W) + 2 * vcfg.n_patch * 3 * vcfg.patch ** 2 * W + 2 * W * vcfg.proj
    E = c.d_model
    H_map = E * L // 2
    enc_f, dec_f = _t5_stack_flops(c, L, T)
    fps = vit_f + 3 * 2 * (vcfg.proj * H_map + H_map * E * L) + 2 * (enc_f + dec_f + 2 * T * E * c.vocab)
    line = {"metric": "mapper_train_samples_per_sec", "value": round(B * steps / dt, 2), "unit": "samples/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": "t0_3b_cc_train: " + T0["desc"] + f"; per-GPU batch {B}, encoder positions {L}, decoder positions {T}; fwd+bwd+AdamW; "
                                   "random-init weights", "global_batch": B, "seq_len": L + T, "parallelism": "dp1",
                       "algorithmic_gflop_per_sample": round(fps / 1e9, 1), "step_tflops": round(B * steps / dt * fps / 1e12, 1),
                       "final_loss": round(float(loss.item()), 4)},
            "roofline": roof}
    del stepper, vit, model, vct0, opt, sync, batch
    torch.cuda.empty_cache()
    return line


def _event():
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def ops_gemm(a, b, out):
    from eavqa_amd import ops
    return ops.gemm(a, b, out=out)


def cpu_baseline(name, n_samples, threads):
    """The oracle (CPU restatement, fp32) on a bounded sample of the same workload: one training step
    (ViT encode, mapper, LM forward, backward into the mapper, AdamW) over ``n_samples`` samples."""
    import oracle
    from eavqa_amd.data.synthetic import cc_batch
    from eavqa_amd.models.clip_vit import KNOWN_VITS, random_init_vit_state_dict
    from eavqa_amd.models.lm import KNOWN_CONFIGS, LMConfig, random_init_state_dict

    torch.set_num_threads(threads)
    w = WORKLOADS[name]
    vcfg, lcfg = KNOWN_VITS[w["vit"]], LMConfig.from_hf_dict(KNOWN_CONFIGS[w["lm"]])
    vsd = random_init_vit_state_dict(vcfg, 2021, "cpu")
    lsd = random_init_state_dict(lcfg, 2021, "cpu")
    L, E, D = w["prefix_length"], lcfg.n_embd, vcfg.proj
    torch.manual_seed(2021)
    l0, l2 = torch.nn.Linear(D, E * L // 2), torch.nn.Linear(E * L // 2, E * L)
    mapper = {"model.0.weight": l0.weight.detach().requires_grad_(True), "model.0.bias": l0.bias.detach().requires_grad_(True),
              "model.2.weight": l2.weight.detach().requires_grad_(True), "model.2.bias": l2.bias.detach().requires_grad_(True)}
    b = cc_batch(n_samples, lcfg.vocab, lcfg.eos_token_id, image_size=vcfg.image, max_len=w["text_len"], seed=2021, device="cpu")
    ocfg = dict(arch=lcfg.arch, n_layer=lcfg.n_layer, n_head=lcfg.n_head, act=lcfg.act)
    vc = dict(width=vcfg.width, n_layer=vcfg.n_layer, n_head=vcfg.n_head, patch=vcfg.patch)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in mapper.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        emb = oracle.clip_vit_encode(vsd, vc, b["pixel_values"])
    loss, _ = oracle.clipcap_forward(lsd, ocfg, mapper, dict(prefix_length=L, mapping_type="mlp"), b["input_ids"], emb,
                                     b["attention_mask"], b["labels"])
    loss.backward()
    with torch.no_grad():
        for k, p in mapper.items():
            oracle.adamw_step(p, p.grad, state[k][0], state[k][1], 1, 1e-4)
    dt = time.perf_counter() - t0
    return dict(value=round(n_samples / dt, 3), unit="samples/s", cores=threads, kind="port",
                sample=f"1 training step of {n_samples} samples of the same workload (oracle/ref_cpu.py, fp32, {dt:.1f} s)")


def log(msg):
    """Progress on stderr (stdout carries only the JSON line)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """CPU threads this process may really use: the affinity mask, capped at the GPU box's 16-core share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("EAVQA_CPU_THREADS", "16"))))


def train_leg(workload, dtype_name, steps, warmup, rank, world, device, args, log_prefix=""):
    """One training workload: build, ``warmup`` untimed steps, exactly ``steps`` timed steps between barriers (max over ranks),
    then - behind the timed region - the instrumented roofline step.  Returns the JSON object of the leg (no cpu_baseline / extra)."""
    from eavqa_amd import ops
    from eavqa_amd.trainers.data_parallel import GradSync
    from eavqa_amd.trainers.optim import ShardedAdamW, choose_dp_exchange, dp_exchange_costs

    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    weight_format = "fp8" if dtype_name == "fp8" else "native"
    log(f"{log_prefix}building workload {workload} ({dtype_name}) on {device}")
    w, vcfg, lcfg, vit, model, opt, batch, pad = build_workload(workload, dtype, device, rank, args.mapping_type, weight_format)
    torch.cuda.synchronize()
    log(f"{log_prefix}workload built")
    # N > 1: which exchange carries the mapper gradient
    flat = model.clip_project.flat
    has_factors = hasattr(model.clip_project, "dp_factor_exchange")
    exchange = args.dp_exchange
    if world == 1:
        exchange = "none"
    elif exchange == "auto":
        exchange = choose_dp_exchange(flat.numel, flat.numel if has_factors else 0, w["batch"], world)
    if exchange == "factors" and not has_factors:
        raise SystemExit("--dp-exchange factors needs the MLP mapper")
    factors = exchange == "factors"
    if factors:
        model.clip_project.dp_factor_exchange = True
    if exchange == "sharded":
        # exchange + update in one object (arm / start / finish)
        opt = ShardedAdamW(flat, lr=1e-4, grad_transport=torch.bfloat16 if args.grad_transport == "bf16" else None)
        sync = opt
    else:
        sync = GradSync(flat.grad, world, exchange=not factors)
    stepper = Stepper(vit, model, opt, batch, pad, sync, overlap_vit=not args.no_overlap, pipelined_update=args.pipelined_update)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        stepper.step()
        if i == 0:
            torch.cuda.synchronize()
            log(f"{log_prefix}first step done")
    stepper.flush()
    barrier()
    log(f"{log_prefix}warmup done")
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = stepper.step()
    stepper.flush()
    barrier()
    dt = time.perf_counter() - t0
    log(f"{log_prefix}timed region: {steps} steps in {dt:.3f} s")
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    B, S = w["batch"], w["prefix_length"] + batch["input_ids"].shape[1]
    value = world * B * steps / dt
    ms_per_step = 1e3 * dt / steps

    roof = None
    if not args.no_roofline:
        # every rank runs it (the step contains the exchange); rank 0 reports
        roof = step_roofline(vit, model, opt, batch, pad, sync, ops, workload, dtype_name, ms_per_step, getattr(args, "hbm_bytes_out", None))
        if rank == 0:
            log(f"{log_prefix}roofline pass done: {roof}")
    fps = flops_per_sample(vcfg, lcfg, w["prefix_length"], S, vcfg.proj, w["mapping_type"], w.get("clip_length", w["prefix_length"]))
    dist_cfg = {}
    if world > 1:
        import torch.distributed as dist
        dist_cfg = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                    "dp_exchange": {"factors": "mapper gradient factors (all-gather; whole-batch weight gradient on every rank)",
                                    "sharded": "reduce-scatter + sharded AdamW + all-gather of the bf16 operand copy",
                                    "allreduce": "flat gradient all-reduce"}[exchange],
                    # MODELLED, never measured on a multi-GPU box: what the chooser compared
                    "dp_exchange_model_ms_unmeasured": {k: round(v * 1e3, 2) for k, v in dp_exchange_costs(
                        flat.numel, flat.numel if has_factors else 0, w["batch"], world).items()},
                    # the same model if RCCL's reduce-scatter / all-gather spread over all seven xGMI links of a GPU: the first multi-GPU run's
                    # dp_exchange_ms_per_step below says which of the two the hardware is closer to
                    "dp_exchange_model_ms_7_links_unmeasured": {k: round(v * 1e3, 2) for k, v in dp_exchange_costs(
                        flat.numel, flat.numel if has_factors else 0, w["batch"], world, links=7).items()},
                    "grad_transport": args.grad_transport,
                    "dp_exchange_ms_per_step": stepper.exchange_ms()}
    line = {
        "metric": "mapper_train_samples_per_sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"{workload}: {w['desc']}" + (f" [--mapping-type {args.mapping_type} overrides the mapper]" if args.mapping_type else "")
                               + f"; per-GPU batch {B}, S={S}; fwd+bwd+AdamW; random-init weights",
                   "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}", **dist_cfg,
                   "algorithmic_gflop_per_sample": round(fps / 1e9, 1),
                   "step_tflops": round(value * fps / 1e12, 1), "final_loss": round(float(loss.item()), 4)},
        "roofline": roof,
    }
    del stepper, vit, model, opt, sync, batch          # free the workload's HBM before the next leg
    torch.cuda.empty_cache()
    return line


# secondary legs carried under "extra" by the default run (N = 1): the other 1-GPU BASELINE configs, short
EXTRA_TRAIN_LEGS = (("cfg3", "bf16", 10, 5), ("cfg5", "fp8", 10, 5))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: bf16 activations, the frozen LM's Linear weights in e4m3 on the block-scaled MFMA (cfg5)")
    ap.add_argument("--cpu-baseline-samples", type=int, default=64, help="0 disables the CPU baseline leg (default: one whole step of the workload, ~10 s)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dp-exchange", choices=["auto", "factors", "allreduce", "sharded"], default="auto",
                    help="N > 1: 'factors' all-gathers the MLP mapper's gradient factors, 'sharded' = reduce-scatter + sharded AdamW + "
                         "all-gather, 'allreduce' = flat gradient all-reduce; 'auto' takes the cheapest under the cost model of "
                         "eavqa_amd.trainers.optim.choose_dp_exchange (DESIGN.md section 7)")
    ap.add_argument("--mapping-type", choices=["mlp", "transformer"], default=None, help="override the workload's mapper")
    ap.add_argument("--no-overlap", action="store_true", help="run the CLIP encode on the main stream (no cross-step pipelining in the training legs, no look-ahead encode in the few-shot legs)")
    ap.add_argument("--no-fewshot", action="store_true", help="skip the few-shot generate leg (metric M2, reported under 'extra')")
    ap.add_argument("--no-extra-train", action="store_true",
                    help="skip the short cfg3 (bf16) and cfg5 (fp8) training legs that the default cfg2 run reports under 'extra'")
    ap.add_argument("--grad-transport", choices=["f32", "bf16"], default="f32",
                    help="N > 1, sharded exchange: dtype of the gradient on the wire (bf16: 4 instead of 6 B / parameter, the sum rounded per hop)")
    ap.add_argument("--pipelined-update", action="store_true",
                    help="AdamW chunk by chunk on its own stream, the mapper's forward waiting per layer (FusedAdamW.step(chunks=...)); default: one "
                         "launch on the main stream in front of the forward.  Measured equal on one GPU (cfg2 4763 / 4767, cfg5 525.0 / 523.7 "
                         "samples/s: the HBM-bound update and the GEMMs it would hide under slow each other down)")
    ap.add_argument("--no-t0", action="store_true",
                    help="skip the two T0_3B legs (the reference's headline model: few-shot generate and CC mapper training) reported under 'extra'")
    ap.add_argument("--hbm-bytes-out", default=None,
                    help="write the algorithmic bytes per launch of the HBM-bound ops of one step as JSON (input of tools/round_profile_report.py)")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: whatever libraries print there meanwhile (gloo's "[Gloo] Rank 0 is connected ...", RCCL / HIP
    # notices) is sent to stderr - file descriptor 1 points at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from eavqa_amd import _lib
    from eavqa_amd.trainers.data_parallel import init_from_env

    rank, local, world = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP extension is the only compute path")
    # EAVQA_FORCE_DEVICE: rehearse the N > 1 path on a one-GPU box (all ranks on one card, gloo backend)
    dev_index = int(os.environ.get("EAVQA_FORCE_DEVICE", local))
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if _lib.load().eavqa_check_device() != 0:
        raise SystemExit("device is not gfx950")
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16

    line = train_leg(args.workload, args.dtype, args.steps, args.warmup, rank, world, device, args)

    extra = []
    solo = rank == 0 and world == 1
    if solo and not args.no_fewshot:
        try:
            log("few-shot generate leg ...")
            extra.append(fewshot_qps(dtype, device, encode_ahead=not args.no_overlap))
            log(f"few-shot leg done: {extra[-1]}")
        except Exception as e:                # never lose the headline line to a secondary metric
            extra.append({"metric": "fewshot_vqa_questions_per_sec", "error": repr(e)[:300]})
    if solo and not args.no_extra_train and args.workload == "cfg2" and args.dtype == "bf16" and not args.mapping_type:
        for name, dt_name, k, wu in EXTRA_TRAIN_LEGS:
            try:
                leg_args = argparse.Namespace(**{**vars(args), "hbm_bytes_out": None})
                extra.append(train_leg(name, dt_name, k, wu, rank, world, device, leg_args, log_prefix=f"[{name} {dt_name}] "))
            except Exception as e:
                extra.append({"metric": "mapper_train_samples_per_sec", "config": {"workload": name}, "dtype": dt_name, "error": repr(e)[:300]})
                torch.cuda.empty_cache()
    if solo and not args.no_t0 and args.workload == "cfg2" and args.dtype == "bf16" and not args.mapping_type:
        for name, fn in (("t0_3b_fewshot", lambda: t0_fewshot_qps(dtype, device, encode_ahead=not args.no_overlap)), ("t0_3b_cc_train", lambda: t0_cc_train_leg("bf16", 10, 5, device, args))):
            try:
                log(f"{name} leg ...")
                extra.append(fn())
                log(f"{name} leg done: {extra[-1]}")
            except Exception as e:
                extra.append({"metric": name, "error": repr(e)[:300]})
                torch.cuda.empty_cache()
    cpu = None
    if solo and args.cpu_baseline_samples > 0:
        log(f"cpu baseline on {host_threads()} threads ...")
        cpu = cpu_baseline(args.workload, args.cpu_baseline_samples, host_threads())
        log(f"cpu baseline done: {cpu}")

    if rank == 0:
        line["cpu_baseline"] = cpu
        line["extra"] = extra or None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    os.close(json_fd)


def gemm_source_digest():
    """sha256 (first 12 hex digits) over the GEMM kernel sources: stamps the PMC traffic file, so that a traffic figure measured
    on other kernels is never paired with this run's timings."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_k64.hip", "gemm_fp8.hip", "common.h"):
        with open(os.path.join(ROOT, "explicit-alignment-for-vqa-tasks_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


# HBM-bound ops of a step: (ops attribute, rocprof kernel-name pattern, algorithmic bytes of one call).  Bytes = every operand read once and
# every result written once; the patterns join this table with a rocprofv3 kernel-stats CSV in
# tools/round_profile_report.py.
def _hbm_models():
    es = lambda t: t.element_size()

    def ln_fwd(x, gamma, beta, eps, out_dtype, save_stats=False, out=None):
        rows, cols = x.shape
        return rows * cols * (es(x) + torch.empty(0, dtype=out_dtype).element_size()) + (8 * rows if save_stats else 0)

    def ln_bwd(x, dy, gamma, mean, rstd, dres=None, dgamma=None, dbeta=None, out=None, lowp_out=None):
        rows, cols = x.shape
        return rows * cols * (es(x) + es(dy) + (4 if dres is not None else 0) + 4 + (es(lowp_out) if lowp_out is not None else 0)) + 8 * rows

    def ce_fwd(logits, labels, V):
        return logits.shape[0] * V * 4

    def ce_bwd(logits, labels, V, row_lse, count, gscale, dtype, ldd):
        return logits.shape[0] * (V * 4 + ldd * torch.empty(0, dtype=dtype).element_size())

    def adamw(param, grad, m, v, *a, shadow=None, **k):
        return param.numel() * (28 + (es(shadow) if shadow is not None else 0))      # p, m, v read + written, grad read, shadow written

    def quant(x):
        return x.shape[0] * x.shape[1] * (es(x) + 1) + 4 * x.shape[0]

    def transpose(x, out=None):
        return 2 * x.numel() * es(x)

    return {"layernorm_fwd": ("ln_fwd_kernel", ln_fwd), "layernorm_bwd": ("ln_bwd_kernel", ln_bwd), "ce_fwd": ("ce_fwd", ce_fwd),
            "ce_bwd": ("ce_bwd_kernel", ce_bwd), "adamw": ("adamw_kernel", adamw), "quantize_rows_fp8": ("quantize_rows_fp8", quant),
            "transpose": ("transpose", transpose)}


def step_roofline(vit, model, opt, batch, pad, sync, ops, workload="cfg2", dtype_name="bf16", ms_per_step=None, hbm_bytes_out=None):
    """HIP events (torch.cuda.Event records on the current stream = the stream the kernels are launched on) around every GEMM launch
    and every HBM-bound op of ONE extra step that runs right behind the timed region.  That step runs the CLIP encode ON THE MAIN
    STREAM (no cross-step overlap), so no bracket contains another stream's kernels: a bracket on the main stream beside a busy side
    stream would charge the contention to the GEMM (round 2's cfg3 line reported 43.8 ms of GEMM inside a 30.6 ms step that way).
      roofline      = the GEMM launches of the timed step's MAIN stream (mapper + LM forward / dgrad / wgrad): achieved = algorithmic
                      FLOPs / sum of their durations.  They run one after another on that stream in the timed step too (beside the
                      side stream they can only be slower), so gemm_ms_per_step <= ms_per_step by construction - checked here.
      vit_tower_gemms = the CLIP tower's GEMMs (the side stream's work in the timed step), same measurement, reported beside it.
      hbm_kernels   = LayerNorm fwd / bwd, CE fwd / bwd, AdamW, fp8 row quantiser, transposes: algorithmic bytes / duration / 8 TB/s.
    With --dtype fp8 the dominant kernel is the fp8 GEMM (eavqa_gemm_fp8, priced against the 5 PFLOP/s dense fp8 peak); the bf16
    launches of the main stream (the mapper) are reported beside it."""
    real = {"gemm": ops.gemm, "gemm_fp8": ops.gemm_fp8}
    models = _hbm_models()
    for name in models:
        real[name] = getattr(ops, name)
    recs = []
    state = {"phase": "main"}

    def bracket(kind, flops, bytes_, call):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = call()
        e1.record()
        # an empty pair right behind it: what two event markers cost by themselves on this stream (subtracted below)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        c1.record()
        recs.append((kind, state["phase"], flops, bytes_, e0, e1, c0, c1))
        return out

    def timed(a, b, *, a_kc=True, b_kc=True, **kw):
        M, K = a.shape if a_kc else (a.shape[1], a.shape[0])
        N = b.shape[0] if b_kc else b.shape[1]
        o = kw.get("out")
        out_b = 4 if (kw.get("out_f32") or (o is not None and o.dtype == torch.float32)) else a.element_size()
        aux_b = a.element_size() if (kw.get("aux_in") is not None or kw.get("aux_out") is not None) else 0
        res = kw.get("residual")
        res_b = res.element_size() if res is not None else 0
        algo = (M * K + N * K) * a.element_size() + M * N * (out_b + aux_b + res_b)
        return bracket("bf16", 2.0 * M * N * K, algo, lambda: real["gemm"](a, b, a_kc=a_kc, b_kc=b_kc, **kw))

    def timed8(aq, a_scale, bq, b_scale, **kw):
        M, K = aq.shape
        N = bq.shape[0]
        o = kw.get("out")
        out_b = 4 if (kw.get("out_f32") or (o is not None and o.dtype == torch.float32)) else 2
        aux_b = 2 if (kw.get("aux_in") is not None or kw.get("aux_out") is not None) else 0
        res_b = 4 if kw.get("residual") is not None else 0
        algo = (M * K + N * K) + M * N * (out_b + aux_b + res_b)
        return bracket("fp8", 2.0 * M * N * K, algo, lambda: real["gemm_fp8"](aq, a_scale, bq, b_scale, **kw))

    def hbm_wrapper(name):
        fn, model_bytes = real[name], models[name][1]
        return lambda *a, **k: bracket("hbm:" + name, 0.0, float(model_bytes(*a, **k)), lambda: fn(*a, **k))

    serial = Stepper(vit, model, opt, batch, pad, sync, overlap_vit=False, pipelined_update=False)      # one stream: clean brackets
    enc = vit.encode_image

    def tagged_encode(px):
        state["phase"] = "vit"
        try:
            return enc(px)
        finally:
            state["phase"] = "main"

    ops.gemm, ops.gemm_fp8 = timed, timed8
    for name in models:
        setattr(ops, name, hbm_wrapper(name))
    vit.encode_image = tagged_encode
    try:
        # head start: keep the GPU busy for > 100 ms so that the whole instrumented step is enqueued before the GPU
        # reaches it - the event pairs then bracket pure device time, not host latency between record and launch
        blk_a = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
        blk_c = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
        for _ in range(150):                      # ~0.9 ms each
            real["gemm"](blk_a, blk_a, out=blk_c)
        serial.step()
        serial.flush()
        torch.cuda.synchronize()
    finally:
        for name, fn in real.items():
            setattr(ops, name, fn)
        vit.encode_image = enc
        del blk_a, blk_c

    def seconds(rs):
        raw = sum(r[4].elapsed_time(r[5]) for r in rs) * 1e-3
        marker = sum(r[6].elapsed_time(r[7]) for r in rs) * 1e-3      # event-marker cost, per pair on average marker / n
        return max(raw - marker, 0.5 * raw), marker

    def summarise(kind, phase, peak):
        rs = [r for r in recs if r[0] == kind and r[1] == phase]
        if not rs:
            return None
        flops, algo, n = sum(r[2] for r in rs), sum(r[3] for r in rs), len(rs)
        secs, marker = seconds(rs)
        ach = flops / secs / 1e12
        return {"achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "launches_per_step": n,
                "gflop_per_launch": round(flops / n / 1e9, 2), "avg_launch_us": round(secs / n * 1e6, 2),
                "event_marker_us": round(marker / n * 1e6, 2), "gemm_ms_per_step": round(secs * 1e3, 3),
                "algorithmic_bytes_per_launch": round(algo / n)}

    hbm, hbm_dump = [], {}
    for name, (pattern, _) in models.items():
        rs = [r for r in recs if r[0] == "hbm:" + name]
        if not rs:
            continue
        secs, _ = seconds(rs)
        byts = sum(r[3] for r in rs)
        gbps = byts / secs / 1e9
        hbm.append({"op": "eavqa_" + name, "kernel": pattern, "launches_per_step": len(rs), "bytes_per_launch": round(byts / len(rs)),
                    "avg_launch_us": round(secs / len(rs) * 1e6, 2), "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(gbps / 8000.0, 4)})
        hbm_dump[name] = {"kernel_pattern": pattern, "launches_per_step": len(rs), "bytes_per_step": byts}
    if hbm_bytes_out:
        with open(hbm_bytes_out, "w") as f:
            json.dump({"workload": workload, "dtype": dtype_name, "ops": hbm_dump}, f, indent=1)

    if dtype_name == "f32":
        r = summarise("bf16", "main", 157.3)
        return {"bound": "mfma", "kernel": "eavqa_gemm: gemm_f32_kernel (v_mfma_f32_32x32x2_f32)", **r, "traffic": None, "traffic_source": None,
                "vit_tower_gemms": summarise("bf16", "vit", 157.3), "hbm_kernels": hbm}
    main_kind = "fp8" if dtype_name == "fp8" else "bf16"
    r = summarise(main_kind, "main", 5000.0 if main_kind == "fp8" else 2500.0)
    # HBM bytes per GEMM launch from committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs, FETCH_SIZE x 2 on gfx950, KiB units; tools/pmc_traffic.py).  The file carries the digest of the GEMM sources it was
    # measured on: with other sources the figure is dropped, not reused.
    traffic, traffic_source = None, None
    for rnd in ("round4", "round3", "round2"):
        tfile = os.path.join(ROOT, "profiles", f"{rnd}_gemm_traffic_{workload}_{dtype_name}.json")
        if os.path.exists(tfile):
            break
    if os.path.exists(tfile):
        with open(tfile) as f:
            t = json.load(f)
        if t.get("gemm_source_digest") == gemm_source_digest():
            fam = t.get("families", {}).get(main_kind, t)            # the dominant kernel's own launches (fp8 / bf16), not the mix
            traffic, traffic_source = round(fam["bytes_per_launch"]), f"profiles/{os.path.basename(tfile)} @ gemm sources {t['gemm_source_digest']}"
        else:
            traffic_source = f"dropped: {os.path.basename(tfile)} was measured on gemm sources {t.get('gemm_source_digest')}, this run is {gemm_source_digest()}"
    kernel = ("eavqa_gemm_fp8: gemm_fp8_k128s_kernel (v_mfma_scale_f32_16x16x128_f8f6f4, loader / consumer specialised full-line tiles)"
              if main_kind == "fp8" else
              "eavqa_gemm: gemm_bf16_k64s_kernel (loader / consumer specialised full-line tiles 128x80 / 256x128 / 256x160 / 128x128 / 128x256) "
              "+ gemm_bf16_big_kernel (256x256)")
    out = {"bound": "mfma", "kernel": kernel, "scope": "GEMM launches of the step's main stream (mapper + frozen LM); CLIP tower: vit_tower_gemms",
           **r, "traffic": traffic, "traffic_source": traffic_source}
    if ms_per_step is not None:
        out["gemm_ms_within_step"] = bool(r["gemm_ms_per_step"] <= ms_per_step)
    if main_kind == "fp8":
        out["bf16_gemms_same_step"] = summarise("bf16", "main", 2500.0)
    out["vit_tower_gemms"] = summarise("bf16", "vit", 2500.0)
    out["hbm_kernels"] = hbm
    return out


if __name__ == "__main__":
    main()
