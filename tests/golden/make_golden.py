#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference and transformers):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it executes (nothing from the reference is copied into this repo; the .npz
files hold inputs, seeded random-init weights and the reference's outputs):

* ``/root/reference/src/models/clipcap.py`` - ``MLP``, ``TransformerMapper``,
  ``ClipCaptionPrefix.forward`` / ``.generate`` on locally saved random-init
  ``GPT2LMHeadModel`` (eager attention), via the import shim SURVEY.md 8(c)
  describes (``transformers.AdamW`` no longer exists in transformers 5.x).
* the same reference ``forward``/``generate`` code driving an HF ``OPTForCausalLM``
  (the reference hard-wires GPT-2, SURVEY F4: the OPT fixture swaps ``model.gpt``
  for the OPT model and aliases ``.transformer.wte`` so the reference's own
  concat/label/greedy code runs unchanged).
* ``/root/reference/src/models/vct0.py`` ``VCT0Model.insert_prefix_into_input``
  (called unbound with a namespace ``self``; it only reads ``prefix_length`` and
  ``lm_embedding_size``) on the two cases of ``src/models/vct0_test.py`` and one
  4-shot case.
* HF ``CLIPVisionModelWithProjection`` (quick_gelu) as the stand-in for OpenAI CLIP's
  ``encode_image`` (the ``clip`` package is absent and un-pinned).

Known-answer tables that need no execution (label masking, restated from
``src/trainers/clipcap_exector.py:134-150`` because the executor module cannot be
imported here) are written from hand-derived expectations.
"""
import os
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

import numpy as np
import torch
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/models"


def _import_reference():
    # shim 1: symbol imported by the reference, never used.  transformers 5.x swaps its lazy
    # module object in sys.modules while resolving names, so re-apply before each import.
    sys.modules["transformers"].AdamW = torch.optim.AdamW
    stub = types.ModuleType("flamingo_pytorch")  # shim 2: only mapping_type="perceiver" needs it
    stub.PerceiverResampler = type("PerceiverResampler", (torch.nn.Module,), {})
    sys.modules.setdefault("flamingo_pytorch", stub)
    sys.path.insert(0, REF)
    import clipcap  # noqa
    sys.modules["transformers"].AdamW = torch.optim.AdamW
    import vct0  # noqa
    return clipcap, vct0


def _np(sd, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays")


def ragged_batch(gen, B, T, V, pad_id, min_len):
    lens = torch.randint(min_len, T + 1, (B,), generator=gen)
    lens[0] = T
    ids = torch.randint(0, V - 2, (B, T), generator=gen)
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    ids = ids * mask + pad_id * (1 - mask)
    return ids, mask


def main():
    clipcap, vct0 = _import_reference()
    from transformers import (CLIPVisionConfig, CLIPVisionModelWithProjection, GPT2Config, GPT2LMHeadModel,
                              OPTConfig, OPTForCausalLM)

    tmp = tempfile.mkdtemp(prefix="eavqa_golden_")
    gen = torch.Generator().manual_seed(2021)  # reference seed, configs/vqa2/clip_cap.jsonnet:17

    # ------------------------------------------------------------------ mappers
    torch.manual_seed(2021)
    D, E, L = 24, 32, 4
    mlp = clipcap.MLP((D, (E * L) // 2, E * L))
    x = torch.randn(3, D, generator=gen)
    gy = torch.randn(3, E * L, generator=gen)
    y = mlp(x)
    (y * gy).sum().backward()
    save("mapper_mlp.npz", x=x.numpy(), gy=gy.numpy(), y=y.detach().numpy(),
         **_np(mlp.state_dict(), "w."), **{"g." + n: p.grad.numpy() for n, p in mlp.named_parameters()})

    torch.manual_seed(2021)
    E, L, CL, NL = 64, 4, 3, 2  # 8 heads fixed -> hd 8
    tm = clipcap.TransformerMapper(D, E, L, CL, NL)
    x = torch.randn(3, D, generator=gen)
    gy = torch.randn(3, L, E, generator=gen)
    y = tm(x)
    (y * gy).sum().backward()
    save("mapper_transformer.npz", x=x.numpy(), gy=gy.numpy(), y=y.detach().numpy(),
         cfg=np.array([D, E, L, CL, NL]),
         **_np(tm.state_dict(), "w."), **{"g." + n: p.grad.numpy() for n, p in tm.named_parameters()})

    # ------------------------------------------------------------------ tiny GPT-2 + reference model
    def build_gpt2(seed, V, E, n_layer, n_head, n_pos):
        torch.manual_seed(seed)
        cfg = GPT2Config(vocab_size=V, n_embd=E, n_layer=n_layer, n_head=n_head, n_positions=n_pos,
                         bos_token_id=V - 1, eos_token_id=V - 1, attn_implementation="eager",
                         resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
        m = GPT2LMHeadModel(cfg)
        d = os.path.join(tmp, f"gpt2_{seed}_{E}")
        m.save_pretrained(d)
        return d, cfg

    V, E, NLAY, NH, L, D = 320, 64, 2, 4, 4, 24
    pad_id = eos_id = V - 1
    d_gpt2, hfcfg = build_gpt2(2021, V, E, NLAY, NH, 64)
    for mapping_type in ("mlp", "transformer"):
        torch.manual_seed(7)
        model = clipcap.ClipCaptionPrefix(prefix_length=L, clip_length=3, prefix_size=D, num_layers=2,
                                          mapping_type=mapping_type, model_version=d_gpt2)
        model.gpt.config._attn_implementation = "eager"
        model.train()
        B, T = 4, 9
        ids, mask = ragged_batch(gen, B, T, V, pad_id, 3)
        labels = ids.clone()
        labels[labels == pad_id] = -100
        prefix = torch.randn(B, D, generator=gen)
        out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels, pad_token_id=pad_id)
        out.loss.backward()
        grads = {"g." + n: p.grad.numpy() for n, p in model.clip_project.named_parameters()}
        # generation: first a free run to learn which tokens appear, then force an early EOS row
        model.eval()
        gids, gmask = ragged_batch(gen, B, 6, V, pad_id, 2)
        with torch.no_grad():
            free = model.generate(question_tokens=gids, prefix=prefix, question_mask=gmask, max_length=6,
                                  pad_token_id=pad_id, eos_token_id=None)
            forced_eos = int(free[1][2])  # row 1 finishes after emitting its 3rd token
            forced = model.generate(question_tokens=gids, prefix=prefix, question_mask=gmask, max_length=6,
                                    pad_token_id=pad_id, eos_token_id=forced_eos)
            all_eos = int(free[0][0])
            allrows = [int(r[0]) for r in free]
            early = model.generate(question_tokens=gids[:1], prefix=prefix[:1], question_mask=gmask[:1],
                                   max_length=6, pad_token_id=pad_id, eos_token_id=all_eos)
        save(f"clipcap_gpt2_{mapping_type}.npz",
             cfg=np.array([V, E, NLAY, NH, 64, L, D, 3, 2]),  # V,E,n_layer,n_head,n_pos,L,D,clip_length,num_layers
             ids=ids.numpy(), mask=mask.numpy(), labels=labels.numpy(), prefix=prefix.numpy(), pad_id=np.array(pad_id),
             loss=out.loss.detach().numpy(), logits=out.logits.detach().numpy(),
             gen_ids=gids.numpy(), gen_mask=gmask.numpy(),
             gen_free=np.array(free), gen_forced=np.array(forced), gen_forced_eos=np.array(forced_eos),
             gen_early=np.array(early), gen_early_eos=np.array(all_eos), gen_first=np.array(allrows),
             **_np(model.gpt.state_dict(), "lm."), **_np(model.clip_project.state_dict(), "map."), **grads)

    # ------------------------------------------------------------------ tiny OPT driven by the reference wrapper code
    torch.manual_seed(11)
    V, E, NLAY, NH, FFN = 336, 64, 2, 4, 96
    ocfg = OPTConfig(vocab_size=V, hidden_size=E, num_hidden_layers=NLAY, num_attention_heads=NH, ffn_dim=FFN,
                     max_position_embeddings=64, word_embed_proj_dim=E, pad_token_id=1, bos_token_id=2,
                     eos_token_id=2, attn_implementation="eager", dropout=0.0, attention_dropout=0.0)
    opt = OPTForCausalLM(ocfg).eval()
    torch.manual_seed(13)
    model = clipcap.ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", model_version=d_gpt2)
    # same MLP shapes (E equal); swap the LM, keep the reference's forward/generate code
    opt.transformer = types.SimpleNamespace(wte=opt.model.decoder.embed_tokens)
    model.gpt = opt
    model.train()
    pad_id = 1
    B, T = 4, 9
    ids, mask = ragged_batch(gen, B, T, V, pad_id, 3)
    ids = ids.clamp(min=3)
    ids = ids * mask + pad_id * (1 - mask)
    labels = ids.clone()
    labels[labels == pad_id] = -100
    prefix = torch.randn(B, D, generator=gen)
    out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels, pad_token_id=pad_id)
    out.loss.backward()
    grads = {"g." + n: p.grad.numpy() for n, p in model.clip_project.named_parameters()}
    model.eval()
    gids, gmask = ragged_batch(gen, B, 6, V, pad_id, 2)
    gids = (gids.clamp(min=3)) * gmask + pad_id * (1 - gmask)
    with torch.no_grad():
        free = model.generate(question_tokens=gids, prefix=prefix, question_mask=gmask, max_length=5,
                              pad_token_id=pad_id, eos_token_id=None)
        forced_eos = int(free[2][1])
        forced = model.generate(question_tokens=gids, prefix=prefix, question_mask=gmask, max_length=5,
                                pad_token_id=pad_id, eos_token_id=forced_eos)
    sd = {k: v for k, v in opt.state_dict().items()}
    save("clipcap_opt_mlp.npz",
         cfg=np.array([V, E, NLAY, NH, 64, L, D, FFN]),
         ids=ids.numpy(), mask=mask.numpy(), labels=labels.numpy(), prefix=prefix.numpy(), pad_id=np.array(pad_id),
         loss=out.loss.detach().numpy(), logits=out.logits.detach().numpy(),
         gen_ids=gids.numpy(), gen_mask=gmask.numpy(), gen_free=np.array(free), gen_forced=np.array(forced),
         gen_forced_eos=np.array(forced_eos),
         **_np(sd, "lm."), **_np(model.clip_project.state_dict(), "map."), **grads)

    # ------------------------------------------------------------------ insert_prefix_into_input
    def ref_insert(L, E, num_shots, toks, text, pp, masks, special=32099):
        self_ = types.SimpleNamespace(prefix_length=L, lm_embedding_size=E)
        emb, msk = vct0.VCT0Model.insert_prefix_into_input(self_, toks.shape[0], num_shots, toks, text, pp, masks,
                                                           special_token_id=special)
        return emb, msk

    # the two known-answer cases of src/models/vct0_test.py:79-211 (values are test DATA, re-typed here)
    text = torch.tensor([[[100., 101, 102], [103, 104, 105], [106, 107, 108], [109, 110, 111], [130, 131, 132],
                          [133, 134, 135], [99, 98, 97]],
                         [[112., 113, 114], [115, 116, 117], [117, 118, 119], [120, 121, 122], [140, 141, 142],
                          [143, 144, 145], [96, 95, 94]]])
    pp_few = -torch.tensor([[[[100., 101, 102], [103, 104, 105]], [[106, 107, 108], [109, 110, 111]],
                             [[130, 131, 132], [133, 134, 135]]],
                            [[[112., 113, 114], [115, 116, 117]], [[117, 118, 119], [120, 121, 122]],
                             [[140, 141, 142], [143, 144, 145]]]])
    pp_zero = pp_few[:, :1].contiguous()
    masks = torch.tensor([[1, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1]], dtype=int)
    tok0 = torch.tensor([[32099, 20414, 11, 11, 11, 48, 0], [20414, 32099, 11, 48, 48, 48, 10]], dtype=int)
    tok2 = torch.tensor([[32099, 20414, 32098, 11, 32097, 48, 0], [20414, 32099, 11, 32098, 48, 32097, 10]], dtype=int)
    e0, m0 = ref_insert(2, 3, 0, tok0, text, pp_zero, masks)
    e2, m2 = ref_insert(2, 3, 2, tok2, text, pp_few, masks)
    # the expected values spelled out in vct0_test.py (checked here against the reference's own output)
    exp_m0 = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1, 1]])
    exp_m2 = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 1, 1, 0], [1] * 10])
    assert torch.equal(m0, exp_m0) and torch.equal(m2, exp_m2)
    exp_e2_row0 = torch.stack([*pp_few[0, 0], text[0, 1], *pp_few[0, 1], text[0, 3], *pp_few[0, 2], text[0, 5], text[0, 6]])
    assert torch.equal(e2[0], exp_e2_row0)
    # 4-shot case, L=3, E=5, random text
    B, T, E5, L3, shots = 3, 17, 5, 3, 4
    toks = torch.randint(5, 1000, (B, T), generator=gen)
    for b in range(B):
        pos = torch.randperm(T, generator=gen)[: shots + 1].sort().values
        for i, p_ in enumerate(pos):
            toks[b, p_] = 32099 - i
    text4 = torch.randn(B, T, E5, generator=gen)
    pp4 = torch.randn(B, shots + 1, L3, E5, generator=gen)
    m4 = (torch.rand(B, T, generator=gen) > 0.2).long()
    e4, mm4 = ref_insert(L3, E5, shots, toks, text4, pp4, m4)
    save("insert_prefix.npz",
         z_tok=tok0.numpy(), z_text=text.numpy(), z_pp=pp_zero.numpy(), z_mask=masks.numpy(), z_emb=e0.numpy(), z_out_mask=m0.numpy(),
         f_tok=tok2.numpy(), f_text=text.numpy(), f_pp=pp_few.numpy(), f_mask=masks.numpy(), f_emb=e2.numpy(), f_out_mask=m2.numpy(),
         s_tok=toks.numpy(), s_text=text4.numpy(), s_pp=pp4.numpy(), s_mask=m4.numpy(), s_emb=e4.numpy(), s_out_mask=mm4.numpy(),
         s_cfg=np.array([L3, E5, shots]))

    # ------------------------------------------------------------------ label masking known answers
    # HAND-DERIVED from src/trainers/clipcap_exector.py:134-150 (pad == eos == 9, bos == 7); the executor cannot be imported here
    # (pytorch_lightning / wandb absent).  The rule, line by line:
    #   :135  labels = input_ids.clone()                      :136  labels[labels == pad] = -100      (EVERY pad, also inside text)
    #   :138  per row i, walk j = 0 .. T-1 with answer_tokens = False:
    #   :141-143    token == -100 (a pad)   -> labels[i, j] = pad (== eos: the end-of-answer target), then BREAK out of the row
    #   :144-147    token == <BOS>          -> answer_tokens = True, labels[i, j] = -100, continue
    #   :148-149    answer_tokens           -> keep the token as its own label
    #   :150        otherwise (question)    -> labels[i, j] = -100
    # Consequences the rows below spell out:
    #   row 0  question 3 4 | <BOS> | answer 5 6 | first pad restored to 9 | later pads stay -100 (set at :136, never visited: break)
    #   row 1  same with a one-token answer
    #   row 2  NO pad in the row: the walk never breaks and nothing is restored; tokens after <BOS> (8) are the only labels
    #   row 3  NO <BOS>: the whole question is masked, the first pad still becomes 9 (a lone eos target)
    #   row 4  TWO <BOS>: both are -100 themselves, every other token after the first one is kept
    # The "untouched tail" quirk (eavqa_amd.trainers.clipcap_executor.vqa_label_count): positions BEHIND the first pad are never
    # visited, so a non-pad token there (possible only with pad tokens inside the text, e.g. 3 9 4 5) keeps labels[i, j] =
    # input_ids[i, j] from :135 - it is scored.  tests/test_executor_gpu.py::test_vqa_label_count_matches_the_masking_rule covers it.
    lm_in = np.array([[3, 4, 7, 5, 6, 9, 9, 9],      # question 3 4, <BOS>, answer 5 6, then pads
                      [3, 7, 5, 9, 9, 9, 9, 9],
                      [3, 4, 5, 6, 2, 1, 7, 8],      # no pad at all: answer = last token only
                      [3, 4, 5, 6, 9, 9, 9, 9],      # no <BOS>: everything masked, first pad restored
                      [7, 5, 7, 6, 9, 9, 9, 9]])     # two <BOS>: both masked, tokens after the first kept
    lm_out = np.array([[-100, -100, -100, 5, 6, 9, -100, -100],
                       [-100, -100, 5, 9, -100, -100, -100, -100],
                       [-100, -100, -100, -100, -100, -100, -100, 8],
                       [-100, -100, -100, -100, 9, -100, -100, -100],
                       [-100, 5, -100, 6, 9, -100, -100, -100]])
    save("label_mask.npz", input_ids=lm_in, labels=lm_out, pad_id=np.array(9), bos_id=np.array(7))

    # ------------------------------------------------------------------ tiny CLIP ViT
    torch.manual_seed(5)
    ccfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                            image_size=48, patch_size=16, projection_dim=24, hidden_act="quick_gelu",
                            attn_implementation="eager")
    clip = CLIPVisionModelWithProjection(ccfg).eval()
    # HF init leaves class_embedding ~N(0,1)*scale; fine.
    px = torch.randn(3, 3, 48, 48, generator=gen)
    with torch.no_grad():
        emb = clip(pixel_values=px).image_embeds
    save("clip_vit.npz", cfg=np.array([64, 128, 2, 4, 48, 16, 24]), pixels=px.numpy(), image_embeds=emb.numpy(),
         **_np(clip.state_dict(), "w."))

    # a ViT with a patch size that is not a multiple of 8 (ViT-L/14-like: K = 3*14*14 = 588)
    torch.manual_seed(6)
    ccfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
                            image_size=42, patch_size=14, projection_dim=24, hidden_act="quick_gelu",
                            attn_implementation="eager")
    clip = CLIPVisionModelWithProjection(ccfg).eval()
    px = torch.randn(2, 3, 42, 42, generator=gen)
    with torch.no_grad():
        emb = clip(pixel_values=px).image_embeds
    save("clip_vit_p14.npz", cfg=np.array([64, 128, 1, 4, 42, 14, 24]), pixels=px.numpy(), image_embeds=emb.numpy(),
         **_np(clip.state_dict(), "w."))


def vct0_golden():
    """The reference's own ``VCT0Prefix`` (src/models/vct0.py:301-549) on two tiny random-init ``T5ForConditionalGeneration`` saved locally
    (``model_version=<directory>``): v1.1 / T0 style (gated gelu_new FFN, untied lm_head) and v1.0 style (ReLU FFN, tied embeddings with the
    d_model^-0.5 rescale).  Outputs: ``forward(prefix, labels)`` loss / logits / mapper gradients (the CC training step of
    src/trainers/vct0_exector.py:143-146), and greedy ``generate`` ids + per-step scores for (a) the prefix-only path (:485-491), (b) the
    few-shot interleaved path (:452-466), (c) one example at a time (:426-442), (d) text only (:409-424).  The reference hard-codes T5's
    sentinel id 32099 in ``generate``; the tiny vocabulary has none, so the subclass below maps ``special_token_id`` 32099 - i to V - 1 - i
    before delegating - the function under test is otherwise the reference's."""
    clipcap, vct0 = _import_reference()
    from transformers import T5Config, T5ForConditionalGeneration
    tmp = tempfile.mkdtemp(prefix="eavqa_vct0_")
    V, E, DKV, H, F, NL, L, D = 96, 64, 16, 4, 128, 2, 3, 24
    out = {}
    for tag, gated, tied in (("t0", True, False), ("t5v10", False, True)):
        torch.manual_seed(2021)
        cfg = T5Config(vocab_size=V, d_model=E, d_kv=DKV, num_heads=H, d_ff=F, num_layers=NL, num_decoder_layers=NL, dropout_rate=0.0,
                       feed_forward_proj="gated-gelu" if gated else "relu", tie_word_embeddings=tied, decoder_start_token_id=0, pad_token_id=0,
                       eos_token_id=1, relative_attention_num_buckets=32, relative_attention_max_distance=128)
        cfg._attn_implementation = "eager"
        lm = T5ForConditionalGeneration(cfg).eval()
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():                  # HF initialises the layer norms to ones and biases the init by a factor: make every tensor informative
            for n, p in lm.named_parameters():
                if "layer_norm" in n:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                elif "relative_attention_bias" in n:
                    p.copy_(0.5 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(p * 1.0 + 0.05 * torch.randn(p.shape, generator=g))
        path = os.path.join(tmp, tag)
        lm.save_pretrained(path)

        class TinyVCT0(vct0.VCT0Prefix):
            def insert_prefix_into_input(self, *a, special_token_id=32099, **k):
                return super().insert_prefix_into_input(*a, special_token_id=special_token_id - 32099 + (V - 1), **k)

        torch.manual_seed(2021)
        model = TinyVCT0(prefix_length=L, prefix_size=D, mapping_type="mlp", model_version=path).eval()
        model.lm.config._attn_implementation = "eager"
        sd = {k: v.detach().clone() for k, v in model.lm.state_dict().items()}
        gen = torch.Generator().manual_seed(11)
        # ---- forward(prefix, labels): CC training step
        B, T = 3, 7
        prefix = torch.randn(B, D, generator=gen)
        labels = torch.randint(2, V - 8, (B, T), generator=gen)
        labels[1, 5:] = -100
        labels[2, 3:] = -100
        labels[0, T - 1] = 1                                      # eos
        res = model(prefix=prefix, labels=labels)
        res.loss.backward()
        arrs = dict(cfg=np.array([V, E, DKV, H, F, NL, L, D, int(gated), int(tied)]), prefix=prefix.numpy(), labels=labels.numpy(),
                    loss=res.loss.detach().numpy(), logits=res.logits.detach().numpy())
        arrs.update(_np(sd, "lm."))
        arrs.update(_np(model.clip_project.state_dict(), "map."))
        arrs.update({"gmap." + n: p.grad.numpy() for n, p in model.clip_project.named_parameters()})
        # ---- generate: prefix only
        kw = dict(max_length=9, output_scores=True, return_dict_in_generate=True, do_sample=False, num_beams=1)
        with torch.no_grad():
            o = model.generate(prefix=prefix, **kw)
        arrs["gen_prefix_ids"] = o.sequences.numpy()
        arrs["gen_prefix_scores"] = torch.stack(o.scores).numpy()
        # ---- generate: few-shot (2 shots + query = 3 images), ragged right padding
        n_img, Tq = 3, 12
        q = torch.randint(2, V - 8, (B, Tq), generator=gen)
        for b_ in range(B):
            for i, pos in enumerate(sorted(torch.randperm(Tq - 3, generator=gen)[:n_img].tolist())):
                q[b_, pos] = V - 1 - i
        qm = torch.ones(B, Tq, dtype=torch.long)
        qm[1, Tq - 2:] = 0
        q[1, Tq - 2:] = 0
        pf = torch.randn(B, n_img, 1, D, generator=gen)
        with torch.no_grad():
            o = model.generate(prefix=pf, question_tokens=q, question_mask=qm, **kw)
        arrs.update(fs_tokens=q.numpy(), fs_mask=qm.numpy(), fs_prefix=pf.numpy(), gen_fs_ids=o.sequences.numpy(), gen_fs_scores=torch.stack(o.scores).numpy())
        # ---- generate: one example at a time (each [B, n, T1] row holds ONE sentinel, V - 1 - i for example i)
        T1 = 6
        q1 = torch.randint(2, V - 8, (B, n_img, T1), generator=gen)
        for i in range(n_img):
            q1[:, i, 1 + i] = V - 1 - i
        qm1 = torch.ones(B, n_img, T1, dtype=torch.long)
        qm1[2, 1, T1 - 1] = 0
        with torch.no_grad():
            o = model.generate(prefix=pf, question_tokens=q1, question_mask=qm1, pass_examples_through_encoder_one_at_a_time=True, **kw)
        arrs.update(one_tokens=q1.numpy(), one_mask=qm1.numpy(), gen_one_ids=o.sequences.numpy(), gen_one_scores=torch.stack(o.scores).numpy())
        # ---- generate: text only
        with torch.no_grad():
            o = model.generate(prefix=pf, question_tokens=q, question_mask=qm, no_prefix=True, **kw)
        arrs.update(gen_text_ids=o.sequences.numpy(), gen_text_scores=torch.stack(o.scores).numpy())
        # ---- generate: decoder prompt (vct0.py:468-480, fed by few_shot_vqa_executor.py:205-206): the encoder sees the query image's prefix only
        #      (ONE sentinel per row), the decoder continues a prompt.  (a) no row starts with the decoder start id: HF prepends it, and the
        #      reference's `outputs[:, decoder_input_ids.shape[1]:]` then keeps the prompt's last token; (b) left-padded prompts as the reference
        #      tokenises them (module_parser.py:397-399): pad id == start id, so HF prepends nothing, the pads are masked keys.
        #      (plain sequences: the reference slices the output, which a return_dict_in_generate object does not support)
        Td = 8
        qd = torch.randint(2, V - 8, (B, Td), generator=gen)
        qd[:, 2] = V - 1
        qmd = torch.ones(B, Td, dtype=torch.long)
        qmd[2, Td - 1] = 0
        qd[2, Td - 1] = 0
        dec_a = torch.randint(2, V - 8, (B, 3), generator=gen)
        dec_b = torch.randint(2, V - 8, (B, 3), generator=gen)
        dec_b_mask = torch.ones(B, 3, dtype=torch.long)
        dec_b[1, :2], dec_b_mask[1, :2] = 0, 0
        dec_b[2, :1], dec_b_mask[2, :1] = 0, 0
        kw2 = dict(max_length=9, do_sample=False, num_beams=1)
        with torch.no_grad():
            oa = model.generate(prefix=pf, question_tokens=qd, question_mask=qmd, decoder_input_ids=dec_a, decoder_attention_mask=torch.ones_like(dec_a), **kw2)
            ob = model.generate(prefix=pf, question_tokens=qd, question_mask=qmd, decoder_input_ids=dec_b, decoder_attention_mask=dec_b_mask, **kw2)
        arrs.update(dp_tokens=qd.numpy(), dp_mask=qmd.numpy(), dp_dec_a=dec_a.numpy(), dp_dec_b=dec_b.numpy(), dp_dec_b_mask=dec_b_mask.numpy(),
                    gen_dp_a_ids=oa.numpy(), gen_dp_b_ids=ob.numpy())
        save(f"vct0_{tag}.npz", **arrs)
    shutil.rmtree(tmp, ignore_errors=True)


def _install_import_stubs():
    """In-memory stand-ins for the two third-party modules the reference's data / formatter modules import at the top and this
    container lacks: ``clip`` (unused by the code under test) and ``easydict``.  ONE recursive ``EasyDict`` serves both the formatter
    and the ModuleParser step (round 2 registered a simpler one first, and a full run of this script then handed it to the
    ModuleParser step through ``sys.modules.setdefault``, which needs the recursive behaviour).  Returns the EasyDict class."""
    if "clip" not in sys.modules:
        sys.modules["clip"] = types.ModuleType("clip")
    if "easydict" in sys.modules and hasattr(sys.modules["easydict"], "EasyDict"):
        return sys.modules["easydict"].EasyDict
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            elif isinstance(v, list):
                v = [EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v]
            super().__setitem__(k, v)
        __getattr__ = dict.__getitem__
        __setattr__ = __setitem__
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    return EasyDict


def formatter_golden():
    """Strings produced by the reference ``InContextExampleFormatter`` (src/utils/in_context_examples.py:114-218) for every
    format type, 0 and 2 in-context examples, joined and per-example modes.  The module imports ``clip`` (absent, un-pinned)
    and ``easydict`` (absent) at the top: both are stubbed in memory, neither is used by the formatter."""
    import json
    EasyDict = _install_import_stubs()
    sys.path.insert(0, "/root/reference/src/utils")
    import in_context_examples as ice
    # the inputs of src/utils/in_context_examples_test.py:9-51 (test DATA)
    examples = [dict(question_id=508840006, img_key=508840, question="What color is the boys hat?", gold_answer="red"),
                dict(question_id=135938002, img_key=135938, question="Is the man wearing a shirt?", gold_answer="no")]
    query = dict(question_id=262148000, question="Where is he looking?", gold_answer="down")
    out = {"examples": examples, "query": query, "cases": []}
    for fmt in ice.InContextExampleFormatter.formats:
        if fmt.endswith("_list"):
            continue
        for n in (0, 2):
            for one_at_a_time in (False, True):
                for ensemble in (False, True):
                    f = ice.InContextExampleFormatter(fmt, pass_examples_through_encoder_one_at_a_time=one_at_a_time,
                                                      ensemble_one_shots=ensemble)
                    got = f.format_input([EasyDict(e) for e in examples[:n]], EasyDict(query))
                    out["cases"].append(dict(format_type=fmt, n=n, one_at_a_time=one_at_a_time, ensemble=ensemble, output=got))
    with open(os.path.join(HERE, "formatter.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"wrote formatter.json: {len(out['cases'])} cases")


def vqa_eval_golden():
    """Outputs of the reference ``VQAEval`` (src/utils/vqaEval.py; pure ``re`` + ``sys``, imported as is): the two
    normalisers on every key of its contraction table and on a set of answer strings, and ``evaluate()`` on a synthetic
    annotation set through a minimal stand-in for the ``VQA`` helper (only ``.qa`` and ``getQuesIds`` are touched)."""
    import copy
    import json
    import random
    sys.path.insert(0, "/root/reference/src/utils")
    import vqaEval

    class Helper:
        def __init__(self, qa):
            self.qa = qa

        def getQuesIds(self):
            return list(self.qa.keys())

    ev = vqaEval.VQAEval(Helper({}), Helper({}), n=2)
    strings = sorted(ev.contractions.keys()) + [
        "Two dogs, and a cat!", "the man's hat is red.", "1,000 people", "3.5 meters.", "it's 10 o'clock", "yes", "No.", "A  frisbee",
        "dont know", "they're playing (tennis)", "on the table; near the window", "U.S.A.", "half-full", "what?!", "none", "ten", "0.5",
        "e.g. this / that", "[bracket]", "semi;colon", "a_b", "x > y", "email@host", "`tick`", "tab\tsep", "new\nline", " the ",
        "An apple a day", "Im sure youve seen it", "somebody'd", "let's go", "she's here", "wouldn'tve", "...", "1.2.3", "v1.",
        "." * 40 + "x", "a, b", "a ,b", "mid,dle", "12,5", "question?", "two  three four five six seven eight nine zero one",
    ]
    cases = [dict(text=t, punctuation=ev.processPunctuation(t), digit_article=ev.processDigitArticle(t),
                  both=ev.processDigitArticle(ev.processPunctuation(t))) for t in strings]
    rng = random.Random(2021)
    pool = ["yes", "no", "2", "two", "red", "the red one", "Red.", "a frisbee", "frisbee", "dont know", "don't know", "1,000", "tennis",
            "playing tennis", "white and black", "black and white", "on table", "on the table", "3", "three", "0", "none", "blue", "U.S."]
    qa, res = {}, {}
    for qid in range(60):
        k = rng.choice([1, 2, 3, 4])
        choices = rng.sample(pool, k)
        answers = [dict(answer=rng.choice(choices), answer_confidence="yes", answer_id=i + 1) for i in range(10)]
        qa[qid] = dict(question_id=qid, question_type=rng.choice(["what color", "is the", "how many"]),
                       answer_type=rng.choice(["yes/no", "number", "other"]), answers=answers)
        res[qid] = dict(question_id=qid, answer=rng.choice(choices + pool[:3]) + rng.choice(["", ".", " ", "\n"]))
    ev = vqaEval.VQAEval(Helper(copy.deepcopy(qa)), Helper(copy.deepcopy(res)), n=2)
    ev.evaluate()
    out = dict(cases=cases, annotations=qa, results=res, accuracy=ev.accuracy,
               evalQA={str(k): v for k, v in ev.evalQA.items()})
    with open(os.path.join(HERE, "vqa_eval.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"wrote vqa_eval.json: {len(cases)} strings, {len(qa)} questions, overall {ev.accuracy['overall']}")


def dropin_golden():
    """Fixtures for the REAL construction path of the reference executor (src/trainers/clipcap_exector.py:52-56):
    ``ModelClass(**model_args)`` -> ``GPT2LMHeadModel.from_pretrained(model_version)`` (src/models/clipcap.py:252) ->
    ``self.model.gpt.resize_token_embeddings(len(self.tokenizer))`` after the <BOS> token was added (:55-56).

    * ``hf_gpt2_tiny/`` and ``hf_opt_tiny/``: HF-format directories (config.json + model.safetensors, DATA written by
      ``save_pretrained``) of the same seeded 2-layer models whose weights the ``clipcap_*_mlp.npz`` fixtures hold, so a test
      can hand ``model_version=<dir>`` to the executor and compare with those fixtures;
    * ``resize_gpt2.npz``: the reference model after ``resize_token_embeddings(V + 1)`` on a VQA-style batch whose answers are
      introduced by the NEW token id V: the grown embedding matrix, logits [B, L+T, V+1], loss, mapper gradients and
      generated ids."""
    import json
    import shutil
    clipcap, _ = _import_reference()
    from transformers import GPT2Config, GPT2LMHeadModel, OPTConfig, OPTForCausalLM
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import oracle
    tmp = tempfile.mkdtemp(prefix="eavqa_dropin_")

    def keep_dir(src, name):
        dst = os.path.join(HERE, name)
        os.makedirs(dst, exist_ok=True)
        for f in ("config.json", "model.safetensors"):
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        print(f"wrote {name}/: " + ", ".join(f"{f} {os.path.getsize(os.path.join(dst, f))/1024:.1f} KiB" for f in ("config.json", "model.safetensors")))

    # the tiny GPT-2 of main() (same seed, same config -> same weights; checked against the committed fixture below)
    V, E, NLAY, NH, L, D = 320, 64, 2, 4, 4, 24
    torch.manual_seed(2021)
    cfg = GPT2Config(vocab_size=V, n_embd=E, n_layer=NLAY, n_head=NH, n_positions=64, bos_token_id=V - 1, eos_token_id=V - 1,
                     attn_implementation="eager", resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    d_gpt2 = os.path.join(tmp, "gpt2")
    GPT2LMHeadModel(cfg).save_pretrained(d_gpt2)
    with np.load(os.path.join(HERE, "clipcap_gpt2_mlp.npz")) as z:
        fix = {k: z[k] for k in z.files}
    torch.manual_seed(7)
    model = clipcap.ClipCaptionPrefix(prefix_length=L, clip_length=3, prefix_size=D, num_layers=2, mapping_type="mlp", model_version=d_gpt2)
    model.gpt.config._attn_implementation = "eager"
    for k, v in model.gpt.state_dict().items():
        assert np.array_equal(v.numpy(), fix["lm." + k]), f"tiny GPT-2 differs from the committed fixture at {k}"
    for k, v in model.clip_project.state_dict().items():
        assert np.array_equal(v.numpy(), fix["map." + k]), k
    keep_dir(d_gpt2, "hf_gpt2_tiny")

    # the reference executor's two lines: a tokenizer that gained "<BOS>" -> len V + 1
    torch.manual_seed(3)
    model.gpt.resize_token_embeddings(V + 1)
    bos, pad = V, V - 1
    gen = torch.Generator().manual_seed(99)
    B, T = 4, 10
    ids = torch.randint(0, V - 2, (B, T), generator=gen)
    mask = torch.zeros(B, T, dtype=torch.long)
    for b, (ql, al) in enumerate(((3, 2), (5, 1), (2, 3), (4, 4))):
        ids[b, ql] = bos
        end = ql + 1 + al
        ids[b, end:] = pad
        mask[b, :end] = 1
    labels = oracle.label_mask_vqa(ids, pad, bos)            # the executor's rule (restated; the executor cannot be imported)
    prefix = torch.randn(B, D, generator=gen)
    model.train()
    out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels, pad_token_id=pad)
    out.loss.backward()
    grads = {"g." + n: p.grad.numpy() for n, p in model.clip_project.named_parameters()}
    model.eval()
    gq = ids[:, :6].clone()
    gm = mask[:, :6].clone()
    with torch.no_grad():
        gen_ids = model.generate(question_tokens=gq, prefix=prefix, question_mask=gm, max_length=5, pad_token_id=pad, eos_token_id=pad)
    save("resize_gpt2.npz", cfg=np.array([V, E, NLAY, NH, 64, L, D]), new_vocab=np.array(V + 1), bos_id=np.array(bos), pad_id=np.array(pad),
         wte=model.gpt.transformer.wte.weight.detach().numpy(), ids=ids.numpy(), mask=mask.numpy(), labels=labels.numpy(),
         prefix=prefix.numpy(), logits=out.logits.detach().numpy(), loss=out.loss.detach().numpy(), gen_ids=gq.numpy(), gen_mask=gm.numpy(),
         gen=np.array(gen_ids), **_np(model.clip_project.state_dict(), "map."), **grads)

    # the tiny OPT of main()
    with np.load(os.path.join(HERE, "clipcap_opt_mlp.npz")) as z:
        fix = {k: z[k] for k in z.files}
    torch.manual_seed(11)
    V, E, NLAY, NH, FFN = 336, 64, 2, 4, 96
    ocfg = OPTConfig(vocab_size=V, hidden_size=E, num_hidden_layers=NLAY, num_attention_heads=NH, ffn_dim=FFN, max_position_embeddings=64,
                     word_embed_proj_dim=E, pad_token_id=1, bos_token_id=2, eos_token_id=2, attn_implementation="eager", dropout=0.0,
                     attention_dropout=0.0)
    opt = OPTForCausalLM(ocfg).eval()
    for k, v in opt.state_dict().items():
        assert np.array_equal(v.numpy(), fix["lm." + k]), f"tiny OPT differs from the committed fixture at {k}"
    d_opt = os.path.join(tmp, "opt")
    opt.save_pretrained(d_opt)
    keep_dir(d_opt, "hf_opt_tiny")
    shutil.rmtree(tmp, ignore_errors=True)



MP_WORDS = ["question", "answer", "what", "color", "is", "the", "boys", "hat", "red", "man", "wearing", "a", "shirt", "no", "where", "he",
            "looking", "down", "combine", "facts", "and", "this", "two", "dogs", "yes", "how", "many", "are", "there", "?", ":", "."]


def build_word_tokenizer(words, eos="</s>", pad=None):
    """A real HuggingFace fast tokenizer built offline from a word list (WordLevel model, lower-cased, whitespace / punctuation
    split): what stands in for GPT2Tokenizer / AutoTokenizer.from_pretrained(name), which needs the network."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, eos: 1}
    if pad:
        vocab[pad] = 2
    for w in words:
        vocab.setdefault(w, len(vocab))
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.normalizer = normalizers.Lowercase()
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    return PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", eos_token=eos, pad_token=pad)


def module_parser_golden():
    """Batch dicts produced by the REFERENCE ``ModuleParser`` (src/data_loader_manager/module_parser.py, loaded by file path;
    ``easydict`` / ``clip`` stubbed in memory as for the formatter) and the reference's ``collate_fn`` flow
    (src/data_loader_manager/datasets/vqa2_datasets.py:94-181, restated around the reference's own ``parse_modules`` /
    ``post_processing`` because the dataset module imports cv2 / timm) for the two module configurations of the causal path:
    training (``configs/vqa2/clip_cap.jsonnet:46-75``: QAInput + EmbeddingInput) and few-shot generation
    (``configs/vqa2/few_shot_vqa_hotpotqa.jsonnet:46-56``: QInput + EmbeddingInput), with and without permutations."""
    import importlib.util
    import json
    EasyDict = _install_import_stubs()
    sys.path.insert(0, "/root/reference/src")
    spec = importlib.util.spec_from_file_location("ref_module_parser", "/root/reference/src/data_loader_manager/module_parser.py")
    mp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp)

    examples = [dict(question_id=508840006, img_key=508840, question="What color is the boys hat?", gold_answer="red"),
                dict(question_id=135938002, img_key=135938, question="Is the man wearing a shirt?", gold_answer="no"),
                dict(question_id=7, img_key=77, question="How many dogs are there?", gold_answer="two")]
    items = [dict(question_id=262148000, img_key=262148, question="Where is he looking?", gold_answer="down", answers=["down"] * 10),
             dict(question_id=262148001, img_key=135938, question="Is the man wearing a hat?", gold_answer="yes", answers=["yes", "no"])]
    gen = torch.Generator().manual_seed(5)
    D = 6
    store = {str(k): torch.randn(1, D, generator=gen).numpy() for k in (508840, 135938, 77, 262148)}

    def run(module_cfg, additional, num_shots, special_tokens, n_sentinels):
        tok = build_word_tokenizer(MP_WORDS)
        # data_loader_wrapper.py:57-62 (+ the sentinel tokens a causal LM needs, registered in reverse order)
        extra = [f"<extra_id_{i}>" for i in reversed(range(n_sentinels))]
        st = dict(special_tokens)
        own = list(getattr(tok, "additional_special_tokens", None) or getattr(tok, "extra_special_tokens", None) or [])   # renamed in transformers 5
        st["additional_special_tokens"] = own + st.get("additional_special_tokens", []) + extra
        tok.add_special_tokens(st)
        tok.pad_token = tok.eos_token                                     # clipcap_exector.py:55
        cfg = EasyDict(data_loader=EasyDict(additional=EasyDict(additional)), model_config=EasyDict(module_cfg))
        parser = mp.ModuleParser()
        parser.config, parser.tokenizer, parser.decoder_tokenizer = cfg, tok, tok
        batch = []
        for it in items:                                                  # vqa2_datasets.py:65-91
            ctx = [] if num_shots == 0 else [EasyDict(e) for e in examples][-num_shots:]
            embs = [store[str(e.img_key)] for e in ctx] + [store[str(it["img_key"])]]
            batch.append(EasyDict(question_id=it["question_id"], question=it["question"], gold_answer=it["gold_answer"],
                                  answers=it["answers"], clip_embedding=embs, in_context_examples=ctx))
        out = {}
        for kind in ("input", "decoder_input", "output"):                 # vqa2_datasets.py:105-157
            spec_ = cfg.model_config[kind + "_modules"]
            data = {}
            for sample in batch:
                for key, value in parser.parse_modules(sample, spec_.module_list, type=kind).items():
                    data.setdefault(key, []).append(value)
            out.update(parser.post_processing(EasyDict(data), spec_.postprocess_module_list))
        js = {}
        for k, v in out.items():
            js[k] = v.tolist() if torch.is_tensor(v) else v
        return dict(module_cfg=module_cfg, additional=additional, num_shots=num_shots, special_tokens=special_tokens,
                    n_sentinels=n_sentinels, vocab_size=len(tok), bos_token_id=tok.bos_token_id, pad_token_id=tok.pad_token_id,
                    sentinel_ids=[tok.convert_tokens_to_ids(f"<extra_id_{i}>") for i in range(n_sentinels)], batch=js)

    sep = {"start": "question:", "end": "answer:"}
    train_cfg = dict(
        input_modules=dict(module_list=[dict(type="QAInput", option="default", separation_tokens=sep), dict(type="EmbeddingInput", option="default")],
                           postprocess_module_list=[dict(type="PostProcessInputTokenization", option="default"),
                                                    dict(type="PostProcessClipEmbeddings", option="default")]),
        decoder_input_modules=dict(module_list=[dict(type="QuestionInput", option="default", separation_tokens=sep)],
                                   postprocess_module_list=[dict(type="PostProcessInputTokenization", option="generation")]),
        output_modules=dict(module_list=[dict(type="GenerationOutput", option="default")],
                            postprocess_module_list=[dict(type="PostProcessOutputTokenization", option="default")]))
    fewshot_cfg = dict(
        input_modules=dict(module_list=[dict(type="QInput", option="hotpotqa", separation_tokens={"start": "", "end": ""}),
                                        dict(type="EmbeddingInput", option="default")],
                           postprocess_module_list=[dict(type="PostProcessClipEmbeddings", option="default"),
                                                    dict(type="PostProcessInputTokenization", option="generation")]),
        decoder_input_modules=dict(module_list=[], postprocess_module_list=[]),
        output_modules=dict(module_list=[dict(type="GenerationOutput", option="default")],
                            postprocess_module_list=[dict(type="PostProcessOutputTokenization", option="default")]))
    base_add = dict(max_source_length=64, max_decoder_source_length=64, max_target_length=8, pass_examples_through_encoder_one_at_a_time=0,
                    sample_templates=0, ensemble_one_shots=0, num_permutations_of_in_context_examples=0)
    cases = [
        dict(name="train_qa", **run(train_cfg, dict(base_add, num_shots=0), 0, {"bos_token": "<BOS>", "additional_special_tokens": []}, 0)),
        dict(name="fewshot_2", **run(fewshot_cfg, dict(base_add, num_shots=2), 2, {"additional_special_tokens": []}, 3)),
        dict(name="zeroshot", **run(fewshot_cfg, dict(base_add, num_shots=0), 0, {"additional_special_tokens": []}, 1)),
        dict(name="fewshot_3_perm2", **run(fewshot_cfg, dict(base_add, num_shots=3, num_permutations_of_in_context_examples=2), 3,
                                           {"additional_special_tokens": []}, 4)),
    ]
    out = dict(words=MP_WORDS, examples=examples, items=items, store={k: v.tolist() for k, v in store.items()}, cases=cases)
    with open(os.path.join(HERE, "module_parser.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"wrote module_parser.json: {len(cases)} cases; " + "; ".join(f"{c['name']}: keys {sorted(c['batch'])}" for c in cases[:2]))


if __name__ == "__main__":
    if "--vct0-only" in sys.argv:
        vct0_golden()
        sys.exit(0)
    if "--module-parser-only" in sys.argv:
        module_parser_golden()
        sys.exit(0)
    if "--dropin-only" in sys.argv:
        dropin_golden()
        sys.exit(0)
    if "--vqa-eval-only" in sys.argv:
        vqa_eval_golden()
        sys.exit(0)
    if "--formatter-only" in sys.argv:
        formatter_golden()
    else:
        main()
        formatter_golden()
        vqa_eval_golden()
        dropin_golden()
        module_parser_golden()
        vct0_golden()
