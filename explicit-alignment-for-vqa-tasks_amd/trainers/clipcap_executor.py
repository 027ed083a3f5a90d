"""``ClipCapExecutor``: the reference's Lightning executor surface (src/trainers/clipcap_exector.py)
over the HIP model classes.

Kept from the reference: construction from ``config.model_config.{ModelClass, model_args}`` by name
lookup (:52-53), ``tokenizer.pad_token = eos_token`` + ``resize_token_embeddings`` (:55-56),
``configure_optimizers`` (AdamW defaults + constant-with-warmup schedule, :58-130),
``training_step(sample_batched, batch_idx)`` -> ``{"loss": ...}`` with the reference's label
construction (:134-150) and ``train/loss`` / ``train/lr[i]`` logging (:176-190), and
``_generative_step`` -> ``model.generate`` + decode (:213-311).

``pytorch_lightning`` is not available offline, so the executor is a plain class whose methods
have Lightning's names and signatures; ``fit`` below is the minimal loop (accumulate_grad_batches as
src/main.py:118 passes it to the Trainer, data-parallel gradient exchange over RCCL).
In addition to the reference's pre-extracted ``clip_embeddings`` a batch may carry raw
``pixel_values``: then the CLIP ViT runs in the loop (the north-star path).
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, Optional

import torch

from .. import ops
from ..models.clipcap import ClipCaptionModel, ClipCaptionPrefix  # noqa: F401  (looked up by name, :52)
from ..utils.attrdict import AttrDict
from .data_parallel import GradSync
from .optim import ConstantScheduleWithWarmup, CosineAnnealing, FusedAdamW, LinearScheduleWithWarmup


def vqa_label_count(input_ids: torch.Tensor, pad_token_id: int, bos_token_id) -> int:
    """Number of labels the VQA masking rule (clipcap_exector.py:134-150, on the device: ``eavqa_build_labels``) leaves
    different from -100, counted on the collate's host tensor: per row the tokens after the first <BOS> and before the
    first pad that are not <BOS> themselves, plus the first pad position (restored to the pad/eos id).  It sizes the
    scored-row compaction of the lm_head; an over-count would only pad it."""
    ids = input_ids.detach().cpu()
    T = ids.shape[1]
    ar = torch.arange(T)[None]
    is_pad = ids == pad_token_id
    first_pad = torch.where(is_pad.any(1), is_pad.float().argmax(1), torch.full((ids.shape[0],), T))
    before_pad = ar < first_pad[:, None]
    if bos_token_id is None:
        answer = torch.zeros_like(before_pad)
    else:
        is_bos = (ids == bos_token_id) & before_pad
        first_bos = torch.where(is_bos.any(1), is_bos.float().argmax(1), torch.full((ids.shape[0],), T))
        answer = before_pad & (ar > first_bos[:, None]) & (ids != bos_token_id)
    untouched = (ar > first_pad[:, None]) & ~is_pad
    return int(answer.sum() + is_pad.any(1).sum() + untouched.sum())


class ClipCapExecutor:
    def __init__(self, config, data_loader=None, *, vision_encoder=None, model=None, dtype=torch.bfloat16, device="cuda"):
        self.config = config
        self.data_loader = data_loader
        self.device = torch.device(device)
        self.tokenizer = getattr(data_loader, "tokenizer", None)
        self.decoder_tokenizer = getattr(data_loader, "decoder_tokenizer", self.tokenizer)
        if model is None:
            ModelClass = globals()[self.config.model_config.ModelClass]                     # :52
            model = ModelClass(**dict(self.config.model_config.model_args), dtype=dtype, device=device)   # :53
        self.model = model
        self.vision_encoder = vision_encoder
        if self.tokenizer is not None:
            # unconditional, as the reference: a tokenizer that HAS a pad token (OPT: pad 1, eos 2) pads with eos too from here
            # on, so the restored first-pad label (:146) is the eos id and generation's early stop sees it
            self.tokenizer.pad_token = self.tokenizer.eos_token                             # :55
            self.model.gpt.resize_token_embeddings(len(self.tokenizer))                     # :56
        self.global_step = 0
        self.logged: Dict[str, Any] = {}
        self.optimizer = None
        self.scheduler = None
        self.grad_sync: Optional[GradSync] = None

    # ------------------------------------------------------------------ Lightning-named hooks
    def log(self, name, value, **kwargs):
        self.logged[name] = value

    def configure_optimizers(self, num_training_steps: Optional[int] = None):
        """clipcap_exector.py:58-130: AdamW (torch defaults) + "linear" | "cosine" | constant-with-warmup schedule.
        ``num_training_steps`` stands in for ``trainer.estimated_stepping_batches`` (:90)."""
        tr = self.config.train
        self.optimizer = FusedAdamW(self.model.clip_project.flat, lr=tr.lr)
        sched = tr.get("scheduler", "none")
        warm = tr.get("additional", {}).get("warmup_steps", 0)
        if sched == "linear":
            if num_training_steps is None:
                raise ValueError("the linear schedule needs num_training_steps (trainer.estimated_stepping_batches)")
            self.scheduler = LinearScheduleWithWarmup(self.optimizer, warm, num_training_steps)
        elif sched == "cosine":
            self.scheduler = CosineAnnealing(self.optimizer, tr.epochs, eta_min=1e-5)
        else:
            self.scheduler = ConstantScheduleWithWarmup(self.optimizer, warm)
        return {"optimizer": self.optimizer, "lr_scheduler": {"scheduler": self.scheduler, "interval": "step", "frequency": 1}}

    def _clip_embeddings(self, batch) -> torch.Tensor:
        if "clip_embeddings" in batch:
            return batch["clip_embeddings"].to(self.device)                                 # :158-160
        if self.vision_encoder is None:
            raise KeyError("batch has no clip_embeddings and no vision encoder was given")
        px = batch["pixel_values"].to(self.device)
        if px.dim() == 5:                                                                   # [B, n_img, 3, H, W]
            B, n = px.shape[:2]
            return self.vision_encoder.encode_image(px.reshape(B * n, *px.shape[2:])).view(B, n, -1)
        return self.vision_encoder.encode_image(px)

    def training_step(self, sample_batched, batch_idx):
        """clipcap_exector.py:132-195.  VQA batches (``input_ids`` with a <BOS>-separated answer) get the
        reference's label masking; CC batches arrive with ``labels`` already masked by the collate."""
        ids = sample_batched["input_ids"].to(self.device)
        mask = sample_batched["attention_mask"].to(self.device)
        pad_id = self._pad_id()
        label_count = None
        if "labels" in sample_batched and self.config.get("data_loader", {}).get("type", "") == "DataLoaderConceptualCaptions":
            labels = sample_batched["labels"]
            if not labels.is_cuda:
                label_count = int((labels != -100).sum())       # the collate's tensor is on the host: counting is free
            labels = labels.to(self.device)
        else:
            bos = getattr(self.tokenizer, "bos_token_id", None)
            labels = ops.build_labels(ids, 0, pad_id, -1 if bos is None else bos, mode=0)   # :134-150
            if not sample_batched["input_ids"].is_cuda:
                label_count = vqa_label_count(sample_batched["input_ids"], pad_id, bos)
        prefix = self._clip_embeddings(sample_batched)
        out = self.model(question_tokens=ids, labels=labels, prefix=prefix, question_mask=mask, pad_token_id=pad_id,
                         label_count=label_count)   # :165-171
        loss = out.loss
        for i, lr in enumerate(self.scheduler.get_last_lr() if self.scheduler else []):
            self.log(f"train/lr[{i}]", lr, prog_bar=True, on_step=True, logger=True)
        self.log("train/loss", loss, on_step=True, on_epoch=True, logger=True)
        return {"loss": loss}

    def _pad_id(self) -> int:
        if self.tokenizer is not None and getattr(self.tokenizer, "pad_token_id", None) is not None:
            return self.tokenizer.pad_token_id
        cfg = self.model.gpt.cfg
        return cfg.pad_token_id if cfg.pad_token_id is not None else cfg.eos_token_id

    def validation_step(self, sample_batched, batch_idx):
        return self._generative_step(sample_batched, batch_idx)

    def test_step(self, sample_batched, batch_idx):
        return self._generative_step(sample_batched, batch_idx)

    def _generative_step(self, sample_batched, batch_idx):
        """clipcap_exector.py:213-311 (the wandb table / vqa lookup bookkeeping stays with the caller)."""
        ids = sample_batched["generative_input_ids"].to(self.device)
        mask = sample_batched["generative_attention_mask"].to(self.device)
        prefix = self._clip_embeddings(sample_batched)
        max_length = self.config.data_loader.additional.max_target_length
        eos = getattr(self.tokenizer, "eos_token_id", self.model.gpt.cfg.eos_token_id)
        outputs = self.model.generate(question_tokens=ids, question_mask=mask, prefix=prefix, max_length=max_length,
                                      pad_token_id=self._pad_id(), eos_token_id=eos)        # :236-243
        predictions = []
        bos = getattr(self.decoder_tokenizer, "bos_token_id", None)
        for index, output_sequence in enumerate(outputs):
            if bos is not None and bos in output_sequence:
                output_sequence = output_sequence[output_sequence.index(bos):]              # :261-264
            decoded = (self.decoder_tokenizer.decode(output_sequence, skip_special_tokens=True)
                       if self.decoder_tokenizer is not None else output_sequence)
            qid = sample_batched["question_ids"][index] if "question_ids" in sample_batched else index
            predictions.append({"question_id": qid, "answer": decoded})
        return {"predictions": predictions, "outputs": outputs,
                "question_ids": sample_batched.get("question_ids"), "answers": sample_batched.get("answers")}

    # ------------------------------------------------------------------ minimal trainer loop
    def fit(self, batches: Iterable[dict], accumulate_grad_batches: int = 1, max_steps: Optional[int] = None):
        """What ``trainer.fit(executor)`` does for this path (src/main.py:184-187): forward, backward into the
        mapper, every ``accumulate_grad_batches``-th batch average gradients across ranks and apply AdamW."""
        if self.optimizer is None:
            self.configure_optimizers()
        if self.grad_sync is None:
            import torch.distributed as dist
            mapper = self.model.clip_project
            # N > 1 with the MLP mapper: all-gather the gradient factors instead of all-reducing the flat gradient
            # (data_parallel.py / models/clipcap.py); the transformer mapper keeps the all-reduce
            factors = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and hasattr(mapper, "dp_factor_exchange")
            if factors:
                mapper.dp_factor_exchange = True
            self.grad_sync = GradSync(mapper.flat.grad, exchange=not factors)
        self.model.train()
        losses = []
        pending = 0                                  # micro-batches whose gradients sit in the flat buffer

        def apply():
            nonlocal pending
            self.grad_sync.start()
            self.grad_sync.finish()
            # Lightning divides every micro-batch loss by accumulate_grad_batches, also in the short last group of an epoch
            self.optimizer.step(grad_scale=self.grad_sync.grad_scale / accumulate_grad_batches)
            self.optimizer.zero_grad()
            self.scheduler.step()
            self.global_step += 1
            pending = 0

        for batch_idx, batch in enumerate(batches):
            if max_steps is not None and self.global_step >= max_steps:
                break
            loss = self.training_step(batch, batch_idx)["loss"]
            loss.backward()
            losses.append(loss.detach())
            pending += 1
            if pending == accumulate_grad_batches:
                apply()
        if pending:
            # the trailing micro-batches of the epoch: Lightning steps on the last batch whatever the remainder is;
            # leaving them would also leak their gradients into the next fit() call
            if max_steps is None or self.global_step < max_steps:
                apply()
            else:
                self.optimizer.zero_grad()
        return losses

    # ------------------------------------------------------------------ checkpoints (mapper only + LM identity)
    def state_dict(self):
        """COLLECTIVE under the sharded optimiser: every rank must call it (the reference saves on rank 0 only through Lightning's
        ModelCheckpoint, src/main.py:97-110 - with ``ShardedAdamW`` call ``state_dict()`` on all ranks and write the file on one)."""
        # under the sharded optimiser (bf16 operand mode) every rank updates only ITS shards of the fp32 master copy: gather the
        # whole master first, or the checkpoint would hold stale rows for the shards this rank does not own
        gather = getattr(self.optimizer, "gather_master", None)
        if gather is not None:
            gather()
        sd = {"model.clip_project." + k: v.detach().cpu() for k, v in self.model.clip_project.state_dict().items()}
        return {"state_dict": sd, "global_step": self.global_step,
                "optimizer": None if self.optimizer is None else {k: (v.cpu() if torch.is_tensor(v) else v)
                                                                  for k, v in self.optimizer.state_dict().items()}}

    def load_state_dict(self, ckpt) -> None:
        """Accepts this build's checkpoints and a Lightning ``.ckpt`` state_dict of the reference, whose keys are
        prefixed ``model.`` and include the whole frozen LM (``model.gpt.*`` entries are ignored: the LM is
        identified by ``model_args.model_version``)."""
        sd = ckpt.get("state_dict", ckpt)
        mapper = {k[len("model.clip_project."):]: v for k, v in sd.items() if k.startswith("model.clip_project.")}
        self.model.clip_project.load_state_dict(mapper, strict=True)
        # a reference checkpoint carries the frozen LM too; the only part of it that differs from `model_version` is the token
        # embedding grown by resize_token_embeddings (:56, rows drawn at random by HF): take it when its shape fits
        for key in ("model.gpt.transformer.wte.weight", "model.gpt.model.decoder.embed_tokens.weight"):
            if key in sd and tuple(sd[key].shape) == tuple(self.model.gpt.wte.shape):
                self.model.gpt.load_token_embeddings(sd[key])
        self.global_step = int(ckpt.get("global_step", 0))
        if ckpt.get("optimizer") and self.optimizer is not None:
            self.optimizer.load_state_dict({k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in ckpt["optimizer"].items()})
        if self.scheduler is not None:
            # the reference builds its schedulers with last_epoch = global_step (:96-124): a resumed run continues the schedule
            self.scheduler.resume(self.global_step)
