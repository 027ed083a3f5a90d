"""Readers / writers for the reference's on-disk embedding stores.

* Conceptual-Captions parquet: one row per image with columns ``image_url``, ``caption`` and ``clip_embeddings``
  (float32 list of length D), written by ``dataset.map(get_embeddings_from_images).to_parquet`` in
  src/tools/extract_clip_embeddings_conceptual_captions.py:60-124 and consumed by ``collate_fn`` in
  src/data_loader_manager/data_loader_conceptual_captions.py:78-104 - ``cc_collate`` below returns the same dict.
* Image-embedding pickle for VQA: ``{str(image_id): float32[1, D]}`` written by
  src/tools/extract_contrastive_image_embeddings.py:44-72.  ``load_embedding_pickle`` reads it with an unpickler that
  only reconstructs numpy arrays and builtin containers - a pickle naming any other global is refused, nothing in the
  file is executed.
* The reference's Lightning ``.ckpt`` is read by ``ClipCapExecutor.load_state_dict`` (trainers/clipcap_executor.py).

pyarrow does the parquet I/O; row groups are read lazily so a multi-GB store is never resident as a whole.
"""
from __future__ import annotations

import io
import pickle
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

CC_COLUMNS = ("image_url", "caption", "clip_embeddings")


def add_period(caption: str) -> str:
    """Caption clean-up applied before a row is stored (extract_clip_embeddings_conceptual_captions.py:101-107):
    end with exactly one period, no space before it."""
    caption = caption.strip()
    if caption[-1] != ".":
        return caption + "."
    if caption[-2] == " ":
        return caption[:-2] + "."
    return caption


def _unwrap(v):
    """The reference's collate reads ``sample["image_url"][0]`` / ``sample["caption"][0]``: its stores hold one-element
    lists.  A plain string (what ``to_parquet`` writes for a string column) is taken as is."""
    if isinstance(v, (list, tuple, np.ndarray)):
        return v[0]
    return v


def write_cc_parquet(path: str, image_urls: Sequence[str], captions: Sequence[str], embeddings, wrap: bool = False,
                     row_group_size: int = 4096) -> None:
    """``embeddings``: [N, D] float32 (tensor or array).  ``wrap=True`` stores url / caption as one-element lists."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    emb = np.asarray(embeddings.detach().cpu() if torch.is_tensor(embeddings) else embeddings, dtype=np.float32)
    if emb.ndim != 2 or emb.shape[0] != len(image_urls) or len(captions) != len(image_urls):
        raise ValueError("write_cc_parquet: need N urls, N captions and an [N, D] embedding matrix")
    urls = [[u] for u in image_urls] if wrap else list(image_urls)
    caps = [[c] for c in captions] if wrap else list(captions)
    flat = pa.array(emb.reshape(-1), type=pa.float32())
    offsets = pa.array(np.arange(0, emb.size + 1, emb.shape[1], dtype=np.int32))
    table = pa.table({"image_url": urls, "caption": caps, "clip_embeddings": pa.ListArray.from_arrays(offsets, flat)})
    pq.write_table(table, path, row_group_size=row_group_size)


class ConceptualCaptionsParquet:
    """Random access + streaming over a CC embedding store.  ``ds[i]`` is the reference's sample dict."""

    def __init__(self, path: str):
        import pyarrow.parquet as pq
        self._pf = pq.ParquetFile(path)
        missing = [c for c in CC_COLUMNS if c not in self._pf.schema_arrow.names]
        if missing:
            raise ValueError(f"{path}: not a Conceptual-Captions embedding store (missing columns {missing})")
        sizes = [self._pf.metadata.row_group(g).num_rows for g in range(self._pf.num_row_groups)]
        self._starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self._cached = (-1, None)

    def __len__(self) -> int:
        return int(self._starts[-1])

    def _group(self, g: int):
        if self._cached[0] != g:
            self._cached = (g, self._pf.read_row_group(g, columns=list(CC_COLUMNS)))
        return self._cached[1]

    def __getitem__(self, i: int) -> Dict[str, object]:
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        g = int(np.searchsorted(self._starts, i, side="right") - 1)
        t = self._group(g)
        r = i - int(self._starts[g])
        return {c: t.column(c)[r].as_py() for c in CC_COLUMNS}

    def iter_samples(self, indices: Optional[Iterable[int]] = None) -> Iterator[Dict[str, object]]:
        if indices is not None:
            for i in indices:
                yield self[i]
            return
        for g in range(self._pf.num_row_groups):
            t = self._pf.read_row_group(g, columns=list(CC_COLUMNS))
            cols = {c: t.column(c).to_pylist() for c in CC_COLUMNS}
            for r in range(t.num_rows):
                yield {c: cols[c][r] for c in CC_COLUMNS}

    def iter_batches(self, batch_size: int, tokenizer, max_source_length: int, indices: Optional[Iterable[int]] = None,
                     drop_last: bool = False) -> Iterator[Dict[str, object]]:
        buf: List[Dict[str, object]] = []
        for s in self.iter_samples(indices):
            buf.append(s)
            if len(buf) == batch_size:
                yield cc_collate(buf, tokenizer, max_source_length)
                buf = []
        if buf and not drop_last:
            yield cc_collate(buf, tokenizer, max_source_length)


def cc_collate(batch: Sequence[Dict[str, object]], tokenizer, max_source_length: int) -> Dict[str, object]:
    """``collate_fn`` of data_loader_conceptual_captions.py:78-104.  ``tokenizer`` is HF-style: called with
    ``(captions, padding="longest", max_length=..., truncation=True, return_tensors="pt")`` it returns an object with
    ``input_ids`` / ``attention_mask`` [B, T] int64, and it has ``pad_token_id``.  The executor's training step reads
    ``labels`` (ids with pad -> -100) and ``labels_attention_mask``; ``input_ids`` is added for this build's packed path."""
    image_urls = [_unwrap(s["image_url"]) for s in batch]
    captions = [_unwrap(s["caption"]) for s in batch]
    clip_embeddings = torch.stack([torch.tensor(np.asarray(s["clip_embeddings"], dtype=np.float32)) for s in batch])
    tok = tokenizer(captions, padding="longest", max_length=max_source_length, truncation=True, return_tensors="pt")
    input_ids = tok.input_ids if hasattr(tok, "input_ids") else tok["input_ids"]
    mask = tok.attention_mask if hasattr(tok, "attention_mask") else tok["attention_mask"]
    labels = input_ids.clone()
    labels[labels == tokenizer.pad_token_id] = -100
    return {"image_urls": image_urls, "captions": captions, "clip_embeddings": clip_embeddings, "labels": labels,
            "labels_attention_mask": mask, "input_ids": input_ids, "attention_mask": mask}


# ------------------------------------------------------------------------------------------ pickle stores
_ALLOWED_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("collections", "OrderedDict"),
}


class _ArrayOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"embedding store refers to {module}.{name}: only numpy arrays in builtin containers are accepted")


def load_embedding_pickle(path: str) -> Dict[str, np.ndarray]:
    """``{image_id: float32[1, D]}`` (extract_contrastive_image_embeddings.py:62-72).  Object arrays are refused too
    (their elements would be unpickled as arbitrary objects)."""
    with open(path, "rb") as fh:
        obj = _ArrayOnlyUnpickler(io.BytesIO(fh.read())).load()
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: expected a dict of embeddings, got {type(obj).__name__}")
    out: Dict[str, np.ndarray] = {}
    for k, v in obj.items():
        a = np.asarray(v)
        if a.dtype == object:
            raise ValueError(f"{path}: entry {k!r} is not a numeric array")
        out[str(k)] = a.astype(np.float32, copy=False)
    return out


def save_embedding_pickle(path: str, embeddings: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as fh:
        pickle.dump({str(k): np.asarray(v, dtype=np.float32) for k, v in embeddings.items()}, fh)


class EmbeddingStore:
    """Lookup of pre-extracted image embeddings by image key, shaped like the reference's ``EmbeddingInput`` +
    ``PostProcessClipEmbeddings`` (src/data_loader_manager/module_parser.py:234-260,466-478): each stored entry is
    ``[1, D]``, a sample with ``n_img`` images stacks to ``[n_img, 1, D]`` and a batch to ``[B, n_img, 1, D]``."""

    def __init__(self, embeddings: Dict[str, np.ndarray]):
        self._e = embeddings
        first = next(iter(embeddings.values()))
        self.dim = int(np.asarray(first).shape[-1])

    @classmethod
    def from_pickle(cls, path: str) -> "EmbeddingStore":
        return cls(load_embedding_pickle(path))

    def __contains__(self, key) -> bool:
        return str(key) in self._e

    def __len__(self) -> int:
        return len(self._e)

    def lookup(self, img_keys: Sequence[Sequence[object]]) -> torch.Tensor:
        """``img_keys[b]`` = the image keys of sample b (in-context images first, query image last)."""
        rows = []
        for keys in img_keys:
            rows.append(np.stack([np.asarray(self._e[str(k)], dtype=np.float32).reshape(1, self.dim) for k in keys]))
        return torch.from_numpy(np.stack(rows))
