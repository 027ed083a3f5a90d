"""On-disk formats of the reference's embedding stores: parquet column contract,
collate output, safe pickle loading.  CPU only."""
import os
import pickle
import types

import numpy as np
import pytest
import torch

from eavqa_amd.data import stores


class WordTokenizer:
    """HF-style call contract on a whitespace vocabulary (no pretrained files exist offline)."""
    pad_token_id = 0

    def __init__(self):
        self.vocab = {}

    def __call__(self, texts, padding, max_length, truncation, return_tensors):
        assert padding == "longest" and truncation and return_tensors == "pt"
        rows = [[self.vocab.setdefault(w, len(self.vocab) + 1) for w in t.split()][:max_length] for t in texts]
        T = max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (T - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (T - len(r)) for r in rows])
        return types.SimpleNamespace(input_ids=ids, attention_mask=mask)


@pytest.mark.parametrize("wrap", [False, True])
def test_cc_parquet_roundtrip_and_collate(tmp_path, wrap):
    rng = np.random.default_rng(0)
    N, D = 37, 16
    emb = rng.standard_normal((N, D)).astype(np.float32)
    urls = [f"http://x/{i}.jpg" for i in range(N)]
    caps = [stores.add_period(" ".join(["w%d" % (j % 7) for j in range(3 + i % 9)])) for i in range(N)]
    path = str(tmp_path / "cc.parquet")
    stores.write_cc_parquet(path, urls, caps, emb, wrap=wrap, row_group_size=10)
    ds = stores.ConceptualCaptionsParquet(path)
    assert len(ds) == N
    s = ds[23]
    assert stores._unwrap(s["image_url"]) == urls[23] and stores._unwrap(s["caption"]) == caps[23]
    np.testing.assert_array_equal(np.asarray(s["clip_embeddings"], dtype=np.float32), emb[23])
    assert ds[-1]["clip_embeddings"] == pytest.approx(emb[-1].tolist())
    with pytest.raises(IndexError):
        ds[N]
    tok = WordTokenizer()
    batches = list(ds.iter_batches(8, tok, max_source_length=6))
    assert [len(b["captions"]) for b in batches] == [8, 8, 8, 8, 5]
    b = batches[1]
    assert b["image_urls"] == urls[8:16] and b["captions"] == caps[8:16]
    assert b["clip_embeddings"].dtype == torch.float32 and tuple(b["clip_embeddings"].shape) == (8, D)
    np.testing.assert_array_equal(b["clip_embeddings"].numpy(), emb[8:16])
    assert b["labels"].shape == b["labels_attention_mask"].shape and b["labels"].shape[1] <= 6
    assert torch.equal(b["labels"] == -100, b["labels_attention_mask"] == 0)      # pad -> -100 (collate :94-95)
    assert torch.equal(b["input_ids"] * b["attention_mask"], torch.where(b["labels"] < 0, 0, b["labels"]))
    assert len(list(ds.iter_batches(8, tok, 6, drop_last=True))) == 4
    picked = list(ds.iter_batches(2, tok, 6, indices=[36, 0]))
    assert picked[0]["image_urls"] == [urls[36], urls[0]]


def test_rejects_a_parquet_without_the_columns(tmp_path):
    import pyarrow as pa
    import pyarrow.parquet as pq
    p = str(tmp_path / "other.parquet")
    pq.write_table(pa.table({"a": [1, 2]}), p)
    with pytest.raises(ValueError, match="clip_embeddings"):
        stores.ConceptualCaptionsParquet(p)


def test_add_period():
    assert stores.add_period(" a dog ") == "a dog."
    assert stores.add_period("a dog.") == "a dog."
    assert stores.add_period("a dog .") == "a dog."


def test_embedding_pickle_safe_loading(tmp_path):
    e = {str(i): np.random.default_rng(i).standard_normal((1, 8)).astype(np.float32) for i in (11, 5, 7)}
    p = str(tmp_path / "emb.pkl")
    stores.save_embedding_pickle(p, e)
    got = stores.load_embedding_pickle(p)
    assert set(got) == set(e) and all(np.array_equal(got[k], e[k]) for k in e)
    st = stores.EmbeddingStore.from_pickle(p)
    x = st.lookup([["11", 5], [7, "7"]])
    assert tuple(x.shape) == (2, 2, 1, 8) and np.array_equal(x[0, 1, 0].numpy(), e["5"][0]) and "5" in st and len(st) == 3

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    bad = str(tmp_path / "bad.pkl")
    with open(bad, "wb") as fh:
        pickle.dump({"1": Evil()}, fh)
    with pytest.raises(pickle.UnpicklingError, match="only numpy arrays"):
        stores.load_embedding_pickle(bad)
    obj = str(tmp_path / "obj.pkl")
    with open(obj, "wb") as fh:
        pickle.dump([1, 2], fh)
    with pytest.raises(ValueError, match="expected a dict"):
        stores.load_embedding_pickle(obj)
