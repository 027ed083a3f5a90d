"""Selection among several greedy generations of the same question by sequence log-probability.

Mirror of ``FewShotVQAExecutor.generate_from_ensembles`` (src/trainers/few_shot_vqa_executor.py:293-332): the
reference generates once per permutation of the in-context examples (or per single shot with
``ensemble_one_shots``), sums ``log softmax`` of the emitted tokens over the steps - skipping the special ids
``[0, 1, 2]`` - and keeps, per question, the generation with the highest sum.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

IGNORED_TOKEN_IDS = (0, 1, 2)      # few_shot_vqa_executor.py:319


def sequence_scores(sequences: Sequence[Sequence[int]], token_logprobs, ignored_ids: Sequence[int] = IGNORED_TOKEN_IDS) -> np.ndarray:
    """``token_logprobs[j][k]`` = log-probability of token k of sequence j (``greedy_decode(..., output_scores=True)``).
    Returns float64 [B]: the sum over the tokens whose id is not in ``ignored_ids`` (:316-322)."""
    lp = np.asarray(token_logprobs, dtype=np.float64)
    out = np.zeros(len(sequences), dtype=np.float64)
    for j, seq in enumerate(sequences):
        for k, tok in enumerate(seq):
            if tok not in ignored_ids:
                out[j] += lp[j, k]
    return out


def select_best(ensembled_outputs: Sequence[Sequence[Sequence[int]]], batch_sequence_scores: np.ndarray) -> List[List[int]]:
    """``ensembled_outputs[i][j]``: tokens of question j in ensemble member i; ``batch_sequence_scores``: [B, n_ens].
    ``np.argmax`` picks the FIRST member on ties, as in the reference (:328-330)."""
    best = np.argmax(batch_sequence_scores, axis=1)
    return [list(ensembled_outputs[ind][j]) for j, ind in enumerate(best)]


def generate_from_ensembles(generate: Callable[[int], Tuple[Sequence[Sequence[int]], object]], num_ensembles: int,
                            ignored_ids: Sequence[int] = IGNORED_TOKEN_IDS) -> List[List[int]]:
    """``generate(i)`` runs ensemble member i (its own prompt permutation / single shot and CLIP embeddings) and returns
    ``(sequences, token_logprobs)``, e.g. ``model.generate_fewshot(..., output_scores=True)``."""
    outputs, scores = [], []
    for i in range(num_ensembles):
        seqs, lp = generate(i)
        outputs.append(seqs)
        scores.append(sequence_scores(seqs, lp, ignored_ids))
    return select_best(outputs, np.stack(scores, axis=1))
