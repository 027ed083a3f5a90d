"""Sequence-score ensembling (few_shot_vqa_executor.py:293-332) - host logic, no GPU."""
import numpy as np

from eavqa_amd.utils import ensembling


def reference_loop(outputs_scores, sequences):
    """The reference's scoring loop restated on [steps, B, V] log-probs for sequences WITHOUT a decoder-start token
    (its ``k - 1`` index compensates for T5's leading pad, few_shot_vqa_executor.py:316-322)."""
    out = np.zeros(len(sequences))
    for j, seq in enumerate(sequences):
        s = 0.0
        for k, tok in enumerate(seq):
            if tok not in [0, 1, 2]:
                s += outputs_scores[k, j, tok]
        out[j] = s
    return out


def test_sequence_scores_and_selection():
    rng = np.random.default_rng(0)
    steps, B, V, n_ens = 5, 4, 11, 3
    members, lps, want = [], [], []
    for _ in range(n_ens):
        logp = np.log(rng.dirichlet(np.ones(V), size=(steps, B)))
        seqs = rng.integers(0, V, size=(B, steps)).tolist()
        seqs[1][3:] = [2, 1]                      # eos then pad: skipped
        token_lp = np.array([[logp[k, j, seqs[j][k]] for k in range(steps)] for j in range(B)])
        members.append(seqs)
        lps.append(token_lp)
        want.append(reference_loop(logp, seqs))
    got = [ensembling.sequence_scores(m, lp) for m, lp in zip(members, lps)]
    np.testing.assert_allclose(np.stack(got, 1), np.stack(want, 1), rtol=0, atol=1e-12)
    best = ensembling.generate_from_ensembles(lambda i: (members[i], lps[i]), n_ens)
    ind = np.argmax(np.stack(want, 1), axis=1)
    assert best == [members[i][j] for j, i in enumerate(ind)]


def test_ties_pick_the_first_member():
    seqs = [[5, 6], [7, 8]]
    lp = np.zeros((2, 2))
    assert ensembling.select_best([seqs, [[9, 9], [9, 9]]], np.zeros((2, 2))) == seqs
    assert ensembling.sequence_scores([[0, 1, 2, 3]], np.array([[-1.0, -2.0, -3.0, -4.0]]))[0] == -4.0
