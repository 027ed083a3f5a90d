"""Answer scoring against outputs of the reference VQAEval
(tests/golden/vqa_eval.json, written by tests/golden/make_golden.py --vqa-eval-only)."""
import json
import os

import pytest

from eavqa_amd.utils import vqa_eval

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "vqa_eval.json")) as fh:
        return json.load(fh)


def test_normalisers_match_reference_strings(golden):
    assert len(golden["cases"]) > 150
    for c in golden["cases"]:
        assert vqa_eval.normalize_punctuation(c["text"]) == c["punctuation"], c["text"]
        assert vqa_eval.normalize_digits_articles(c["text"]) == c["digit_article"], c["text"]
        assert vqa_eval.normalize_answer(c["text"]) == c["both"], c["text"]


def test_contraction_table_has_the_reference_keys(golden):
    # the first 120 golden strings are the reference table's keys; each maps to something different from itself except
    # the two identity entries
    assert len(vqa_eval.CONTRACTIONS) == 120
    same = [k for k, v in vqa_eval.CONTRACTIONS.items() if k == v]
    assert sorted(same) == ["let's", "she's"]
    assert vqa_eval.CONTRACTIONS["somebody'd"] == "somebodyd"
    assert vqa_eval.CONTRACTIONS["y'alld've"] == "y'all'd've"


def test_evaluate_matches_reference(golden):
    ann = {int(k): v for k, v in golden["annotations"].items()}
    res = {int(k): v["answer"] for k, v in golden["results"].items()}
    got = vqa_eval.evaluate(ann, res, n=2)
    assert got["overall"] == golden["accuracy"]["overall"]
    assert got["perQuestionType"] == golden["accuracy"]["perQuestionType"]
    assert got["perAnswerType"] == golden["accuracy"]["perAnswerType"]
    assert {str(k): v for k, v in got["perQuestion"].items()} == golden["evalQA"]
    # the reference's result-list form and an explicit id subset
    as_list = [{"question_id": k, "answer": v} for k, v in res.items()]
    assert vqa_eval.evaluate(ann, as_list)["overall"] == got["overall"]
    sub = vqa_eval.evaluate(ann, res, question_ids=[0, 1, 2])
    assert set(sub["perQuestion"]) == {0, 1, 2}
    names = vqa_eval.metrics_to_log(got)
    assert names["accuracy_overall"] == got["overall"] and any(k.startswith("accuracy_QuestionType_") for k in names)


def test_accuracy_rule():
    gt = ["red"] * 2 + ["blue"] * 8
    # "red" matches 2 annotators: leaving one of them out leaves 1 match (1/3), leaving a "blue" out leaves 2 (2/3)
    assert vqa_eval.question_accuracy("Red.", gt) == pytest.approx((2 * (1 / 3) + 8 * (2 / 3)) / 10)
    assert vqa_eval.question_accuracy("blue", gt) == 1.0
    # unanimous annotators: NO normalisation on either side (reference quirk, vqaEval.py:97-102)
    assert vqa_eval.question_accuracy("Yes.", ["yes"] * 10) == 0.0
    assert vqa_eval.question_accuracy("yes", ["yes"] * 10) == 1.0
    with pytest.raises(ValueError):
        vqa_eval.evaluate({}, {})
