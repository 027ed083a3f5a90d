"""VQA answer normalisation and accuracy (answer scoring).

Host-side mirror of the reference's ``VQAEval`` (src/utils/vqaEval.py:11-158, driven from
``compute_vqa_scores`` src/trainers/metrics_processors.py:373-444) on plain dicts instead of
the ``VQA`` helper objects:

* ``normalize_punctuation``  - ``processPunctuation`` (vqaEval.py:130-140)
* ``normalize_digits_articles`` - ``processDigitArticle`` (vqaEval.py:142-155)
* ``question_accuracy``  - the leave-one-annotator-out rule ``min(1, #matches / 3)`` averaged
  over the annotators (vqaEval.py:83-114), including the reference's quirk that both sides
  are normalised only when the annotators disagree (``len(set(gtAnswers)) > 1``)
* ``evaluate`` - overall / per question type / per answer type accuracy, rounded to ``n``
  places (vqaEval.py:157-160)

The contraction table is regenerated from the list of canonical forms by the rule the
reference table follows (see ``_contraction_map``); tests/golden/vqa_eval.json pins every
entry and a set of sentences against the reference functions.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Union

# canonical spellings; the lookup keys are derived below
_CANONICAL = """
'ow's'at 'twas ain't aren't can't could've couldn't couldn't've didn't doesn't don't hadn't hadn't've hasn't
haven't he'd he'd've he's how'd how'll how's I'd've I'm I've isn't it'd it'd've it'll ma'am might've mightn't
mightn't've must've mustn't needn't not've o'clock oughtn't shan't she'd've should've shouldn't shouldn't've
somebody'd've somebody'll somebody's someone'd someone'd've someone'll someone's something'd something'd've
something'll that's there'd there'd've there're there's they'd they'd've they'll they're they've wasn't
we'd've we've weren't what'll what're what's what've when's where'd where's where've who'd who'd've who'll
who's who've why'll why're why's won't would've wouldn't wouldn't've y'all y'all'd've y'all'll you'd you'd've
you'll you're you've
""".split()


def _contraction_map() -> Dict[str, str]:
    """misspelling -> canonical form.  A form with one apostrophe is found under its apostrophe-free spelling; a form
    with several is found under every spelling that drops exactly ONE of them.  Three entries of the reference table
    do not follow the rule and are kept as they are there: two identities and one reversed pair."""
    table: Dict[str, str] = {}
    for form in _CANONICAL:
        marks = [i for i, ch in enumerate(form) if ch == "'"]
        if len(marks) == 1:
            table[form.replace("'", "")] = form
        else:
            for i in marks:
                table[form[:i] + form[i + 1:]] = form
    table["let's"] = "let's"
    table["she's"] = "she's"
    table["somebody'd"] = "somebodyd"
    return table


CONTRACTIONS = _contraction_map()
NUMBER_WORDS = {"none": "0", "zero": "0", "one": "1", "two": "2", "three": "3", "four": "4", "five": "5", "six": "6",
                "seven": "7", "eight": "8", "nine": "9", "ten": "10"}
ARTICLES = ("a", "an", "the")
PUNCTUATION = (";", "/", "[", "]", '"', "{", "}", "(", ")", "=", "+", "\\", "_", "-", ">", "<", "@", "`", ",", "?", "!")
# the reference's patterns, kept verbatim in behaviour: "(?!<=\d)" is a negative lookahead for the literal "<=digit"
# (a typo for a lookbehind upstream), so a period is removed unless a digit FOLLOWS it
_PERIOD = re.compile(r"(?!<=\d)(\.)(?!\d)")
_DIGIT_COMMA = re.compile(r"(\d)(\,)(\d)")


def normalize_punctuation(text: str) -> str:
    """vqaEval.py:130-140.  A mark is deleted when it touches a space anywhere in the INPUT (or the input contains a
    digit,digit comma), otherwise replaced by a space; then periods not followed by a digit are deleted.  The reference
    passes ``re.UNICODE`` (= 32) in the *count* position of ``sub``: at most 32 periods are removed."""
    out = text
    has_digit_comma = _DIGIT_COMMA.search(text) is not None
    for mark in PUNCTUATION:
        if (mark + " " in text or " " + mark in text) or has_digit_comma:
            out = out.replace(mark, "")
        else:
            out = out.replace(mark, " ")
    return _PERIOD.sub("", out, int(re.UNICODE))


def normalize_digits_articles(text: str) -> str:
    """vqaEval.py:142-155: lower-case, number words -> digits, drop articles, repair contractions."""
    words: List[str] = []
    for word in text.lower().split():
        word = NUMBER_WORDS.get(word, word)
        if word not in ARTICLES:
            words.append(word)
    return " ".join(CONTRACTIONS.get(w, w) for w in words)


def normalize_answer(text: str) -> str:
    return normalize_digits_articles(normalize_punctuation(text))


def _clean(text: str) -> str:
    return text.replace("\n", " ").replace("\t", " ").strip()


def question_accuracy(prediction: str, gt_answers: Sequence[str]) -> float:
    """Accuracy in [0, 1] of one prediction against the (usually 10) annotator answers (vqaEval.py:83-114)."""
    gts = [_clean(a) for a in gt_answers]
    pred = _clean(prediction)
    if len(set(gts)) > 1:
        gts = [normalize_answer(a) for a in gts]
        pred = normalize_answer(pred)
    accs = []
    for i in range(len(gts)):
        others = gts[:i] + gts[i + 1:]
        accs.append(min(1.0, sum(1 for a in others if a == pred) / 3.0))
    return sum(accs) / len(accs)


Prediction = Union[str, Mapping[str, object]]


def evaluate(annotations: Mapping[object, Mapping[str, object]],
             predictions: Union[Mapping[object, str], Iterable[Mapping[str, object]]],
             question_ids: Optional[Iterable[object]] = None, n: int = 2) -> Dict[str, object]:
    """``annotations[qid] = {"answers": [{"answer": str}, ...] | [str, ...], "question_type": str, "answer_type": str}``;
    ``predictions`` is ``{qid: answer}`` or the reference's result list ``[{"question_id": qid, "answer": str}, ...]``
    (``VQA.loadResFromDict``, src/utils/vqa_tools.py:209-242).  Returns the ``VQAEval.accuracy`` dict (percentages rounded
    to ``n`` places) plus ``perQuestion``; the metric names logged by the reference are ``metrics_to_log(result)``."""
    if not isinstance(predictions, Mapping):
        predictions = {p["question_id"]: p["answer"] for p in predictions}
    ids = list(question_ids) if question_ids is not None else list(annotations.keys())
    per_q: Dict[object, float] = {}
    by_qtype: Dict[str, List[float]] = {}
    by_atype: Dict[str, List[float]] = {}
    for qid in ids:
        ann = annotations[qid]
        gts = [a["answer"] if isinstance(a, Mapping) else a for a in ann["answers"]]
        acc = question_accuracy(str(predictions[qid]), gts)
        per_q[qid] = acc
        by_qtype.setdefault(str(ann.get("question_type", "")), []).append(acc)
        by_atype.setdefault(str(ann.get("answer_type", "")), []).append(acc)
    if not per_q:
        raise ValueError("evaluate: no questions")
    pct = lambda xs: round(100.0 * float(sum(xs)) / len(xs), n)
    return {
        "overall": pct(list(per_q.values())),
        "perQuestionType": {k: pct(v) for k, v in by_qtype.items()},
        "perAnswerType": {k: pct(v) for k, v in by_atype.items()},
        "perQuestion": {k: round(100.0 * v, n) for k, v in per_q.items()},
    }


def metrics_to_log(result: Mapping[str, object]) -> Dict[str, float]:
    """Flat metric names of ``compute_vqa_scores`` (metrics_processors.py:424-433)."""
    out = {"accuracy_overall": result["overall"]}
    for k, v in result["perQuestionType"].items():
        out[f"accuracy_QuestionType_{k}"] = v
    for k, v in result["perAnswerType"].items():
        out[f"accuracy_AnswerType_{k}"] = v
    return out
