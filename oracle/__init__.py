"""CPU oracle for the CLIP-ViT -> mapping network -> causal-LM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``explicit-alignment-for-vqa-tasks_amd`` / ``eavqa_amd``).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.

Parity status: PINNED by fixtures generated in-container from the reference's
own model code (``/root/reference/src/models/clipcap.py`` and ``vct0.py``
executing on transformers 5.15.0 / torch 2.10 CPU fp32, seeded random-init
weights; generator ``tests/golden/make_golden.py``) plus the reference's own
known-answer vectors (``src/models/vct0_test.py:79-211``).  The reference's own
tests pin nothing else on this path.
"""
from .ref_cpu import *  # noqa: F401,F403
