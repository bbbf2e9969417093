"""CPU oracle: a restatement of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under ``brainfm_amd/`` imports this
package; the only legal callers are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  It is written from the
behaviour of jhuldr/BrainFM (file:line cited on every function), built on
stock ``torch`` CPU ops / NumPy, and pinned against golden vectors that were
produced by importing the real reference in the build container
(``tests/golden/make_golden_*.py`` -> ``tests/golden/*.npz``).

Parity status: the reference ships no known-answer tests for this path
(SURVEY.md section 4), so the oracle is pinned by reference-generated fixtures,
not by reference-owned vectors.
"""
