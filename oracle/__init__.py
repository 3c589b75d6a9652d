"""oracle/ — TEST INFRASTRUCTURE ONLY.

A CPU restatement (PyTorch fp32 on the host cores — the reference is pure Python/PyTorch, so
the restatement is too) of the reference's contrast_train hot path: `Net.forward`, the loss
step, `PolyOptimizer`, and the multi-scale inference post-process.  Every function cites the
reference file:line it follows.

Parity status: PINNED.  `oracle/make_goldens.py` imports the reference itself from
/root/reference (allowed in the build container), feeds both the same
procedural weights / inputs / dropout masks / Python-RNG seed, and writes the small fixtures
under tests/golden/.  `tests/test_oracle_golden.py` re-checks the restatement against those
fixtures on any machine (the reference itself never travels).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package — never wseg_amd/ (the product path has no CPU fallback).
"""
