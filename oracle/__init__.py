"""oracle/ — CPU restatement of the reference's arithmetic for the CLIP dual-encoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under clip_dplm_amd/ (the product) may import this package; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker / the timed
CPU baseline — never as a fallback for the HIP path.

Every function cites the reference file:line it restates.  The restatement is plain PyTorch f32 on the
CPU (the path is floating point end to end; tolerance-based parity).

Pinning: the reference holds no golden vectors or known-answer tests for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
tools/make_golden.py (which imports /root/reference and the third-party transformers.EsmModel the
reference calls) and committed as data under tests/golden/.  tests/test_oracle_golden.py re-checks the
oracle against those fixtures on every run.
"""
