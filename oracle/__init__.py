"""CPU oracle for the Loopy-SLAM hot path — TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (torch-CPU fp32 + numpy) of the
reference's per-frame neural-point render/optimise path.  It exists so that the
HIP kernels in ``loopy_slam_amd/csrc`` can be checked for parity.

Rules (enforced by tests/test_product_hygiene.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * the product package ``loopy_slam_amd`` never imports it and has no CPU
    fallback — it raises if the HIP library is missing.

Pinning: every function is checked against golden vectors captured from the
imported reference (tools/gen_golden.py -> tests/golden/*.npz), see
tests/test_oracle_golden.py.  The one exception is the neighbour search: the
reference uses faiss-gpu 1.7.2 IndexIVFFlat (approximate, un-vendored, source not
in /root/reference), so kNN parity against FAISS is *unpinned*; the contract here
is the exact radius-limited top-k by (d2, index).
"""
