"""CPU oracle for the manifold-gp hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain CPU restatement (numpy / torch-CPU / one small C file) of the
arithmetic the reference performs on its sparse graph-Laplacian GP path.  It exists to
CHECK the HIP product path; it is never the thing measured or shipped:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import, call, link or execute anything under ``oracle/``;
  * nothing under ``manifold_gp_amd/`` imports it, and the product path raises when the HIP
    extension is missing instead of falling back to this code.

Pinning status ("Oracle"):
  * Laplacian / precision / Schur / scale / noise / bump arithmetic: PINNED against the
    reference's own dense restatement ``test/_dense_operators.py`` and
    ``manifold_gp/utils/torch_utils.py::bump_function`` imported by file path in the build
    container (``tests/golden/make_golden.py``), fixtures committed under ``tests/golden/``.
  * Eigen / features / out-of-sample: pinned through the dense formulas of
    ``test/_test_functions.py:107-163`` restated on top of the imported dense Laplacian.
  * k-NN search, edge coalescing, CG stopping rule, Lanczos, GP posterior: the arithmetic
    lives in third-party wheels that are absent here (faiss, torch_sparse/torch_scatter,
    linear_operator, gpytorch; all un-pinned in the reference's ``setup.py:27-33``) and no
    reference test holds a golden vector for them -> "parity unpinned" for those rows; the
    oracle restates the published algorithm (exact brute-force L2 top-k; sort + mean
    coalesce; linear_cg; dense fp64 GP formulas) and anchors on the reference call sites.

Every function cites the reference file:line it follows (paths relative to the reference
checkout root).
"""
