"""CPU oracle for the joint geometry+attribute point-cloud codec hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy integer
arithmetic + torch-CPU fp32 matmuls + a small C rANS coder; and, as a second
summation order, chain.c: every convolution as one fused multiply-add chain per
output element in the order the product's kernels document — nn.set_order) of the algorithm in
the reference's ``model/{model,transforms,blocks,entropy_models}.py`` and of the
third-party operator semantics it relies on (MinkowskiEngine 0.5.4 — version
unpinned upstream — and compressai 1.2.4, neither present under /root/reference
nor installable here).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product package never imports it and fails loudly without its HIP library.

PARITY PINNING STATUS: "parity unpinned" at the third-party operator boundary.
The reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md §4, §8c), MinkowskiEngine / compressai cannot be imported
(ModuleNotFoundError), and no trained weights exist.  The oracle is pinned
instead by (i) hand-derivable integer known-answer tests, (ii) the structural
facts the reference itself fixes (parameter count 31,469,942 <-> README.md:125,
28-byte header model/model.py:243-250, k order transforms.py:89-127, q-map
channel order utils.py:439), (iii) C / pure-Python rANS twins that must agree
byte for byte, and (iv) committed golden vectors under tests/golden/ generated
by tests/golden/make_golden.py.

The ONE reference-generated fixture is tests/golden/bjontegaard_ref.json: the
reference's own metrics/bjontegaard.py (importable in the build container:
numpy / scipy / matplotlib only) evaluated on RD rows of its
results/Ours/test.csv by tests/golden/make_bd_golden.py.  It pins the product's
Bjontegaard_Model / Bjontegaard_Delta, NOT the codec:
the codec's arithmetic stays "parity unpinned" — that is the environment's
limit (no MinkowskiEngine / compressai), and no stand-in modules are written to
get around it.

The "kernel" summation order (oracle/chain.c, round 4) does not change that status: it
states the arithmetic the PRODUCT documents, so equality with it (tests/test_exact_parity.py:
streams, latents, voxels, colours, up to the full config-2 frame) shows the HIP path computes
exactly what it says on every layer and tile shape — a self-consistency check as strong as a
check can be — while the claim "this is MinkowskiEngine's / compressai's arithmetic" rests on
the BLAS-order restatement and stays unpinned.
"""
