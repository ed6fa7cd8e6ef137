"""CPU oracle for the ZeroEGGS hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-the-formulas restatement (torch-CPU / numpy) of the
reference algorithm for the path named by BASELINE.json's north_star:
mel front-end -> speech encoder -> style-encoder VAE -> autoregressive GRU
gesture decoder -> FK/L1 loss -> RAdam.  Every function cites the reference
file:line it restates (paths relative to /root/reference).

Rules (see DESIGN.md "Oracle"):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import this package; the product package (ubisoft-laforge-zeroeggs_amd/)
    never does and fails loudly when the HIP library is missing;
  * the oracle is pinned against outputs of the *real* reference run in the
    build container: oracle/make_golden.py imports /root/reference through
    the shims in oracle/ref_shims.py and writes tests/golden/*.npz; the
    `-m "not gpu"` tests check this restatement against those fixtures.
  * Parity status: PINNED for nets (incl. the rnn_cond="film" / type="gru"
    variants) / loss / RAdam / mel / dataset index rules / animation pre- and
    post-processing (golden vectors from the reference itself).  UNPINNED for the BS.1770
    loudness stage (pyloudnorm==0.1.0 is a third-party dependency absent from
    this image; golden vectors use normalize_loudness=false).
"""
