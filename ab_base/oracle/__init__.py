"""CPU oracle of the DSKD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Every function here is a CPU restatement of the reference's algorithm for one row of
SURVEY.md section 8a and cites the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
``dskd_amd`` package never does (its hot path fails loudly without the HIP library).

Pinning (what anchors each restatement to the reference):
  * lsap.c            -- scipy 1.15.3 ``linear_sum_assignment`` (the function the reference
                         calls), fuzzed here and frozen in tests/golden/lsap_*.npz.
  * assign_ref.py,
    dskd_losses_ref.py -- golden vectors produced by running the reference's OWN functions
                         (tests/golden/gen_golden.py imports the reference leaf files) plus
                         the reference tests' known answers (GIoU, KD loss).
  * msda_ref.py       -- the algorithm lives in ext-mmcv (mmcv-full>=1.3.17,<=1.6.2), absent
                         from /root/reference; restated from its published CPU formulation
                         (per-level grid_sample, bilinear, zeros, align_corners=False) and
                         cross-checked against the independent copy in the installed
                         ``transformers`` package; the reference holds no fixture for it, so
                         MSDeformAttn parity is "unpinned by the reference" (DESIGN.md).
"""
