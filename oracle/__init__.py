"""CPU oracle for the semi-Markov decode path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import, link or execute it, and there only as the checker /
the reported CPU baseline.  The product path (``action-segmentation_amd/``)
never imports this package and fails loudly when its HIP library is missing.

Contents
--------
``dense_ref.py``   torch (CPU) restatement of the reference's span scoring
                   (``semimarkov_modules.py:26-39, 284-523``) and of the pinned
                   third-party DP it hands the dense potentials to
                   (``harvardnlp/pytorch-struct@1c9b038a`` ``SemiMarkov._dp``,
                   ``to_parts``/``from_parts``, Max/Log semirings).
``smm_oracle.c``   plain-C fp64 factored DP (O(T*(K*C + C^2))) used at sizes
                   where the dense tensor no longer fits.

Parity status: the *scoring* half is pinned by golden vectors generated in the
build container from the reference's own code (``tests/golden/make_golden.py``).
The *DP* half restates a dependency that is absent from ``/root/reference``
(torch_struct) -- it is pinned by the reference's one structural known-answer
test (``src/models/test_semimarkov.py:266-323``) and by brute-force enumeration
of all segmentations on tiny lattices, but NOT by numeric outputs of
torch_struct itself: **numeric DP parity unpinned** (see DESIGN.md).
"""
