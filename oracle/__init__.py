"""CPU oracle for the DistillBEV hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement (numpy / plain C) of the
reference algorithms on the hot path.  It exists to *check* the HIP product
path; it is never the thing shipped or measured.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  ``distill_bev_amd`` (the product package) never imports it and
fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``oracle.lss``      pinned against the imported reference
                        ``mmdet3d/models/necks/view_transformer_mine.py`` via
                        ``tests/golden/lss_*.npz`` (made by
                        ``tests/golden/make_golden.py`` in the build container).
  * ``oracle.voxel``    dynamic/hard voxelize pinned against the reference's own
                        ``voxelization_cpu.cpp`` compiled into ``oracle/_ref``;
                        dynamic_scatter has no reference CPU path
                        ("do not support cpu yet") -> restatement-of-source,
                        pinned through torch.unique semantics + brute force.
  * ``oracle.distill``  fg-mask pinned against imported ``box_np_ops``;
                        FGD loss arithmetic: parity unpinned by reference tests
                        (the reference has none) -- restatement of
                        ``bevdet_distill.py`` checked with fp64 brute force.
"""
