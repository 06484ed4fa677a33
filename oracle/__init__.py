"""CPU oracle for the HRRadarPose hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rt_pose_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline.

Pinning status (see DESIGN.md "Oracle"):
  * backbone / head / loss / predict / label gaussians / PJPE: pinned against
    golden vectors produced by importing the reference's own hot-path files in
    the authoring container (``tests/golden/gen_golden.py``).
  * optimizer rule (Adam + decoupled wd + OneCycle + clip): pinned since round 6
    against seven steps captured from the reference's OWN ``OptimWrapper`` /
    ``OneCycle`` (``det3d/solver/fastai_optim.py``, ``learning_schedules_fastai.py``
    imported at file level with ``collections.Iterable`` aliased:
    ``tests/golden/gen_golden_optim.py`` -> ``optim_golden.npz``).
  * MPJPE aggregation: ``CRUW_POSE_Dataset.evaluation`` called unbound with the
    real ``eval_util`` (``tests/golden/gen_golden_eval.py`` -> ``eval_golden.json``).
  * deformable convolution (``dcn_ref``): PARITY UNPINNED by the reference (it
    ships no tests and its CUDA sources do not build here); pinned by
    known-answer tests only.
"""
