"""CPU oracle for the HRRadarPose hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rt_pose_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline.

Pinning status (see DESIGN.md "Oracle"):
  * backbone / head / loss / predict / label gaussians / PJPE: pinned against
    golden vectors produced by importing the reference's own hot-path files in
    the authoring container (``tests/golden/gen_golden.py``).
  * optimizer rule (Adam + decoupled wd + OneCycle + clip): the reference file
    cannot be imported on py>=3.10, so it is pinned by formula only.
  * deformable convolution (``dcn_ref``): PARITY UNPINNED by the reference (it
    ships no tests and its CUDA sources do not build here); pinned by
    known-answer tests only.
"""
