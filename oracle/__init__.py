"""CPU oracle for the per-frame calibration hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain numpy / torch-CPU restatement of the
reference's algorithm (each function cites the reference file:line it
follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker -- the product
package (``soccernet-calibration-sportlight_amd``) never imports this module
and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * network / decode / camera-math / evaluator / line-join restatements are
    pinned against golden vectors captured by importing the reference in the
    build container (``tools/make_golden.py`` -> ``tests/golden/*.npz``);
  * the camera *solve* arithmetic lives in opencv-python==4.7.0.72 which is
    not available offline: that part of the oracle is "parity unpinned".
"""
