"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's hot path (bg_correction.py + detection.py).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / the timed CPU baseline.  The product
(``imageprocessing_amd``) never imports it.

Pinning status (see DESIGN.md "Oracle"):
  * scikit-image part (32 region properties): pinned by tests/golden/regionprops_*.npz,
    produced by scikit-image 0.18.3 in the build container (tests/golden/make_golden_skimage.py).
  * OpenCV part (findContours / contourArea / boundingRect / drawContours / circle /
    absdiff / bitwise_not / threshold): PARITY UNPINNED -- cv2 is not installable here and the
    reference holds no fixtures; the restatement follows the library's published algorithms
    and is cross-checked against trace-independent definitions.
  * NumPy part (mean / std / min / max): executed with NumPy itself.
"""
