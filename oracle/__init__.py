"""CPU oracle for the SqueezeDet hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference algorithm
(hazenai/SqueezeDet-PyTorch, ``src/model/squeezedet.py``, ``src/model/modules.py``,
``src/engine/detector.py``, ``src/utils/boxes.py``, ``src/datasets/base.py``).  It is
the *checker* for the HIP path; it is never the thing that is shipped or measured.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package (``squeezedet-pytorch_amd``) must
never import it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * rows A-J, L, M (backbone, decode, inference head, loss, autograd backward) are
    pinned against the reference itself, imported from /root/reference in the build
    container by ``tests/golden/make_golden.py``; the resulting vectors are committed
    under ``tests/golden/``.
  * row K (``Detector.filter``): the control flow (top-k, class loop, concatenation
    order, threshold) is pinned by running the reference's own ``Detector.filter`` with
    ``torchvision.ops.nms`` (absent in the container, third-party) bound to this
    oracle's ``nms``.  The NMS arithmetic itself (torchvision 0.3.0, not under
    /root/reference) is a restatement of its published algorithm and is pinned only by
    hand-derived known-answer tests: **NMS parity unpinned** against torchvision.
"""
from .squeezedet_oracle import *  # noqa: F401,F403
