"""CPU oracle for the SCFlow recurrent-refinement hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``scflow_amd/`` may import this
package.  The only allowed users are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- always as the checker / the
reported CPU baseline, never as the thing that is shipped or measured.

Parity status ("Oracle pinning"):
  * the reference ships no tests and no golden vectors;
  * ``oracle.scflow_oracle`` is pinned against vectors produced in the build
    container by executing the reference's own hot-path source files
    (``tests/golden/make_golden.py``):
      - ``corr_lookup.py``, ``pose.py`` and ``CorrelationPyramid`` run with
        import-only stubs (no arithmetic is substituted);
      - every ConvModule / norm-layer based block additionally runs on a
        hand-written mini-mmcv whose semantics are a restatement of
        mmcv 1.3.16 and are UNVERIFIED against the real package, which is
        absent from the image.  Vectors of that kind carry
        ``pinned_under="mini-mmcv shim"`` in their metadata.
"""
from .scflow_oracle import *  # noqa: F401,F403
