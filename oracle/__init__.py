"""TEST INFRASTRUCTURE ONLY — CPU oracle for the pairwise manifold-distance path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / reported baseline — never as the
thing that is measured as the product or shipped.

Two restatements live here:

* ``oracle.ref_port``  — reference-faithful torch-CPU port: the same operation
  sequence as the reference (closed-form eps-fudged 2x2/3x3 eigenvalues,
  gather by ``triu_indices``, autograd backward).  Pinned against golden
  vectors produced by importing the real reference in the development
  container (``tests/golden/gen_golden.py``).
* ``oracle/exact.c`` (+ ``oracle.exact`` ctypes wrapper) — plain-C fp64
  evaluation of the same mathematical formulas (cyclic Jacobi eigensolver,
  analytic gradients).  Pinned against ``ref_port`` fp64 and the same golden
  vectors; used for full-size checks because it runs in seconds.
"""
