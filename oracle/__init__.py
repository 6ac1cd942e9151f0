"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the DGDM hot path: ``csr_oracle`` (numpy, integer/index work),
``dgdm_oracle`` (torch CPU, float work), ``pyg_standin`` + ``capture_golden`` (dev-container
tooling that executes the reference's classes and writes ``tests/golden``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The shipped package
(``dgdm_histopath_lab_amd``) never does.
"""
