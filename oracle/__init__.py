"""CPU parity oracle for the SUCRe hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package; nothing under ``sucre_amd/`` does (tests/test_host_logic.py::test_product_package_never_touches_the_oracle enforces it).
"""
