"""CPU parity oracle for the SUCRe hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package; nothing under ``sucre_amd/`` does (tests/test_layout_rules.py enforces it).
"""
