"""CPU oracle for the AIR hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product
package (``attend_infer_repeat_amd``) never imports this package.
"""
