"""CPU oracle for the sda posterior-sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``sda_amd/`` (the product) may import this
package.  The only legitimate importers are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker /
the timed CPU baseline, never as the thing shipped.

See ``oracle/sda_oracle.py`` for the restatement and ``oracle/README.md`` for how
it is pinned against the reference.
"""
