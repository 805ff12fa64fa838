"""CPU oracle for the T2ONet executor/actor hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product package ``t2onet_amd`` never imports this package and has no CPU
fallback: it raises if the HIP library is missing.

Contents
--------
hsv_spec.py   the RGB<->HSV specification this build owns (kornia is an
              unpinned, un-vendored dependency of the reference; see header there)
cpu_ref.py    eager-PyTorch (CPU, fp32) restatement, op for op, of
              models/operators.py, executors/executor.py, models/attention.py,
              models/action_decoder.py, models/lang_encoder.py,
              models/actor_resnet.py, models/actor.py and the L1 step of
              experiments/t2onet/train_seq2seqL1.py

Pinning: the reference has no tests, golden vectors or fixtures for this path
(SURVEY.md section 4).  The oracle is therefore pinned against OUTPUTS OF THE
REFERENCE ITSELF, generated in the build container by ``tools/gen_golden.py``
(imports /root/reference with import shims for cv2/h5py/kornia/edgeconnect)
and committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py``
replays them.  The HSV core is pinned by hsv_spec.py, not by the reference
(kornia never shipped with it): what the goldens pin for brightness/saturation
is the reference's wrapper around the HSV round trip.
"""
