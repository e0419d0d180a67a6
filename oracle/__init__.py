"""CPU oracle for the SyncFusion generation hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain fp32 PyTorch functional ops on the CPU, the
arithmetic of the path the HIP library accelerates.  It exists to CHECK the
HIP path; it is never the thing shipped or measured.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``syncfusion_amd/`` imports it.

Pinning status:

* ``onsetnet_ref``  -- PINNED.  Checked against the reference itself
  (``/root/reference/main/onset_net.py`` + ``main/resnet.py``, imported in the
  build container) by ``oracle/gen_golden_onsetnet.py``; the resulting
  input/output vectors are committed under ``tests/golden/`` and the oracle is
  re-checked against them in ``tests/test_oracle_onsetnet.py``.
* ``unet_ref`` / ``sampler_ref`` / ``encoder1d_ref`` -- PARITY UNPINNED.  The
  arithmetic lives in third-party packages that are absent from
  ``/root/reference`` and from this image: ``audio-diffusion-pytorch==0.1.3``
  (requirements.txt:23), its unpinned transitive dependency ``a-unet``, and
  ``audio-encoders-pytorch==0.0.22`` (requirements.txt:24).  The restatement
  follows the published algorithm of those packages and
  is anchored on the reference's own call sites (exp/model/diffusion.yaml:11-43,
  main/generation.py:69-83, main/module_diffusion.py:73-77,192-206) through
  structural and analytic invariants only (tests/test_oracle_unet.py).
"""
