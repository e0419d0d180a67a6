"""CPU oracle for the SyncFusion generation hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain fp32 PyTorch functional ops on the CPU, the
arithmetic of the path the HIP library accelerates.  It exists to CHECK the
HIP path; it is never the thing shipped or measured.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``syncfusion_amd/`` imports it.

Pinning status, module by module -- the CPU tests of every module are in ``tests/test_oracle_cpu.py``:

===================  ========  ==========================================================================================
module               status    pinned by / unpinned because
===================  ========  ==========================================================================================
``onsetnet_ref``     PINNED    the reference itself (``/root/reference/main/onset_net.py`` + ``main/resnet.py``, imported in
                               the build container by ``oracle/gen_golden_onsetnet.py``); its input/output vectors are the
                               fixtures ``tests/golden/onsetnet_*.npz``, re-checked on every CPU run
frame transform      PINNED    the ATen antialiased-bilinear kernel torchvision's ``Resize`` calls (``tests/test_oracle_cpu.py``)
``unet_ref``         UNPINNED  the arithmetic lives in ``audio-diffusion-pytorch==0.1.3`` (requirements.txt:23) and its
``sampler_ref``                unpinned transitive dependency ``a-unet``, absent from ``/root/reference`` and from this
                               image (pip has no index): restated from SURVEY.md appendix A, frozen against itself
                               (``tests/golden/oracle_selfcheck.npz``), anchored on the reference's call sites
                               (exp/model/diffusion.yaml:11-43, main/generation.py:69-83, main/module_diffusion.py:73-77,
                               192-206) through structural / analytic invariants only
``encoder1d_ref``    UNPINNED  ``audio-encoders-pytorch==0.0.22`` (requirements.txt:24): absent, as above (appendix A.4)
``resample_ref``     UNPINNED  ``torchaudio==0.13.1`` (requirements.txt:19) is not installed: the published windowed-sinc
                               algorithm of ``torchaudio.functional.resample``, anchored on main/generation.py:91-98 and
                               on analytic properties
===================  ========  ==========================================================================================

``tools/pin_upstream.py --wheelhouse DIR`` pins the three UNPINNED groups in one command on a machine that has the packages.
"""
