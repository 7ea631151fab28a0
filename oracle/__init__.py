"""CPU oracle for the LoongX denoise hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``loongx_amd/``) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

Contents
--------
flux_modules.py  restatement of the diffusers==0.31.0 sub-modules the reference's
                 src/flux functions call (third-party, source NOT under
                 /root/reference -> "parity unpinned" at that boundary)
flux_ref.py      restatement of the reference's own src/flux/{block,transformer}.py
                 functions; pinned against golden vectors produced by running the
                 REAL reference functions in the build container (make_goldens.py)
s4.py            restatement of s4torch.S4Model (third-party, unpinned)
cs3.py, dgf.py   restatement of src/train/model.py encoders / DUAN / fusion; DUAN and
                 FeaturePyramidPooling are pinned by reference-generated goldens
make_goldens.py  imports /root/reference (build container only) and writes
                 tests/golden/*.npz
"""
