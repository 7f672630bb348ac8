"""CPU oracle = TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithms of the reference hot path
(ayushpradhan-dev/robust-speech-analysis-framework):

* ``smile_oracle``   – the openSMILE chain that ``Androids.conf`` specifies
  (called from ``src/opensmile_extractor.py:62-87``).            PARITY UNPINNED (1)
* ``mshds_oracle``   – the Praat analyses behind ``src/mshds_extractor.py``.   PARITY UNPINNED (1)
* ``cnnlstm_oracle`` – ``src/models.py`` (CNNLSTM.forward).  Pinned by golden vectors
  generated from the reference module itself (``tests/golden/make_cnnlstm_golden.py``).
* ``w2v2_oracle``    – the chunk loop of ``src/foundation_model_extractor.py:87-125``
  around HuggingFace ``Wav2Vec2Model`` (third-party dependency, transformers 4.54.1
  pinned by the reference; 5.15.0 installed here).  Pinned by golden vectors generated
  from the installed ``transformers`` (``tests/golden/make_w2v2_golden.py``).

(1) openSMILE 3.0.2 and praat-parselmouth 0.4.6 are third-party binaries that are
absent from /root/reference and from this image, and the reference has no tests or
recorded outputs for inputs we hold.  Those two restatements follow the published
algorithms and the parameters the reference passes; they are checked against
analytic known-answer vectors only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  The product (``robust_speech_analysis_framework_amd`` and
the ``src`` drop-ins) never does; it fails loudly if the HIP library is missing.
"""
