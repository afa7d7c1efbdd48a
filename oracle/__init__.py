"""CPU oracle for the F5-TTS inference hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 CPU restatement (torch CPU ops + numpy) of the
algorithm the reference runs on the path `infer_process -> CFM.sample -> DiT ->
vocoder`.  Every function cites the reference file:line it follows
(F/ = /root/reference/src/server/f5_tts/).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it — as the checker, never as the thing measured or
shipped.  The product package (`tts-indic-server-f5_amd/`) never imports it and
has no CPU fallback: it raises when the HIP library is missing.

Parity status:
  * everything the reference itself wrote (modules.py, dit.py, unett.py,
    cfm.py, chunk_text, duration rule, cross-fade) is PINNED: fixtures under
    tests/golden/ were produced by executing the reference's own source files
    in the build container (tests/golden/gen_golden.py) and this restatement
    reproduces them.
  * third-party leaves that are absent from /root/reference (torchdiffeq
    Euler, x-transformers rotary/RMSNorm, torchaudio MelSpectrogram, vocos,
    BigVGAN) are restated from their published algorithms at the versions the
    reference pins (pyproject.toml) and are "parity unpinned" by the
    reference; known-answer tests in tests/ pin them analytically.
"""
