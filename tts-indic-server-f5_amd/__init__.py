"""MI355X-native F5-TTS inference hot path (drop-in for the reference's
`infer_process()` / `CFM.sample()` / `vocoder.decode()` call surface).

The directory name contains hyphens, so it is imported as
``tts_indic_server_f5_amd`` through the loader module of that name at the repo
root.  All compute goes through the C-ABI library ``csrc/libf5hip.so``
(include/f5hip.h); there is no CPU fallback.
"""
__all__ = ["synth"]
