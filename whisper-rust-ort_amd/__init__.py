"""MI355X-native Whisper hot path (log-mel -> encoder -> greedy decoder with KV past).

The compute lives in `csrc/` (hand-written HIP for gfx950 behind the C ABI declared in
`include/whisper_hip.h`); this package is the thin host-side mirror used by the tests,
`bench.py` and `__graft_entry__.py`.
"""
__version__ = "0.1.0"
