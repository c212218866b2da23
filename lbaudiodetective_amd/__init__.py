"""MI355X-native LBAudioDetective hot path: HIP kernels behind the reference's C interface.

`lbaudiodetective_amd._native.lib()` loads lib/liblbaudiodetective.so (built by
`_native.build()` / `__graft_entry__.build()`); there is no CPU or PyTorch fallback.
"""
from ._native import build, lib, constant, LIB_PATH, PACKED_BYTES, PACKED_WORDS, ROWS_PER_FRAME, SHARD_KEYS  # noqa: F401
from .api import (  # noqa: F401
    Comm, Corpus, Detective, Fingerprint, Frame, Stream, LBAudioDetectiveError, noErr, pack_subfingerprint,
    frames_to_subfingerprints_device, compact_layout, compact_bands, probe_shader_clock, read_audio_url, synth_clips_device, synth_corpus_device, synth_ragged_corpus_device, unpack_packed, unpack_subfingerprint,
)
from .sharded import ShardedCorpus, broadcast_fingerprint, gather_packed, make_comm, shard_range  # noqa: F401

__all__ = [
    "build", "lib", "constant", "Corpus", "Detective", "Fingerprint", "Frame", "Stream", "LBAudioDetectiveError",
    "Comm", "ShardedCorpus", "broadcast_fingerprint", "make_comm", "shard_range", "read_audio_url", "frames_to_subfingerprints_device", "compact_layout", "compact_bands", "pack_subfingerprint", "unpack_subfingerprint", "unpack_packed",
    "synth_clips_device", "synth_corpus_device", "synth_ragged_corpus_device",
]
